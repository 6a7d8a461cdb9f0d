"""The fast kernels' guard band where round 3's constants failed, against the NumPy ORACLE.

Round 3 guarded the composed ("fast") kernels with constants: 1e-6 px around rounding ties / integer bounds and 1e-6 mm
around the camera-2 plane and the depth test.  The difference between the fast and the reference evaluation of a projected
coordinate grows like 1 / (camera-2 depth): with camera 2 centred micrometres behind a frame-1 surface point it exceeds
1e-6 px while the depth guard only caught 1e-9 m (tools/guard_bound_emulation.py reproduces flipped decisions on the CPU,
profiles/r04_guard_bound_emulation.md); and the composed matrix's cancellation error scales with |translation| * fx, which
nothing bounded.  Since round 4 the band is derived per tile from a bound (mspa_common.h guard_from_bounds, slot
MSPA_MAT_BOUNDS).  These tests put the kernels INTO that regime and compare every integer output with
``oracle.np_oracle.frame_pair`` (not with the exact kernel):

  * camera 2 centred delta in {1e-9 .. 1e-3} m behind back-projected frame-1 points, the point engineered onto rounding
    ties and integer bounds +- 2e-6 px, at 96x128 (every output set + the compacted set + the generic kernel), at the
    BASELINE shape 640x480 and at ScanNet's own shape (1296x968 colour over 640x480 depth);
  * the same scene with world coordinates translated by 1e4 m and 1e6 m (the guard widens until everything takes the
    reference chain: slow, and still bit-exact);
  * a ragged shape (generic / linear-mapping kernels);
  * K1 (vertex visibility): cameras centred delta behind scene vertices.
"""
import numpy as np
import pytest
import torch

import adversarial as ADV
from mspa import engine, synth, _lib
from oracle import np_oracle as O
from test_gpu_compact import check_pair as check_compact_pair, poisoned_outputs
from test_gpu_tight import SETS, launch, unpack_bits

DEV = "cuda"
DELTAS = (1e-3, 1e-4, 1e-5, 1e-6, 3e-7, 1e-7, 3e-8, 1e-9)


def upload(K, A, E, depth_np, hw, with_rgb=True, seed=5):
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    rgb_np = np.random.default_rng(seed).integers(0, 256, (len(E),) + tuple(hw) + (3,), dtype=np.uint8) if with_rgb else None
    rgb = torch.from_numpy(rgb_np).to(DEV) if with_rgb else None
    return depth, mats, rgb, rgb_np


def check_integers(res, n, ref, hw):
    """Integer outputs of one pair against the oracle: counters, visibility (bits / bytes), pixel indices of in-view lanes."""
    P = hw[0] * hw[1]
    assert tuple(res["counts"][n]) == (ref["n_valid"], ref["n_vis"]), (tuple(res["counts"][n]), ref["n_valid"], ref["n_vis"])
    if "vis_bits" in res:
        got = unpack_bits(res["vis_bits"][n], P)
        assert np.array_equal(got, ref["vis"]), f"{int((got != ref['vis']).sum())} visibility bits differ from the oracle"
    if "vis_u8" in res:
        assert np.array_equal(res["vis_u8"][n], ref["vis"].astype(np.uint8))
    if "pix_i16" in res:
        with np.errstate(invalid="ignore"):
            inview = ref["valid"] & O.check_point_in_image_boundary(ref["uv2"], hw) & (ref["depth2"] > 0)
        pix = res["pix_i16"][n]
        assert np.array_equal(pix[inview, 0], ref["xi"][inview]) and np.array_equal(pix[inview, 1], ref["yi"][inview])
        assert (pix[~inview] == -1).all()


def run_sets_vs_oracle(K, A, E, depth_np, hw, pair_idx, sets, expect_kernel, compact=False, generic=False):
    depth, mats, rgb, _ = upload(K, A, E, depth_np, hw)
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    refs = [O.frame_pair(depth_np[a], depth_np[b], K, E[a], E[b], A, hw) for a, b in pair_idx]
    for name in sets:
        for stream in (0, _lib.PAIR_STREAM):
            res, kern = launch(depth, mats, rgb, pairs, hw, SETS[name], _lib.PAIR_FAST | stream)
            assert kern == expect_kernel, (name, kern)
            for n, ref in enumerate(refs):
                check_integers(res, n, ref, hw)
    if generic:       # one output more = the generic composed kernel
        res, kern = launch(depth, mats, rgb, pairs, hw, SETS["corr"] + ("valid_u8",), _lib.PAIR_FAST)
        assert kern in (_lib.KERNEL_PAIR_FAST, _lib.KERNEL_PAIR_FAST_LINEAR)
        for n, ref in enumerate(refs):
            check_integers(res, n, ref, hw)
            assert np.array_equal(res["valid_u8"][n].astype(bool), ref["valid"])
    if compact:
        out = poisoned_outputs(len(pair_idx), hw)
        engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST)
        assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_TIGHT
        torch.cuda.synchronize()
        out_np = {k: v.cpu().numpy() for k, v in out.items()}
        for n, ref in enumerate(refs):
            check_compact_pair(out_np, n, ref, hw)
    return refs


@pytest.mark.gpu
def test_camera2_on_frame1_points_96x128():
    """40 pairs, camera 2 centred 1e-9 .. 1e-3 m behind a frame-1 point (both orders): all four tight output sets, plain and
    streaming, the fused compacted set and the generic kernel against the oracle."""
    hw = (96, 128)
    rng = np.random.default_rng(31)
    K, A, E, depth_np, pairs = ADV.near_plane_case(rng, hw, DELTAS, per_delta=5)
    pair_idx = pairs + [(b, a) for a, b in pairs[::4]]
    refs = run_sets_vs_oracle(K, A, E, depth_np, hw, pair_idx, list(SETS), _lib.KERNEL_PAIR_FAST_TIGHT, compact=True, generic=True)
    assert sum(r["n_vis"] for r in refs) > 1000


@pytest.mark.gpu
def test_camera2_on_frame1_points_640x480():
    """The BASELINE shape: 12 pairs (delta = 1e-4, 1e-6, 1e-7, 1e-9 m; fx = 578 makes the evaluation error five times the
    96x128 one), correspondence / minimal sets and the fused compacted set against the oracle."""
    hw = (480, 640)
    rng = np.random.default_rng(32)
    K, A, E, depth_np, pairs = ADV.near_plane_case(rng, hw, (1e-4, 1e-6, 1e-7, 1e-9), per_delta=3)
    refs = run_sets_vs_oracle(K, A, E, depth_np, hw, pairs, ["corr", "minimal"], _lib.KERNEL_PAIR_FAST_TIGHT, compact=True)
    assert sum(r["n_vis"] for r in refs) > 10000


@pytest.mark.gpu
def test_camera2_on_frame1_points_scannet_shape():
    """ScanNet's own shape (1296x968 colour over 640x480 depth): the rectangular-tile kernel (the default there) and the
    wobbling-stripe kernel incl. its last stripe (MSPA_PAIR_WORD_STRIPES)."""
    hw, dhw = (968, 1296), (480, 640)
    rng = np.random.default_rng(33)
    K, A, E, depth_np, pairs = ADV.near_plane_case(rng, hw, (1e-5, 1e-7, 1e-9), per_delta=2, dhw=dhw)
    pairs = pairs[:5]
    depth, mats, _, _ = upload(K, A, E, depth_np, hw, with_rgb=False)
    pt = torch.tensor(pairs, dtype=torch.int32, device=DEV)
    refs = [O.frame_pair(depth_np[a], depth_np[b], K, E[a], E[b], A, hw) for a, b in pairs]
    for name in ("corr", "minimal"):
        for extra, want in ((0, _lib.KERNEL_PAIR_FAST_RECT), (_lib.PAIR_WORD_STRIPES, _lib.KERNEL_PAIR_FAST_SCALED)):
            res, kern = launch(depth, mats, None, pt, hw, SETS[name], _lib.PAIR_FAST | extra)
            assert kern == want
            for n, ref in enumerate(refs):
                check_integers(res, n, ref, hw)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [1e4, 1e6], ids=["1e4m", "1e6m"])
def test_scene_translated_far_from_the_origin(shift):
    """World coordinates 1e4 / 1e6 m from the origin (A' = A inv(T), E' = T E: the aligned scene is unchanged).  The composed
    matrix's cancellation error scales with |translation| * fx; the bound-derived band widens with it (at 1e6 m every lane takes
    the reference chain).  96x128: all ordered pairs of 8 adversarial poses, all output sets; 640x480: 4 pairs."""
    hw = (96, 128)
    rng = np.random.default_rng(34)
    K, A, E = ADV.adversarial_pairs(rng, 8, hw)
    boxes = synth._make_boxes(rng)
    depth_np = [ADV.render_mm(A @ e, K, hw, boxes, rng) for e in E]
    A2, E2 = ADV.translated(A, E, [shift, -shift, shift])
    pair_idx = [(a, b) for a in range(8) for b in range(8)]
    refs = run_sets_vs_oracle(K, A2, E2, depth_np, hw, pair_idx, list(SETS), _lib.KERNEL_PAIR_FAST_TIGHT, compact=True, generic=True)
    assert sum(r["n_vis"] for r in refs) > 10000
    hw = (480, 640)
    K, A, E = ADV.adversarial_pairs(rng, 4, hw)
    depth_np = [ADV.render_mm(A @ e, K, hw, boxes, rng) for e in E]
    A2, E2 = ADV.translated(A, E, [shift, -shift, shift])
    run_sets_vs_oracle(K, A2, E2, depth_np, hw, [(0, 1), (1, 0), (2, 3), (3, 0)], ["corr"], _lib.KERNEL_PAIR_FAST_TIGHT, compact=True)


@pytest.mark.gpu
def test_ragged_shape_near_plane():
    """100x130 (ragged tiles, a width that is not a multiple of 64): the generic composed kernel with stripe and linear mapping."""
    hw = (100, 130)
    rng = np.random.default_rng(35)
    K, A, E, depth_np, pairs = ADV.near_plane_case(rng, hw, (1e-4, 1e-6, 1e-8), per_delta=5)
    depth, mats, rgb, _ = upload(K, A, E, depth_np, hw)
    pt = torch.tensor(pairs, dtype=torch.int32, device=DEV)
    refs = [O.frame_pair(depth_np[a], depth_np[b], K, E[a], E[b], A, hw) for a, b in pairs]
    for outs, want in ((("vis_bits", "pix_i16", "counts"), _lib.KERNEL_PAIR_FAST_LINEAR), (("vis_u8", "pix_i16", "counts"), _lib.KERNEL_PAIR_FAST)):
        res, kern = launch(depth, mats, rgb, pt, hw, outs, _lib.PAIR_FAST)
        assert kern == want
        for n, ref in enumerate(refs):
            check_integers(res, n, ref, hw)


@pytest.mark.gpu
@pytest.mark.parametrize("hw,dhw", [((96, 128), (96, 128)), ((480, 640), (480, 640)), ((968, 1296), (480, 640))],
                         ids=["96x128", "640x480", "scannet"])
def test_vertex_visibility_cameras_on_vertices(hw, dhw):
    """K1: cameras centred 1e-9 .. 1e-3 m behind scene vertices (the vertex on a rounding tie / an image bound +- 2e-6 px),
    plus the same cameras far from the origin: the composed + compacted kernel's masks against the oracle's."""
    rng = np.random.default_rng(36)
    sc = synth.make_scene(777, n_points=4096 + 37, n_frames=2, color_hw=hw, depth_hw=dhw, invalid_pose_frac=0.0, with_color=False)
    K = sc.K
    Kd = ADV.depth_intrinsics(K, hw, dhw)
    pts = np.ascontiguousarray(sc.points[:, :3])
    boxes = synth._make_boxes(rng)
    E_al = ADV.near_vertex_cameras(rng, pts, K, hw, DELTAS, per_delta=3)
    depth_np = [ADV.render_mm(e, Kd, dhw, boxes, rng) for e in E_al]
    cam = torch.from_numpy(engine.camera_matrices(K, E_al)).to(DEV)
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    t = torch.from_numpy(pts).to(DEV)
    out = engine.vertex_visibility(t, cam, depth, hw, ("bits", "mask", "count"))
    torch.cuda.synchronize()
    mask = out["mask"].cpu().numpy().astype(bool)
    total = 0
    for k, e in enumerate(E_al):
        ref, _, _ = O.vertex_visibility(pts, K, e, depth_np[k], hw)
        assert np.array_equal(mask[k], ref), f"camera {k}: {int((mask[k] != ref).sum())} vertices differ from the oracle"
        assert int(out["count"][k]) == int(ref.sum())
        assert np.array_equal(unpack_bits(out["bits"][k].cpu().numpy(), pts.shape[0]), ref)
        total += int(ref.sum())
    assert total > 100
    # the same cameras and vertices 1e5 m from the origin
    shift = np.array([1e5, -1e5, 1e5])
    E_far = [e.copy() for e in E_al]
    for e in E_far:
        e[:3, 3] += shift
    pts_far = pts + shift
    cam = torch.from_numpy(engine.camera_matrices(K, E_far)).to(DEV)
    out = engine.vertex_visibility(torch.from_numpy(pts_far).to(DEV), cam, depth, hw, ("mask",))
    torch.cuda.synchronize()
    mask = out["mask"].cpu().numpy().astype(bool)
    for k, e in enumerate(E_far):
        ref, _, _ = O.vertex_visibility(pts_far, K, e, depth_np[k], hw)
        assert np.array_equal(mask[k], ref), f"far camera {k}"
