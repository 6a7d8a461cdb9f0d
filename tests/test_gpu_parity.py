"""HIP path (through the C ABI of include/mspa.h) against the oracles and the frozen reference outputs.

Bars (BASELINE.json north_star): visibility masks, pixel indices, counts and overlap ratios are
bit-exact; float64 products are bit-identical to the C oracle (same operation order) and within
1e-9 relative (bar: 1e-5) of the NumPy oracle / reference; float32 points equal float32(oracle).
"""
import json
import os

import numpy as np
import pytest
import torch

from golden_util import GoldenScene, close_f64, same_f64
from mspa import engine, synth, _lib
from oracle import c_oracle as C
from oracle import np_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def nan_equal_bits(a, b):
    """Bit-identical where finite/inf, NaN where NaN (NaN payload/sign is not part of the contract)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    na, nb = np.isnan(a), np.isnan(b)
    return a.shape == b.shape and np.array_equal(na, nb) and \
        np.array_equal(np.where(na, 0.0, a).view(np.int64), np.where(nb, 0.0, b).view(np.int64))


def upload_scene(g, frame_ids):
    depth = engine.depth_to_device(np.stack([g.depth[i] for i in frame_ids]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(g.K, g.A, [g.E[i] for i in frame_ids])).to(DEV)
    rgb = None
    if g.color:
        rgb = torch.from_numpy(np.stack([g.color[i] for i in frame_ids])).to(DEV)
    return depth, mats, rgb


ALL_OUT = ("vis_bits", "vis_u8", "valid_u8", "pix_i16", "xyz_f32", "xyz_f64", "uv_f64", "depth_f64", "counts")


FAST_OUT = ("vis_bits", "vis_u8", "valid_u8", "pix_i16", "xyz_f32", "counts")


def run_pairs(g, frame_ids, pair_idx, outputs=ALL_OUT, flags=0):
    depth, mats, rgb = upload_scene(g, frame_ids)
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV).reshape(-1, 2)
    outs = tuple(outputs) + (("rgba",) if rgb is not None else ())
    out = engine.alloc_pair_outputs(len(pair_idx), g.color_hw, outs, DEV)
    for t in out.values():
        t.fill_(0x5A if t.dtype in (torch.uint8,) else 0)   # poison: every element must be written
    engine.pair_reproject(depth, mats, pairs, g.color_hw, out, rgb=rgb, flags=flags)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def unpack_bits(words, n):
    b = np.unpackbits(words.view(np.uint8), bitorder="little")
    return b[:n].astype(bool)


def check_pair_against(res, n, ref_c, ref_np, image_hw, color=None):
    H, W = image_hw
    P = H * W
    valid = ref_c["valid"]
    assert np.array_equal(res["valid_u8"][n].astype(bool), valid)
    assert np.array_equal(res["vis_u8"][n].astype(bool), ref_c["vis"])
    assert np.array_equal(res["vis_u8"][n].astype(bool), ref_np["vis"])
    assert np.array_equal(unpack_bits(res["vis_bits"][n], P), ref_np["vis"])
    assert not unpack_bits(res["vis_bits"][n], res["vis_bits"][n].size * 64)[P:].any()
    assert tuple(res["counts"][n]) == (ref_np["n_valid"], ref_np["n_vis"])
    pix = res["pix_i16"][n]
    with np.errstate(invalid="ignore"):     # candidate correspondences: inside frame 2, in front of camera 2
        inview = valid & O.check_point_in_image_boundary(ref_np["uv2"], image_hw) & (ref_np["depth2"] > 0)
    assert np.array_equal(pix[inview, 0], ref_np["xi"][inview]) and np.array_equal(pix[inview, 1], ref_np["yi"][inview])
    assert (pix[~inview] == -1).all()
    assert not ref_np["vis"][~inview].any()
    if color is not None and "rgba" in res:
        rgba = res["rgba"][n].view(np.uint32)
        exp = color.reshape(-1, 3).astype(np.uint32)
        exp = exp[:, 0] | (exp[:, 1] << 8) | (exp[:, 2] << 16) | np.where(valid, 0xFF000000, 0).astype(np.uint32)
        assert np.array_equal(rgba, exp)
    f32 = res["xyz_f32"][n]
    assert np.isnan(f32[~valid]).all()
    if "xyz_f64" not in res:      # fast path: integers exact (above), float32 points to ~1 ulp of float32
        assert np.allclose(f32[valid], ref_c["xyz"][valid], rtol=2e-7, atol=1e-7)
        return
    # float64: bit-identical to the C oracle, tight against NumPy
    assert nan_equal_bits(res["xyz_f64"][n], ref_c["xyz"])
    assert nan_equal_bits(res["uv_f64"][n], ref_c["uv2"])
    assert nan_equal_bits(res["depth_f64"][n], ref_c["depth2"])
    assert close_f64(res["xyz_f64"][n], ref_np["xyz"], rtol=1e-9, scale=1e-3)
    assert close_f64(res["uv_f64"][n][valid], ref_np["uv2"][valid], rtol=1e-9, scale=1e-3)
    assert close_f64(res["depth_f64"][n], ref_np["depth2"], rtol=1e-9, scale=1e-3)
    assert np.array_equal(f32[valid], ref_c["xyz"][valid].astype(np.float32))


MODES = [("exact", 0, ALL_OUT), ("fast", _lib.PAIR_FAST, FAST_OUT)]


@pytest.mark.parametrize("mode,flags,outs", MODES, ids=["exact", "fast"])
@pytest.mark.parametrize("name", ["scene_ident", "scene_scaled"])
def test_pair_reproject_golden(name, mode, flags, outs):
    """K3 against the reference's frozen outputs (tests/golden) and both oracles."""
    g = GoldenScene(name)
    assert engine.fast_path_ok(g.K)
    ids = g.valid_image_ids
    fidx = {i: n for n, i in enumerate(ids)}
    pair_ids = [(str(a), str(b)) for a, b in g["pair_ids"]]
    res = run_pairs(g, ids, [(fidx[a], fidx[b]) for a, b in pair_ids], outputs=outs, flags=flags)
    H, W = g.color_hw
    for n, (id1, id2) in enumerate(pair_ids):
        col = g.color.get(id1)
        ref_np = O.frame_pair(g.depth[id1], g.depth[id2], g.K, g.E[id1], g.E[id2], g.A, g.color_hw,
                              col if col is not None else np.zeros((H, W, 3), np.uint8))
        ref_c = C.frame_pair(g.depth[id1], g.depth[id2], g.K, g.E[id1], g.E[id2], g.A, g.color_hw)
        check_pair_against(res, n, ref_c, ref_np, g.color_hw, col)
        # and straight against the reference arrays
        v = res["valid_u8"][n].astype(bool)
        assert np.array_equal(res["vis_u8"][n].astype(bool)[v], g[f"pair{n}_vis"])
        if mode == "fast":
            assert np.allclose(res["xyz_f32"][n][v], g[f"pair{n}_xyzrgb"][:, :3], rtol=2e-7, atol=1e-7)
            continue
        assert close_f64(res["xyz_f64"][n][v], g[f"pair{n}_xyzrgb"][:, :3], rtol=1e-9, scale=1e-3)
        assert close_f64(res["uv_f64"][n][v], g[f"pair{n}_uv"], rtol=1e-9, scale=1e-3)
        assert close_f64(res["depth_f64"][n][v], g[f"pair{n}_depth"], rtol=1e-9, scale=1e-3)


@pytest.mark.parametrize("mode,flags,outs", MODES, ids=["exact", "fast"])
@pytest.mark.parametrize("color_hw,depth_hw", [((480, 640), (480, 640)), ((968, 1296), (480, 640)),
                                               ((61, 83), (37, 53))])
def test_pair_reproject_seeded(color_hw, depth_hw, mode, flags, outs):
    """Seeded scenes at BASELINE sizes (and an odd ragged size): every output of every pixel."""
    sc = synth.make_scene(1003, n_points=64, n_frames=4, color_hw=color_hw, depth_hw=depth_hw,
                          invalid_pose_frac=0.0, with_color=(color_hw[0] <= 480))
    ids = sc.valid_image_ids
    pair_idx = [(0, 1), (2, 0), (3, 3)]
    res = run_pairs(sc, ids, pair_idx, outputs=outs, flags=flags)
    H, W = color_hw
    total_vis = 0
    for n, (a, b) in enumerate(pair_idx):
        id1, id2 = ids[a], ids[b]
        col = sc.color.get(id1)
        ref_np = O.frame_pair(sc.depth[id1], sc.depth[id2], sc.K, sc.E[id1], sc.E[id2], sc.A, color_hw,
                              col if col is not None else np.zeros((H, W, 3), np.uint8))
        ref_c = C.frame_pair(sc.depth[id1], sc.depth[id2], sc.K, sc.E[id1], sc.E[id2], sc.A, color_hw)
        check_pair_against(res, n, ref_c, ref_np, color_hw, col)
        total_vis += ref_np["n_vis"]
    assert total_vis > 0


@pytest.mark.parametrize("mode,flags,outs", MODES, ids=["exact", "fast"])
def test_pair_reproject_half_pixel_ties(mode, flags, outs):
    """Every reprojected coordinate is an exact .5 tie (dyadic geometry): half-to-even must win."""
    H, W = 48, 64
    K = np.eye(4)
    K[0, 0] = K[1, 1] = 64.0
    K[0, 2], K[1, 2] = 32.0, 24.0
    A = np.eye(4)
    A[:3, 3] = [1.0, -2.0, 0.5]
    E1 = np.eye(4)
    E2 = np.eye(4)
    E2[:3, 3] = [1.0 / 64.0, -1.0 / 64.0, 0.0]       # 0.5 px at z = 2, 1 px at z = 1, 0.25 px at z = 4
    rng = np.random.default_rng(3)
    depth1 = rng.choice(np.array([0, 1000, 2000, 2000, 2000, 4000], dtype=np.uint16), size=(H, W))
    depth2 = np.full((H, W), 2000, dtype=np.uint16)
    depth2[::3] = 2001
    sc = synth.SynthScene("ties_pair", K, A, {"00000": E1, "00005": E2}, np.zeros((1, 6)),
                          {"00000": depth1, "00005": depth2}, {}, (H, W), (H, W), np.zeros((0, 2, 3)))
    res = run_pairs(sc, ["00000", "00005"], [(0, 1), (1, 0)], outputs=outs, flags=flags)
    for n, (a, b) in enumerate([("00000", "00005"), ("00005", "00000")]):
        ref_np = O.frame_pair(sc.depth[a], sc.depth[b], K, sc.E[a], sc.E[b], A, (H, W), np.zeros((H, W, 3), np.uint8))
        ref_c = C.frame_pair(sc.depth[a], sc.depth[b], K, sc.E[a], sc.E[b], A, (H, W))
        frac = np.abs(ref_np["uv2"][ref_np["valid"]] % 1.0)
        assert (frac == 0.5).mean() > 0.3                      # the ties are really there
        check_pair_against(res, n, ref_c, ref_np, (H, W), None)


def test_pair_reproject_minimal_outputs_and_empty():
    sc = synth.make_scene(1004, n_points=64, n_frames=3, color_hw=(48, 64), depth_hw=(48, 64), invalid_pose_frac=0,
                          with_color=False)
    ids = sc.valid_image_ids
    full = run_pairs(sc, ids, [(0, 1), (1, 2)])
    for flags in (0, _lib.PAIR_FAST):       # specialised output sets of the fast kernel included
        for outs in (("vis_bits", "counts"), ("vis_bits", "pix_i16", "counts"), ("vis_bits",),
                     ("vis_u8", "pix_i16", "xyz_f32", "counts")):
            res = run_pairs(sc, ids, [(0, 1), (1, 2)], outputs=outs, flags=flags)
            for k in outs:
                if k == "xyz_f32":
                    assert np.allclose(res[k], full[k], rtol=2e-7, atol=1e-7, equal_nan=True)
                else:
                    assert np.array_equal(res[k], full[k]), (flags, outs, k)
    # zero pairs is a no-op, not an error
    depth, mats, rgb = upload_scene(sc, ids)
    out = engine.alloc_pair_outputs(0, sc.color_hw, ("vis_bits", "counts"), DEV)
    engine.pair_reproject(depth, mats, torch.zeros((0, 2), dtype=torch.int32, device=DEV), sc.color_hw, out)
    torch.cuda.synchronize()


def test_pair_reproject_errors():
    sc = synth.make_scene(1004, n_points=64, n_frames=2, color_hw=(48, 64), depth_hw=(48, 64), invalid_pose_frac=0)
    ids = sc.valid_image_ids
    depth, mats, rgb = upload_scene(sc, ids)
    pairs = torch.tensor([[0, 1]], dtype=torch.int32, device=DEV)
    out = engine.alloc_pair_outputs(1, sc.color_hw, ("rgba",), DEV)
    with pytest.raises(_lib.MspaError) as e:
        engine.pair_reproject(depth, mats, pairs, sc.color_hw, out, rgb=None)       # rgba without rgb
    assert e.value.code == _lib.MSPA_EINVAL
    out = engine.alloc_pair_outputs(1, sc.color_hw, ("counts",), DEV)
    with pytest.raises(_lib.MspaError) as e:
        engine.pair_reproject(depth, mats, pairs, sc.color_hw, out, flags=1 << 7)   # unknown flag
    assert e.value.code == _lib.MSPA_EINVAL
    bad = sc.E[ids[0]].copy()
    bad[3, 0] = 1e-9
    with pytest.raises(ValueError):
        engine.frame_matrices(sc.K, sc.A, [bad])
    with pytest.raises(ValueError):
        engine.frame_matrices(sc.K, sc.A, [np.full((4, 4), -np.inf)])


def test_fast_flag_needs_pinhole_and_bad_tensors_are_rejected():
    """MSPA_PAIR_FAST with a K whose third row is not 0 0 1 0 is refused on the host (the kernel would read the camera depth
    off the wrong row); CPU tensors / float depth never reach a kernel."""
    sc = synth.make_scene(1004, n_points=64, n_frames=2, color_hw=(48, 64), depth_hw=(48, 64), invalid_pose_frac=0, with_color=False)
    ids = sc.valid_image_ids
    K2 = sc.K.copy()
    K2[2, 3] = 0.5
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K2, sc.A, [sc.E[i] for i in ids])).to(DEV)
    pairs = torch.tensor([[0, 1]], dtype=torch.int32, device=DEV)
    out = engine.alloc_pair_outputs(1, sc.color_hw, ("vis_bits", "counts"), DEV)
    with pytest.raises(ValueError, match="pinhole"):
        engine.pair_reproject(depth, mats, pairs, sc.color_hw, out, flags=_lib.PAIR_FAST)
    # EVERY frame record is checked, once per table: a pinhole table passes (and is remembered), the same table with a LATER
    # frame's K edited in place is re-checked (version counter) and refused
    mats_ok = torch.from_numpy(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])).to(DEV)
    engine.pair_reproject(depth, mats_ok, pairs, sc.color_hw, out, flags=_lib.PAIR_FAST)
    engine.pair_reproject(depth, mats_ok, pairs, sc.color_hw, out, flags=_lib.PAIR_FAST)
    mats_ok[1, _lib.MAT_K, 11] = 0.5
    with pytest.raises(ValueError, match="pinhole"):
        engine.pair_reproject(depth, mats_ok, pairs, sc.color_hw, out, flags=_lib.PAIR_FAST)
    with pytest.raises(ValueError, match="pinhole"):
        engine.pair_correspondences(depth, mats_ok, pairs, sc.color_hw)
    engine.pair_reproject(depth, mats, pairs, sc.color_hw, out, flags=0)              # the exact kernel takes any affine K
    ref = C.frame_pair(sc.depth[ids[0]], sc.depth[ids[1]], K2, sc.E[ids[0]], sc.E[ids[1]], sc.A, sc.color_hw)
    torch.cuda.synchronize()
    assert tuple(out["counts"][0].tolist()) == (ref["n_valid"], ref["n_vis"])
    xyz = torch.from_numpy(np.ascontiguousarray(sc.points[:, :3])).to(DEV)
    cam = torch.from_numpy(engine.camera_matrices(sc.K, [sc.A @ sc.E[i] for i in ids])).to(DEV)
    with pytest.raises(ValueError):
        engine.vertex_visibility(xyz, cam.cpu(), depth, sc.color_hw)                    # CPU matrices
    with pytest.raises(ValueError):
        engine.vertex_visibility(xyz, cam, depth.to(torch.float32), sc.color_hw)        # depth dtype
    with pytest.raises(ValueError):
        engine.project_samples(xyz.cpu(), cam, depth, sc.color_hw, torch.zeros((1, 2), dtype=torch.int32, device=DEV))


def test_scene_without_valid_frames_gives_empty_tables():
    """All poses non-finite: the reference's loops run over nothing (CFR:176-189, MVI:103-123) -- empty tables, no exception."""
    from mspa.scene import SceneOnDevice
    sc = synth.make_scene(1004, n_points=128, n_frames=3, color_hw=(48, 64), depth_hw=(48, 64), invalid_pose_frac=0, with_color=False)
    E = {k: np.full((4, 4), -np.inf) for k in sc.E}
    scene = SceneOnDevice(sc.K, sc.A, E, sc.depth, sc.color_hw, sc.points, DEV)
    assert scene.ids == [] and scene.frames_relations() == {}
    t = scene.frames_relations_arrays()
    assert all(len(v) == 0 for v in t.values())
    idx = scene.visibility_index()
    assert idx["image_to_points"] == {} and len(idx["point_to_images"]) == 128 and idx["point_to_images"][5] == []
    one = {k: (sc.E[k] if n == 0 else np.full((4, 4), np.nan)) for n, k in enumerate(sc.E)}
    scene1 = SceneOnDevice(sc.K, sc.A, one, sc.depth, sc.color_hw, sc.points, DEV)
    assert len(scene1.ids) == 1 and scene1.frames_relations() == {}
    from mspa import pipeline
    assert pipeline.pair_table_rows(0, scene1).shape == (0, 7)


# ------------------------------------------------------------------------------------------
# K1 vertex visibility
# ------------------------------------------------------------------------------------------
def run_vertices(points_t, K, A, E_list, depth_list, image_hw, want=("bits", "mask", "uv", "depth", "count")):
    cam = torch.from_numpy(engine.camera_matrices(K, [A @ E for E in E_list])).to(DEV)
    depth = engine.depth_to_device(np.stack(depth_list), DEV)
    out = engine.vertex_visibility(points_t, cam, depth, image_hw, want)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("name", ["scene_ident", "scene_scaled", "ties"])
@pytest.mark.parametrize("layout", ["rows3", "rows6", "soa"])
def test_vertex_visibility_golden(name, layout):
    g = GoldenScene(name)
    ids = g.valid_image_ids
    pts = g.points
    if layout == "rows3":
        t = torch.from_numpy(np.ascontiguousarray(pts[:, :3])).to(DEV)
    elif layout == "rows6":
        t = torch.from_numpy(np.ascontiguousarray(pts)).to(DEV)[:, :3]          # aligned_points.npy as is
    else:
        t = torch.from_numpy(np.ascontiguousarray(pts[:, :3].T)).to(DEV).t()     # [3,N] storage
    res = run_vertices(t, g.K, g.A, [g.E[i] for i in ids], [g.depth[i] for i in ids], g.color_hw)
    n = pts.shape[0]
    for k, image_id in enumerate(ids):
        ref_vis = g["ref_vis"][k] if g["ref_vis"].ndim == 2 else g["ref_vis"]
        ref_uv = g["ref_uv"][k] if g["ref_uv"].ndim == 3 else g["ref_uv"]
        ref_d = g["ref_depth"][k] if g["ref_depth"].ndim == 2 else g["ref_depth"]
        assert np.array_equal(res["mask"][k].astype(bool), ref_vis)
        assert np.array_equal(unpack_bits(res["bits"][k], n), ref_vis)
        assert not unpack_bits(res["bits"][k], res["bits"][k].size * 64)[n:].any()
        assert res["count"][k] == int(ref_vis.sum())
        mc, uvc, dc = C.vertex_visibility(pts[:, :3], g.K, g.A @ g.E[image_id], g.depth[image_id], g.color_hw)
        assert nan_equal_bits(res["uv"][k], uvc) and nan_equal_bits(res["depth"][k], dc)
        fin = np.isfinite(ref_uv).all(axis=1)
        assert close_f64(res["uv"][k][fin], ref_uv[fin], rtol=1e-9, scale=1e-3)
        assert close_f64(res["depth"][k], ref_d, rtol=1e-9, scale=1e-3)
        if name == "ties":   # exact arithmetic: bit-identical to the reference itself, inf/nan included
            assert nan_equal_bits(res["uv"][k], ref_uv) and nan_equal_bits(res["depth"][k], ref_d)


def test_vertex_visibility_scene_products():
    """CFR.process_scene / MVI.process_scene rebuilt from K1 + K2 equal the reference's tables."""
    g = GoldenScene("scene_ident")
    ids = g.valid_image_ids
    t = torch.from_numpy(np.ascontiguousarray(g.points[:, :3])).to(DEV)
    cam = torch.from_numpy(engine.camera_matrices(g.K, [g.A @ g.E[i] for i in ids])).to(DEV)
    depth = engine.depth_to_device(np.stack([g.depth[i] for i in ids]), DEV)
    out = engine.vertex_visibility(t, cam, depth, g.color_hw, ("bits", "mask"))
    pairs = engine.all_pairs(len(ids), DEV)
    overlap, inter, uni = engine.pair_overlap(out["bits"], pairs, want_counts=True)
    torch.cuda.synchronize()
    keys = [(ids[i], ids[j]) for i, j in pairs.cpu().numpy()]
    assert keys == [tuple(str(s) for s in k) for k in g["cfr_pairs"]]
    assert same_f64(overlap.cpu().numpy(), g["cfr_values"][:, 0])
    ref = g.json("mvi_json")
    mask = out["mask"].cpu().numpy().astype(bool)
    for k, image_id in enumerate(ids):
        assert np.where(mask[k])[0].tolist() == ref["image_to_points"][image_id]


@pytest.mark.parametrize("name", ["scene_ident", "scene_scaled", "ties"])
def test_vertex_visibility_fast_golden(name):
    """The composed + guarded K1 kernel (what runs when no float64 output is requested) against the reference's frozen masks,
    engineered rounding ties and depth equalities included."""
    g = GoldenScene(name)
    ids = g.valid_image_ids
    t = torch.from_numpy(np.ascontiguousarray(g.points[:, :3])).to(DEV)
    res = run_vertices(t, g.K, g.A, [g.E[i] for i in ids], [g.depth[i] for i in ids], g.color_hw, want=("bits", "mask", "count"))
    n = g.points.shape[0]
    for k in range(len(ids)):
        ref_vis = g["ref_vis"][k] if g["ref_vis"].ndim == 2 else g["ref_vis"]
        assert np.array_equal(res["mask"][k].astype(bool), ref_vis)
        assert np.array_equal(unpack_bits(res["bits"][k], n), ref_vis)
        assert res["count"][k] == int(ref_vis.sum())


@pytest.mark.parametrize("color_hw,depth_hw", [((480, 640), (480, 640)), ((968, 1296), (480, 640)), ((61, 83), (37, 53))])
def test_vertex_visibility_fast_equals_exact_random_poses(color_hw, depth_hw):
    """Adversarial cameras (looking away, inside the geometry, coincident with vertices, grazing) and vertices engineered onto
    the camera plane / the optical axis: bitsets of the fast kernel == bitsets of the reference-order kernel (which is
    bit-identical to the C oracle, test_vertex_visibility_golden), also for a non-pinhole K (reference chain per image)."""
    rng = np.random.default_rng(41)
    K = synth.intrinsics_for(color_hw)
    Kd = K.copy()
    Kd[0] *= depth_hw[1] / color_hw[1]
    Kd[1] *= depth_hw[0] / color_hw[0]
    boxes = synth._make_boxes(rng)
    pts = synth._sample_surface_points(rng, boxes, 20000)
    E, depth = [], []
    for k in range(20):
        eye = rng.uniform([0.3, 0.3, 0.3], [5.7, 5.7, 2.7])
        tgt = synth.ROOM / 2 + rng.normal(0, 1.5, 3)
        e = synth._look_at(eye, tgt)
        if k % 5 == 1:
            e[:3, 3] = pts[rng.integers(len(pts))]                 # camera centre ON a vertex: z = 0, u = v = NaN
        if k % 5 == 2:
            e[:3, 3] = pts[rng.integers(len(pts))] - e[:3, 2] * 0.5   # a vertex exactly on the optical axis
        e = synth._roundtrip_f(e)
        E.append(e)
        z = synth.render_depth(e, Kd, depth_hw, boxes)
        mm = np.clip(np.rint(z * 1000.0 + rng.normal(0, 3.0, z.shape)), 0, 65535).astype(np.uint16)
        mm[rng.random(mm.shape) < 0.05] = 0
        depth.append(mm)
    t = torch.from_numpy(np.ascontiguousarray(pts)).to(DEV)
    A = np.eye(4)
    fast = run_vertices(t, K, A, E, depth, color_hw, want=("bits", "mask", "count"))
    exact = run_vertices(t, K, A, E, depth, color_hw, want=("bits", "mask", "count", "uv"))
    for k in ("bits", "mask", "count"):
        assert np.array_equal(fast[k], exact[k]), k
    assert int(exact["count"].sum()) > 5000
    K2 = K.copy()
    K2[2, 3] = 0.25                                                  # not a pinhole: third row 0 0 1 0.25
    fast2 = run_vertices(t, K2, A, E[:9], depth[:9], color_hw, want=("bits", "count"))
    exact2 = run_vertices(t, K2, A, E[:9], depth[:9], color_hw, want=("bits", "count", "depth"))
    assert np.array_equal(fast2["bits"], exact2["bits"]) and np.array_equal(fast2["count"], exact2["count"])


def test_vertex_visibility_compacted_list_extremes():
    """The compacted K1 kernel's candidate list at its extremes, against the reference-order kernel: every (vertex, image) pair
    a candidate (a wall seen head-on: the list is full, 2 048 entries per block), none (cameras looking away: empty list),
    vertices with NaN / inf coordinates and vertices behind the camera mixed in, an image count and a vertex count that leave a
    ragged last group / last block / last bitset word, depth frames with zeros and with samples that tie the projected depth."""
    rng = np.random.default_rng(77)
    hw = (96, 128)
    K = synth.intrinsics_for(hw)
    n = 256 * 5 + 67
    wall = np.stack([rng.uniform(2.4, 3.6, n), rng.uniform(2.5, 3.5, n), np.full(n, 0.0)], axis=1)      # on the floor z = 0
    E_see, E_away = [], []
    for k in range(9):
        eye = np.array([3.0 + 0.05 * k, 3.0 - 0.04 * k, 2.6])
        E_see.append(synth._roundtrip_f(synth._look_at(eye, np.array([3.4, 3.3, 0.0]))))
        E_away.append(synth._roundtrip_f(synth._look_at(eye, np.array([3.4, 3.3, 6.0]))))
    def frames(E_list, pts):
        out = []
        for e in E_list:
            d = np.full(hw, 0, np.uint16)
            cam = (np.linalg.inv(e) @ np.c_[pts, np.ones(len(pts))].T).T
            ok = cam[:, 2] > 0.05
            uvz = (K[:3, :3] @ cam[ok, :3].T).T
            u, v = np.rint(uvz[:, 0] / uvz[:, 2]).astype(int), np.rint(uvz[:, 1] / uvz[:, 2]).astype(int)
            inb = (u >= 0) & (u < hw[1]) & (v >= 0) & (v < hw[0])
            mm = np.rint(cam[ok, 2][inb] * 1000.0).astype(np.int64)
            d[v[inb], u[inb]] = np.clip(mm + rng.integers(-1, 2, mm.shape), 0, 65535).astype(np.uint16)   # -1 / 0 / +1 mm: ties included
            d[rng.random(hw) < 0.02] = 0
            out.append(d)
        return out
    pts = wall.copy()
    pts[5] = [np.nan, 1.0, 1.0]
    pts[300] = [np.inf, 0.0, 0.0]
    pts[301] = [3.0, 3.0, 9.0]                                           # behind the cameras that look down
    A = np.eye(4)
    t = torch.from_numpy(np.ascontiguousarray(pts)).to(DEV)
    for E_list in (E_see, E_away, E_see[:3] + E_away[:2] + E_see[3:7]):
        depth = frames(E_list, wall)
        fast = run_vertices(t, K, A, E_list, depth, hw, want=("bits", "mask", "count"))
        exact = run_vertices(t, K, A, E_list, depth, hw, want=("bits", "mask", "count", "uv"))
        for k in ("bits", "mask", "count"):
            assert np.array_equal(fast[k], exact[k]), k
        only_bits = run_vertices(t, K, A, E_list, depth, hw, want=("bits",))
        assert np.array_equal(only_bits["bits"], exact["bits"])
        only_count = run_vertices(t, K, A, E_list, depth, hw, want=("count",))          # the per-block atomic form of the counts
        assert np.array_equal(only_count["count"], exact["count"])
    seen = run_vertices(t, K, A, E_see, frames(E_see, wall), hw, want=("count",))["count"]
    assert seen.min() > 0.2 * n                                          # the full-list case really is dense
    away = run_vertices(t, K, A, E_away, frames(E_away, wall), hw, want=("count",))["count"]
    assert away.sum() == 0


def test_vertex_visibility_large_and_edge():
    sc = synth.make_scene(1005, n_points=131072 + 37, n_frames=11, color_hw=(480, 640), invalid_pose_frac=0,
                          with_color=False)
    ids = sc.valid_image_ids
    t = torch.from_numpy(np.ascontiguousarray(sc.points[:, :3])).to(DEV)
    res = run_vertices(t, sc.K, sc.A, [sc.E[i] for i in ids], [sc.depth[i] for i in ids], sc.color_hw,
                       want=("bits", "count", "mask"))
    n = sc.points.shape[0]
    for k, image_id in enumerate(ids):
        m, _, _ = O.vertex_visibility(sc.points[:, :3], sc.K, sc.A @ sc.E[image_id], sc.depth[image_id], sc.color_hw)
        assert np.array_equal(res["mask"][k].astype(bool), m)
        assert np.array_equal(unpack_bits(res["bits"][k], n), m)
        assert res["count"][k] == int(m.sum()) > 0
    # empty inputs are no-ops
    cam = torch.from_numpy(engine.camera_matrices(sc.K, [sc.A @ sc.E[ids[0]]])).to(DEV)
    depth = engine.depth_to_device(sc.depth[ids[0]][None], DEV)
    out = engine.vertex_visibility(torch.zeros((0, 3), dtype=torch.float64, device=DEV), cam, depth, sc.color_hw,
                                   ("bits", "count"))
    torch.cuda.synchronize()
    assert out["bits"].shape == (1, 0) and int(out["count"][0]) == 0


# ------------------------------------------------------------------------------------------
# K2 pair overlap
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_bits", [64, 700, 4096 + 64, 131072])
def test_pair_overlap(n_bits):
    rng = np.random.default_rng(n_bits)
    F = 9
    masks = rng.random((F, n_bits)) < rng.random((F, 1)) * 0.4
    masks[3] = False
    masks[5] = False                       # an empty union -> NaN (CFR:136)
    n_words = (n_bits + 63) // 64
    padded = np.zeros((F, n_words * 64), dtype=bool)
    padded[:, :n_bits] = masks
    words = np.packbits(padded, axis=1, bitorder="little").view(np.int64)
    bits = torch.from_numpy(np.ascontiguousarray(words)).to(DEV)
    pairs = engine.all_pairs(F, DEV)
    overlap, inter, uni = engine.pair_overlap(bits, pairs, want_counts=True)
    torch.cuda.synchronize()
    overlap, inter, uni = overlap.cpu().numpy(), inter.cpu().numpy(), uni.cpu().numpy()
    for p, (i, j) in enumerate(pairs.cpu().numpy()):
        ref = O.calculate_camera_overlap(masks[i], masks[j])
        c, ci, cu = C.pair_overlap(masks[i], masks[j])
        assert inter[p] == ci == int((masks[i] & masks[j]).sum()) and uni[p] == cu
        assert np.isnan(ref) == (cu == 0)
        if np.isnan(ref):                  # empty union (both masks empty), e.g. the pair (3, 5)
            assert np.isnan(overlap[p]) and cu == 0
        else:
            assert same_f64(overlap[p], ref)


@pytest.mark.parametrize("F,n_bits", [(2, 64), (9, 700), (33, 4096 + 64), (70, 131072), (97, 131072 + 64 * 3), (320, 8192)])
def test_scene_overlap_tiled(F, n_bits):
    """The tiled all-pairs K2 (32 x 32 row blocks, LDS-staged chunks, word slices reduced through the workspace) against the
    one-wave-per-pair kernel (bit-equal: same integers, same float64 division) and the oracle; odd word counts, row counts
    that are not multiples of the tile, empty rows (NaN for an empty union, CFR:136)."""
    rng = np.random.default_rng(F * 1000 + n_bits)
    masks = rng.random((F, n_bits)) < rng.random((F, 1)) * 0.4
    masks[F // 2] = False
    if F > 3:
        masks[3] = False
    n_words = (n_bits + 63) // 64
    padded = np.zeros((F, n_words * 64), dtype=bool)
    padded[:, :n_bits] = masks
    bits = torch.from_numpy(np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little").view(np.int64))).to(DEV)
    pairs = engine.all_pairs(F, DEV)
    ref_o, ref_i, ref_u = engine.pair_overlap(bits, pairs, want_counts=True)
    got_o, got_i, got_u = engine.scene_overlap(bits, want_counts=True)
    torch.cuda.synchronize()
    assert torch.equal(ref_i, got_i) and torch.equal(ref_u, got_u)
    assert nan_equal_bits(got_o.cpu().numpy(), ref_o.cpu().numpy())
    assert np.isnan(got_o.cpu().numpy()).sum() == (1 if F > 3 else 0)
    pn = pairs.cpu().numpy()
    for p in rng.choice(len(pn), size=min(len(pn), 40), replace=False):
        i, j = pn[p]
        ref = O.calculate_camera_overlap(masks[i], masks[j])
        assert (np.isnan(ref) and np.isnan(float(got_o[p]))) or same_f64(float(got_o[p]), ref)
    # rectangle form: |a_i & b_j| for two different row sets, and the symmetric shortcut
    nb = max(1, F // 3)
    other = rng.random((nb, n_bits)) < 0.3
    padded_b = np.zeros((nb, n_words * 64), dtype=bool)
    padded_b[:, :n_bits] = other
    bits_b = torch.from_numpy(np.ascontiguousarray(np.packbits(padded_b, axis=1, bitorder="little").view(np.int64))).to(DEV)
    rect = engine.overlap_matrix(bits_b, bits).cpu().numpy()
    assert np.array_equal(rect, other.astype(np.int64) @ masks.astype(np.int64).T)
    sym = engine.overlap_matrix(bits, bits).cpu().numpy()
    assert np.array_equal(sym, masks.astype(np.int64) @ masks.astype(np.int64).T)


@pytest.mark.parametrize("R,n_bits", [(1, 64), (5, 700), (64, 4096 + 64), (70, 131072), (320, 8192 + 64 * 5)])
def test_bitset_csr_and_transpose(R, n_bits):
    """K9: device-side compaction of a bit matrix and of its transpose == np.nonzero, row by row."""
    rng = np.random.default_rng(R * 7 + n_bits)
    masks = rng.random((R, n_bits)) < rng.random((R, 1)) * 0.3
    masks[R // 2] = False
    masks[0, :3] = True
    masks[-1, -1] = True
    n_words = (n_bits + 63) // 64
    padded = np.zeros((R, n_words * 64), dtype=bool)
    padded[:, :n_bits] = masks
    bits = torch.from_numpy(np.ascontiguousarray(np.packbits(padded, axis=1, bitorder="little").view(np.int64))).to(DEV)
    off, idx = engine.bitset_csr(bits)
    torch.cuda.synchronize()
    off, idx = off.cpu().numpy(), idx.cpu().numpy()
    assert off.shape == (R + 1,) and off[0] == 0 and off[-1] == masks.sum() == len(idx)
    for r in range(R):
        assert np.array_equal(idx[off[r]:off[r + 1]], np.nonzero(masks[r])[0])
    t = engine.bits_transpose(bits)
    torch.cuda.synchronize()
    assert tuple(t.shape) == (n_words * 64, (R + 63) // 64)
    tb = np.unpackbits(t.cpu().numpy().view(np.uint8), axis=1, bitorder="little")
    assert np.array_equal(tb[:n_bits, :R].astype(bool), masks.T) and not tb[:, R:].any() and not tb[n_bits:].any()
    off2, idx2 = engine.bitset_csr(t[:n_bits].contiguous())
    off2, idx2 = off2.cpu().numpy(), idx2.cpu().numpy()
    for v in rng.choice(n_bits, size=min(n_bits, 200), replace=False):
        assert np.array_equal(idx2[off2[v]:off2[v + 1]], np.nonzero(masks[:, v])[0])


def test_visibility_index_from_csr_equals_reference():
    """MVI.process_scene through the device-side compaction: the reference's frozen index (tests/golden) and the arrow
    table the split writer streams (same rows as the reference's pkl -> parquet conversion)."""
    from mspa.scene import SceneOnDevice
    from spatial_engine.utils.scannet_utils.make_visibility_info import visibility_dict_to_frame
    g = GoldenScene("scene_ident")
    scene = SceneOnDevice(g.K, g.A, g.E, g.depth, g.color_hw, g.points, DEV)
    ref = g.json("mvi_json")
    csr = scene.visibility_csr()
    got = csr.to_dict()
    assert got["image_to_points"] == ref["image_to_points"]
    assert {str(k): v for k, v in got["point_to_images"].items()} == ref["point_to_images"]
    frame = visibility_dict_to_frame({"scene_x": got})
    table = csr.to_arrow("scene_x").to_pandas()
    assert table["key"].tolist() == frame["key"].tolist() and table["values"].tolist() == frame["values"].tolist()


# ------------------------------------------------------------------------------------------
# size-independent properties at BASELINE.json's full configuration (config 2: 1k 640x480 pairs)
# ------------------------------------------------------------------------------------------
def test_full_size_properties():
    sc = synth.make_scene(1006, n_points=64, n_frames=12, color_hw=(480, 640), invalid_pose_frac=0, with_color=False)
    ids = sc.valid_image_ids
    depth, mats, _ = upload_scene(sc, ids)
    rng = np.random.default_rng(0)
    n_pairs = 1000
    pair_np = rng.integers(0, len(ids), size=(n_pairs, 2)).astype(np.int32)
    pair_np[:len(ids)] = np.arange(len(ids))[:, None]            # identity pairs first
    pairs = torch.from_numpy(pair_np).to(DEV)
    P = 480 * 640
    out = engine.alloc_pair_outputs(n_pairs, sc.color_hw, ("vis_bits", "pix_i16", "counts", "valid_u8"), DEV)
    engine.pair_reproject(depth, mats, pairs, sc.color_hw, out)
    # the fast path must reproduce every integer product of the exact kernel, pixel for pixel
    outf = engine.alloc_pair_outputs(n_pairs, sc.color_hw, ("vis_bits", "pix_i16", "counts"), DEV)
    engine.pair_reproject(depth, mats, pairs, sc.color_hw, outf, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    for k in outf:
        assert torch.equal(out[k], outf[k]), f"fast path differs from exact kernel in {k}"
    # the streaming hint changes cache policy only
    outs = engine.alloc_pair_outputs(n_pairs, sc.color_hw, ("vis_bits", "pix_i16", "counts"), DEV)
    engine.pair_reproject(depth, mats, pairs, sc.color_hw, outs, flags=_lib.PAIR_FAST | _lib.PAIR_STREAM)
    torch.cuda.synchronize()
    for k in outs:
        assert torch.equal(out[k], outs[k]), f"MSPA_PAIR_STREAM changed {k}"
    # shard the same batch in two launches: results must not depend on batching / block mapping
    out2 = engine.alloc_pair_outputs(n_pairs, sc.color_hw, ("vis_bits", "counts"), DEV)
    h = 373
    o_a = {k: v[:h] for k, v in out2.items()}
    o_b = {k: v[h:] for k, v in out2.items()}
    engine.pair_reproject(depth, mats, pairs[:h].contiguous(), sc.color_hw, o_a)
    engine.pair_reproject(depth, mats, pairs[h:].contiguous(), sc.color_hw, o_b)
    torch.cuda.synchronize()
    assert torch.equal(out["vis_bits"], out2["vis_bits"]) and torch.equal(out["counts"], out2["counts"])
    counts = out["counts"].cpu().numpy()
    bits = out["vis_bits"].cpu().numpy()
    pop = np.unpackbits(bits.view(np.uint8), axis=1).sum(axis=1)
    assert np.array_equal(pop, counts[:, 1])                      # checksum of the bitset == counter
    assert (counts[:, 1] <= counts[:, 0]).all() and (counts[:, 0] <= P).all()
    valid = out["valid_u8"].cpu().numpy()
    assert np.array_equal(valid.sum(axis=1), counts[:, 0])
    # valid count only depends on frame 1
    zero_frac = np.array([(sc.depth[i] == 0).mean() for i in ids])
    assert np.allclose(1 - counts[:, 0] / P, zero_frac[pair_np[:, 0]], atol=1e-12)
    # identity pairs: every valid pixel reprojects onto itself
    pix = out["pix_i16"][:len(ids)].cpu().numpy()
    my, mx = np.divmod(np.arange(P), 640)
    for k in range(len(ids)):
        v = valid[k].astype(bool) & (pix[k][:, 0] >= 0)      # border pixels may round to just outside the image
        assert np.array_equal(pix[k][v, 0], mx[v]) and np.array_equal(pix[k][v, 1], my[v])
        interior = valid[k].astype(bool) & (mx > 0) & (mx < 639) & (my > 0) & (my < 479)
        assert (pix[k][interior, 0] >= 0).all()
    # spot-check three random pairs of the big batch against the oracle
    for p in (len(ids) + 1, 500, 999):
        a, b = ids[pair_np[p, 0]], ids[pair_np[p, 1]]
        ref = C.frame_pair(sc.depth[a], sc.depth[b], sc.K, sc.E[a], sc.E[b], sc.A, sc.color_hw)
        assert np.array_equal(unpack_bits(bits[p], P), ref["vis"])
        assert tuple(counts[p]) == (ref["n_valid"], ref["n_vis"])


# ------------------------------------------------------------------------------------------
# K4 pair pose / K5 track geometry
# ------------------------------------------------------------------------------------------
def f64_ok(a, b):
    return same_f64(a, b) or close_f64(a, b, rtol=1e-12, scale=1e-6)


@pytest.mark.parametrize("name", ["scene_ident", "scene_scaled"])
def test_pair_pose_golden(name):
    """K4 against the reference's CFR pair table and CME relative-pose answers (tests/golden)."""
    g = GoldenScene(name)
    ids = g.valid_image_ids
    Ea = [g.A @ g.E[i] for i in ids]
    yaw, pitch = engine.extract_yaw_pitch_host(Ea)
    E_t = torch.from_numpy(np.stack(Ea).reshape(-1, 16)).to(DEV)
    Einv_t = torch.from_numpy(np.stack([np.linalg.inv(e) for e in Ea]).reshape(-1, 16)).to(DEV)
    pairs = engine.all_pairs(len(ids), DEV)
    both = torch.cat([pairs, pairs.flip(1)], dim=0).contiguous()         # (i, j) and the swapped (j, i)
    out = engine.pair_pose(E_t, Einv_t, torch.from_numpy(yaw).to(DEV), torch.from_numpy(pitch).to(DEV), both)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    n = pairs.shape[0]
    ref = g["cfr_values"]
    assert f64_ok(out[:n, 0], ref[:, 1])                                   # distance  CFR:183
    assert same_f64(out[:n, 1], ref[:, 2]) and same_f64(out[:n, 2], ref[:, 3])   # yaw / pitch differences
    for p, (swap, ans) in enumerate(zip(g["cme_swap"], g["cme_answers_json"])):
        a = json.loads(str(ans))
        d = out[p + n, 3:6] if swap else out[p, 3:6]
        assert f64_ok(d, a["displacement_vector"])
        assert [int(v * 1000) for v in d] == [a["x_value"], a["y_value"], a["z_value"]]      # CME:221-223
        assert int(np.linalg.norm(d) * 1000) == a["total_distance"]


def test_extract_yaw_pitch_device_vs_reference_values():
    """mspa_extract_yaw_pitch against the reference's per-frame angles (engine.extract_yaw_pitch_host is its NumPy restatement,
    pinned through the pair table's yaw / pitch columns in test_pair_pose_golden): float64 quantity, 1e-12 absolute in degrees
    (bar 1e-5 relative); degenerate axes included."""
    g = GoldenScene("scene_ident")
    Ea = [g.A @ g.E[i] for i in g.valid_image_ids]
    rng = np.random.default_rng(8)
    for _ in range(200):
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        e = np.eye(4)
        e[:3, :3] = q
        Ea.append(e)
    up = np.eye(4)
    up[:3, :3] = [[1, 0, 0], [0, 0, -1], [0, 1, 0]]           # camera z axis (third column) along -y: yaw exactly -90
    Ea.append(up)
    yaw_h, pitch_h = engine.extract_yaw_pitch_host(Ea)
    E_t = torch.from_numpy(np.stack(Ea).reshape(-1, 16)).to(DEV)
    yaw_d, pitch_d = engine.extract_yaw_pitch(E_t)
    torch.cuda.synchronize()
    assert np.abs(yaw_d.cpu().numpy() - yaw_h).max() < 1e-12 and np.abs(pitch_d.cpu().numpy() - pitch_h).max() < 1e-12
    assert abs(float(yaw_d[-1]) + 90.0) < 1e-12 and abs(float(pitch_d[-1])) < 1e-12


def test_track_geometry_golden():
    """K5 against the reference's TAPVid records (tests/golden/tracks.npz) and the oracle."""
    import os
    from golden_util import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, "tracks.npz"))
    T, P, _ = z["tracks_XYZ"].shape
    hw = tuple(int(v) for v in z["image_hw"])
    tracks = torch.from_numpy(np.ascontiguousarray(z["tracks_XYZ"])).to(DEV)
    w2c_np = np.ascontiguousarray(z["extrinsics_w2c"])
    c2w_np = np.linalg.inv(w2c_np)                                          # OM_C:448, on the host
    c2w = torch.from_numpy(c2w_np.reshape(T, 16)).to(DEV)
    w2c = torch.from_numpy(w2c_np.reshape(T, 16)).to(DEV)
    res = engine.track_to_world(tracks, c2w, z["fx_fy_cx_cy"], hw)
    trip_np = np.ascontiguousarray(z["pairs"].astype(np.int32))
    disp, flags = engine.track_displacement(res["world"], w2c, c2w, torch.from_numpy(trip_np).to(DEV))
    torch.cuda.synchronize()
    world = res["world"].cpu().numpy()
    assert f64_ok(world, z["ref_world"])
    uvn, ok = res["uvn"].cpu().numpy(), res["ok"].cpu().numpy().astype(bool)
    for t in range(T):
        for p in range(0, P, 5):
            r = O.project_point(z["tracks_XYZ"][t, p], z["fx_fy_cx_cy"], hw[0], hw[1])
            assert (r is not None) == bool(ok[t, p])
            if r is not None:
                assert same_f64(uvn[t, p], r)
    disp, flags = disp.cpu().numpy(), flags.cpu().numpy()
    for k, ((f1, f2, p), kept, rec) in enumerate(zip(z["pairs"], z["kept"], z["records_json"])):
        assert bool(kept) == bool(ok[f1, p] and ok[f2, p])                   # OM_C:360-362 skip rule
        o = O.object_displacement(z["ref_world"], z["tracks_XYZ"], z["extrinsics_w2c"], z["fx_fy_cx_cy"], hw,
                                  int(f1), int(f2), int(p))
        dref = np.linalg.norm(z["ref_world"][f2, p] - z["ref_world"][f1, p])
        assert f64_ok(disp[k, 4], np.linalg.norm((z["ref_world"][[f2], p] - z["ref_world"][[f1], p]), axis=1)[0])
        assert flags[k, 0] == int(not (dref < 0.01))
        if not kept:
            continue
        ref = json.loads(str(rec))
        assert flags[k, 0] == ref["point_moving"] == o["point_moving"]
        assert flags[k, 1] == ref["cam_moving"] == o["cam_moving"]
        assert f64_ok(disp[k, 1:4], ref["gt_value"])                          # vector stored in metres (OM_C:393)
        assert int(disp[k, 0] * 1000) == o["gt_total_distance"]
        p1 = (round(uvn[f1, p, 0] * 1000), round(uvn[f1, p, 1] * 1000))       # OM_C:364-365
        p2 = (round(uvn[f2, p, 0] * 1000), round(uvn[f2, p, 1] * 1000))
        assert p1 == tuple(ref["p1"]) and p2 == tuple(ref["p2"])


# ------------------------------------------------------------------------------------------
# fast == exact on adversarially varied geometry (culling, early-out, guard, cold loop)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("hw,dhw", [((96, 128), None), ((48, 64), None), ((61, 83), None), ((61, 83), (48, 64))],
                         ids=["tight96x128", "tight48x64", "ragged61x83", "ragged61x83_over_depth48x64"])
def test_fast_equals_exact_random_poses(hw, dhw):
    """300 random camera pairs per shape -- looking away, nearly coincident, grazing, very close, with
    pure translations that produce exact half-pixel ties -- must give bit-identical integer outputs from
    the fast kernels (tile culling, group early-out, guard band, cold loop) and the exact kernel."""
    H, W = hw
    DH, DW = dhw or hw                                   # depth grid; colour grid (H, W) is scaled onto it when they differ
    rng = np.random.default_rng(H * 1000 + W + DH + int(os.environ.get("MSPA_STRESS_SEED", "0")))   # other seeds: one-off stress runs
    n_frames = 40

    def look_at(eye, tgt):
        fwd = tgt - eye
        fwd = fwd / np.linalg.norm(fwd)
        helper = np.array([0.0, 1.0, 0.0]) if abs(fwd[1]) < 0.9 else np.array([1.0, 0.0, 0.0])
        right = np.cross(helper, fwd)
        right /= np.linalg.norm(right)
        E = np.eye(4)
        E[:3, 0], E[:3, 1], E[:3, 2], E[:3, 3] = right, np.cross(fwd, right), fwd, eye
        return E

    K = np.eye(4)
    K[0, 0] = K[1, 1] = 64.0 if H == 48 else float(rng.uniform(0.6, 1.4) * W)
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    A = np.eye(4)
    A[:3, 3] = [0.5, -0.25, 0.125]
    E_list, depth = [], []
    for f in range(n_frames):
        kind = f % 5
        if kind == 0:                                   # generic look-at
            eye, tgt = rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 3) + [0, 0, 4]
        elif kind == 1:                                 # looking the other way: everything behind / outside
            eye, tgt = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3) - [0, 0, 4]
        elif kind == 2:                                 # nearly coincident with frame 0
            eye, tgt = np.array([0.0, 0.0, 0.0]) + rng.normal(0, 1e-3, 3), np.array([0.0, 0.0, 4.0])
        elif kind == 3:                                 # pure dyadic translation of frame 0: exact ties
            eye, tgt = np.array([1 / 64.0, -1 / 32.0, 0.0]), np.array([1 / 64.0, -1 / 32.0, 4.0])
        else:                                           # sideways / grazing
            eye, tgt = rng.uniform(-3, 3, 3), rng.uniform(-3, 3, 3)
        if f == 0:
            eye, tgt = np.zeros(3), np.array([0.0, 0.0, 4.0])
        E = look_at(np.asarray(eye, float), np.asarray(tgt, float))
        if kind != 3 and f != 0:
            E = synth._roundtrip_f(E)
        E_list.append(np.linalg.inv(A) @ E)
        base = rng.choice([1000, 2000, 4000]) if kind in (0, 3) or f == 0 else int(rng.integers(300, 6000))
        d = np.full((DH, DW), base, dtype=np.int64)
        if kind != 3 and f != 0:
            d = d + rng.integers(-40, 41, (DH, DW))
        if kind == 4:
            d = d + np.add.outer(np.arange(DH), np.arange(DW)) // 7 * int(rng.integers(0, 30))
        d = np.clip(d, 1, 65535).astype(np.uint16)
        d[rng.random((DH, DW)) < 0.05] = 0
        if f % 7 == 6:
            d[:] = 0                                    # a frame without any valid depth
        depth.append(d)
    pairs_np = rng.integers(0, n_frames, (300, 2)).astype(np.int32)
    pairs_np[:n_frames, 0] = 0                          # frame 0 against every frame, and back
    pairs_np[:n_frames, 1] = np.arange(n_frames)
    pairs_np[n_frames:2 * n_frames, 0] = np.arange(n_frames)
    pairs_np[n_frames:2 * n_frames, 1] = 0
    dep = engine.depth_to_device(np.stack(depth), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E_list)).to(DEV)
    pairs = torch.from_numpy(pairs_np).to(DEV)
    outs = ("vis_bits", "pix_i16", "counts")
    ex = engine.alloc_pair_outputs(len(pairs_np), hw, outs, DEV)
    fa = engine.alloc_pair_outputs(len(pairs_np), hw, outs, DEV)
    engine.pair_reproject(dep, mats, pairs, hw, ex)
    engine.pair_reproject(dep, mats, pairs, hw, fa, flags=_lib.PAIR_FAST)
    mn = engine.alloc_pair_outputs(len(pairs_np), hw, ("vis_bits", "counts"), DEV)
    engine.pair_reproject(dep, mats, pairs, hw, mn, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    for k in outs:
        bad = (ex[k] != fa[k]).reshape(len(pairs_np), -1).any(dim=1).nonzero().flatten().tolist()
        assert not bad, f"{k}: fast != exact for pairs {[(int(pairs_np[b, 0]), int(pairs_np[b, 1])) for b in bad[:8]]}"
    assert torch.equal(mn["vis_bits"], ex["vis_bits"]) and torch.equal(mn["counts"], ex["counts"])
    # byte-mask output set: the stripe-mapped general fast kernel on ragged shapes (a bitset request there takes the
    # linear pixel mapping, exercised above)
    outs_b = ("vis_u8", "valid_u8", "pix_i16", "counts")
    exb = engine.alloc_pair_outputs(len(pairs_np), hw, outs_b, DEV)
    fab = engine.alloc_pair_outputs(len(pairs_np), hw, outs_b, DEV)
    engine.pair_reproject(dep, mats, pairs, hw, exb)
    engine.pair_reproject(dep, mats, pairs, hw, fab, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    for k in outs_b:
        assert torch.equal(exb[k], fab[k]), f"{k}: general fast kernel != exact kernel"
    assert torch.equal(exb["counts"], ex["counts"])
    counts = ex["counts"].cpu().numpy()
    assert (counts[:, 1] > 0).sum() > 20 and (counts[:, 1] == 0).sum() > 20      # both regimes exercised
    # anchor the exact kernel itself on a few pairs
    for p in (1, n_frames + 3, 150, 299):
        a, b = pairs_np[p]
        ref = C.frame_pair(depth[a], depth[b], K, E_list[a], E_list[b], A, hw)
        assert tuple(counts[p]) == (ref["n_valid"], ref["n_vis"])
        assert np.array_equal(unpack_bits(ex["vis_bits"][p].cpu().numpy(), H * W), ref["vis"])


# ------------------------------------------------------------------------------------------
# K7 rigidity loss / K8 object extents against the oracle at sizes beyond the golden fixtures
# ------------------------------------------------------------------------------------------
def test_rigidity_loss_vs_oracle():
    tr = synth.make_tracks(71, T=120, P=200, n_groups=6)
    loss = engine.track_rigidity_loss(torch.from_numpy(np.ascontiguousarray(tr.tracks_XYZ)).to(DEV), 0.01).cpu().numpy()
    want = O.rigidity_loss(tr.tracks_XYZ, 0.01)
    assert np.array_equal(loss, want)                       # same operation order, correctly rounded sqrt
    assert (loss > 0).any() and np.array_equal(loss, loss.T)
    empty = engine.track_rigidity_loss(torch.zeros((5, 0, 3), dtype=torch.float64, device=DEV))
    assert empty.shape == (0, 0)
    one = engine.track_rigidity_loss(torch.randn((1, 7, 3), dtype=torch.float64, device=DEV)).cpu().numpy()
    assert (one == 0).all()                                 # a single frame has no change to accumulate


def test_object_extents_vs_oracle_large():
    from mspa import coverage
    from mspa.scene import pack_index_lists
    rng = np.random.default_rng(12)
    V, F, n_obj = 20000, 70, 12
    pts = rng.normal(0, 2, (V, 3))
    label = rng.integers(-1, n_obj, V)                      # -1 = unlabelled
    objs = {o: np.where(label == o)[0] for o in range(n_obj)}
    objs[3] = objs[3][:1]                                   # a one-vertex object
    objs[5] = rng.permutation(objs[5])                      # vertex list in arbitrary order
    lists = [np.where(rng.random(V) < rng.uniform(0.0, 0.3))[0] for _ in range(F)]
    lists[4] = np.zeros(0, dtype=np.int64)                  # an image that sees nothing
    ids = [f"{k:05d}" for k in range(F)]
    bits = torch.from_numpy(pack_index_lists(lists, V)).to(DEV)
    ext = coverage.scene_extents(bits, ids, pts, objs)
    seen = np.zeros((F, V), dtype=bool)
    for f, l in enumerate(lists):
        seen[f, l] = True
    for o, idx in objs.items():
        k = ext.object_index[o]
        m = np.zeros(V, dtype=bool)
        m[idx] = True
        for f in range(F):
            both = seen[f] & m
            assert ext.count[k, f] == both.sum()
            for axis in range(3):
                cov = O.compute_coverage(pts, both, axis)
                if cov is None:
                    assert ext.lo[k, f, axis] == np.inf and ext.hi[k, f, axis] == -np.inf
                else:
                    assert ext.hi[k, f, axis] - ext.lo[k, f, axis] == cov


def test_pair_pose_256_pairs_golden():
    """K4 + the answer fields of the camera-movement head against the reference's 256 answer_values (cme256.npz)."""
    import os
    from golden_util import GOLDEN_DIR
    from mspa import heads
    z = np.load(os.path.join(GOLDEN_DIR, "cme256.npz"))
    Ea = [z["A"] @ e for e in z["E"]]
    E_t = torch.from_numpy(np.stack(Ea).reshape(-1, 16)).to(DEV)
    Einv_t = torch.from_numpy(np.linalg.inv(np.stack(Ea)).reshape(-1, 16)).to(DEV)
    zeros = torch.zeros(len(Ea), dtype=torch.float64, device=DEV)
    idx = z["rows"][:, :2].astype(np.int32)
    both = torch.from_numpy(np.concatenate([idx, idx[:, ::-1]], axis=0).copy()).to(DEV)
    out = engine.pair_pose(E_t, Einv_t, zeros, zeros, both).cpu().numpy()
    n = len(idx)
    assert f64_ok(out[:n, 0], z["rows"][:, 4])                       # the pair-table distance of every row
    for k, ((a, b, yaw, pitch, dist), swap, ans) in enumerate(zip(z["rows"], z["swap"], z["answers_json"])):
        ref = json.loads(str(ans))
        yaw_angle, pitch_angle = (-yaw, -pitch) if swap else (yaw, pitch)
        if abs(yaw_angle) > 180:
            yaw_angle = yaw_angle - 360 if yaw_angle > 0 else yaw_angle + 360
        got = heads.camera_movement_answer_values(out[n + k, 3:6] if swap else out[k, 3:6], yaw_angle, pitch_angle)
        dv_ref, dv_got = ref.pop("displacement_vector"), got.pop("displacement_vector")
        assert got == ref, k
        assert f64_ok(dv_got, dv_ref)
