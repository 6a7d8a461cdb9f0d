"""The tight kernel on rectangular tiles of ANY colour / depth grid pair (MSPA_KERNEL_PAIR_FAST_RECT, round 4): ScanNet's own
shape -- 1296x968 colour over 640x480 depth -- gets the FUSED compacted correspondence output (no dense table:
``mspa_pair_correspondences_workspace_bytes(..., 968, 1296, FAST) == 0``), and every shape with W % 16 == 0, H % 4 == 0,
dw % 4 == 0 the correspondence / minimal / compacted sets.  Against the NumPy oracle (``np.nonzero(oracle vis)`` order for the
compacted set), against the exact kernel on adversarial poses incl. identity pairs (cold loop, in-place rewrite, rebuild), and
against the wobbling-stripe kernel the correspondence set takes by default at ScanNet's shape.

Shapes: a ragged right stripe (16 live columns) and a ragged bottom band (4 / 8 rows); rows that are not whole bitset words
(W = 144, 1296: the 2-byte bitset stores) and rows that are (W = 128 over a 64-wide depth grid: the 8-byte store); equal
grids that are not whole tiles (100x144 over 100x144).
"""
import numpy as np
import pytest
import torch

import adversarial as ADV
from mspa import engine, synth, _lib
from oracle import np_oracle as O
from test_gpu_compact import check_pair as check_compact_pair, poisoned_outputs
from test_gpu_guard import check_integers
from test_gpu_tight import SETS, launch

DEV = "cuda"
SHAPES = {
    "144x100_over_72x48": ((100, 144), (48, 72)),
    "128x96_over_64x48": ((96, 128), (48, 64)),
    "144x100_equal_grids": ((100, 144), (100, 144)),
    "208x52_over_100x26": ((52, 208), (26, 100)),
}


def scene(hw, dhw, seed, frames=5):
    sc = synth.make_scene(seed, n_points=64, n_frames=frames, color_hw=hw, depth_hw=dhw, invalid_pose_frac=0.0, with_color=False,
                          trajectory="sweep", walk_step=0.08)
    ids = sc.valid_image_ids
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])).to(DEV)
    return sc, ids, depth, mats


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SHAPES))
def test_rect_kernel_vs_oracle_small_shapes(name):
    hw, dhw = SHAPES[name]
    sc, ids, depth, mats = scene(hw, dhw, 3030)
    pair_idx = [(0, 1), (1, 0), (0, 4), (3, 3), (4, 2), (2, 1)]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    refs = [O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, hw) for a, b in pair_idx]
    assert sum(r["n_vis"] for r in refs) > 500
    lib = _lib.load()
    assert lib.mspa_pair_correspondences_workspace_bytes(len(pair_idx), dhw[0], dhw[1], hw[0], hw[1], _lib.PAIR_FAST) == 0
    for stream in (0, _lib.PAIR_STREAM):
        for sname in ("corr", "minimal"):
            res, kern = launch(depth, mats, None, pairs, hw, SETS[sname], _lib.PAIR_FAST | stream)
            assert kern == _lib.KERNEL_PAIR_FAST_RECT
            for n, ref in enumerate(refs):
                check_integers(res, n, ref, hw)
        out = poisoned_outputs(len(pair_idx), hw)
        engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST | stream)
        assert lib.mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_RECT
        torch.cuda.synchronize()
        out_np = {k: v.cpu().numpy() for k, v in out.items()}
        for n, ref in enumerate(refs):
            check_compact_pair(out_np, n, ref, hw)
            i, xi, yi = engine.correspondences_rowmajor(out, hw, n)
            nz = np.nonzero(ref["vis"])[0]
            assert np.array_equal(i.cpu().numpy(), nz) and np.array_equal(xi.cpu().numpy(), ref["xi"][nz])
            assert np.array_equal(yi.cpu().numpy(), ref["yi"][nz])
        # nothing written behind a tile's count (the fused kernel stores 4 bytes per visible pixel and nothing else)
        # (not for the identity pair: there every depth test is a tie, and a tile in which the reference chain takes a pixel
        # OUT of the visible set may keep a stale entry behind its count -- include/mspa.h)
        ordinary = [n for n, (a_, b_) in enumerate(pair_idx) if a_ != b_]
        tc = out_np["tile_counts"][ordinary]
        keep = np.arange(out_np["cpix"].shape[2])[None, None, :] >= tc[:, :, None]
        assert (out_np["cpix"][ordinary][keep] == -7).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["144x100_over_72x48", "128x96_over_64x48"])
def test_rect_kernel_equals_exact_on_adversarial_poses(name):
    """160 adversarial pairs incl. identity pairs (every depth test a tie: cold loop, in-place rewrite and the rebuild of the
    compacted segments) -- every integer of corr / minimal / compact equals the exact kernel's (+ stand-alone compaction)."""
    hw, dhw = SHAPES[name]
    rng = np.random.default_rng(78)
    K, A, E = ADV.adversarial_pairs(rng, 16, hw)
    Kd = ADV.depth_intrinsics(K, hw, dhw)
    boxes = synth._make_boxes(rng)
    depth_np = [ADV.render_mm(A @ e, Kd, dhw, boxes, rng) for e in E]
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    pair_np = np.stack([rng.integers(0, len(E), 160), rng.integers(0, len(E), 160)], 1).astype(np.int32)
    pair_np[:16] = np.arange(16)[:, None]
    pairs = torch.from_numpy(pair_np).to(DEV)
    for sname in ("corr", "minimal"):
        fast, kf = launch(depth, mats, None, pairs, hw, SETS[sname], _lib.PAIR_FAST)
        exact, ke = launch(depth, mats, None, pairs, hw, SETS[sname], 0)
        assert kf == _lib.KERNEL_PAIR_FAST_RECT and ke == _lib.KERNEL_PAIR_EXACT
        for k in SETS[sname]:
            assert np.array_equal(fast[k], exact[k]), f"{sname}: {k} differs from the exact kernel"
    fused = poisoned_outputs(len(pair_np), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, fused, flags=_lib.PAIR_FAST | _lib.PAIR_STREAM)
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_RECT
    route = poisoned_outputs(len(pair_np), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, route, flags=0)
    torch.cuda.synchronize()
    for k in ("vis_bits", "counts", "tile_counts"):
        assert torch.equal(fused[k], route[k]), k
    tc = route["tile_counts"].cpu().numpy()
    f, e = fused["cpix"].cpu().numpy(), route["cpix"].cpu().numpy()
    keep = np.arange(f.shape[2])[None, None, :] < tc[:, :, None]
    assert np.array_equal(f[keep], e[keep])
    assert int(route["counts"][:16, 1].sum()) > 500, "identity pairs must see something"


@pytest.mark.gpu
def test_rect_kernel_at_scannet_shape():
    """ScanNet's own shape: the fused compacted set needs no workspace and equals np.nonzero(oracle vis) order; the
    rectangular-tile kernel's correspondence / minimal sets equal the oracle's AND the wobbling-stripe
    kernel's; 30 adversarial pairs against the exact kernel."""
    hw, dhw = (968, 1296), (480, 640)
    lib = _lib.load()
    assert lib.mspa_pair_correspondences_workspace_bytes(1000, 480, 640, 968, 1296, _lib.PAIR_FAST) == 0
    assert lib.mspa_pair_correspondences_workspace_bytes(2, 480, 640, 968, 1296, 0) == 2 * 968 * 1296 * 4
    sc, ids, depth, mats = scene(hw, dhw, 3031, frames=4)
    pair_idx = [(0, 1), (2, 0), (3, 3)]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    refs = [O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, hw) for a, b in pair_idx]
    assert sum(r["n_vis"] for r in refs) > 100000
    out = poisoned_outputs(len(pair_idx), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST | _lib.PAIR_STREAM)   # workspace=None
    assert lib.mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_RECT
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    for n, ref in enumerate(refs):
        check_compact_pair(out_np, n, ref, hw)
        i, xi, yi = engine.correspondences_rowmajor(out, hw, n)
        nz = np.nonzero(ref["vis"])[0]
        assert np.array_equal(i.cpu().numpy(), nz) and np.array_equal(xi.cpu().numpy(), ref["xi"][nz])
        assert np.array_equal(yi.cpu().numpy(), ref["yi"][nz])
    for sname in ("corr", "minimal"):
        wob, kw = launch(depth, mats, None, pairs, hw, SETS[sname], _lib.PAIR_FAST | _lib.PAIR_WORD_STRIPES)
        rect, kr = launch(depth, mats, None, pairs, hw, SETS[sname], _lib.PAIR_FAST)
        assert kw == _lib.KERNEL_PAIR_FAST_SCALED and kr == _lib.KERNEL_PAIR_FAST_RECT
        for k in SETS[sname]:
            assert np.array_equal(wob[k], rect[k]), k
        for n, ref in enumerate(refs):
            check_integers(rect, n, ref, hw)
    # adversarial poses against the exact kernel
    rng = np.random.default_rng(79)
    K, A, E = ADV.adversarial_pairs(rng, 6, hw)
    Kd = ADV.depth_intrinsics(K, hw, dhw)
    boxes = synth._make_boxes(rng)
    depth_np = [ADV.render_mm(A @ e, Kd, dhw, boxes, rng) for e in E]
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    pair_np = np.stack([rng.integers(0, 6, 30), rng.integers(0, 6, 30)], 1).astype(np.int32)
    pair_np[:6] = np.arange(6)[:, None]
    pairs = torch.from_numpy(pair_np).to(DEV)
    fused = poisoned_outputs(30, hw)
    engine.pair_correspondences(depth, mats, pairs, hw, fused, flags=_lib.PAIR_FAST)
    assert lib.mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_RECT
    route = poisoned_outputs(30, hw)
    engine.pair_correspondences(depth, mats, pairs, hw, route, flags=0)
    torch.cuda.synchronize()
    for k in ("vis_bits", "counts", "tile_counts"):
        assert torch.equal(fused[k], route[k]), k
    tc = route["tile_counts"]
    keep = torch.arange(fused["cpix"].shape[2], device=DEV)[None, None, :] < tc[:, :, None]
    assert torch.equal(fused["cpix"][keep], route["cpix"][keep])


@pytest.mark.gpu
def test_rect_kernel_near_camera2_plane():
    """The guard regime of tests/test_gpu_guard.py on a rectangular-tile shape: camera 2 centred 1e-9 .. 1e-4 m behind
    frame-1 points."""
    hw, dhw = SHAPES["144x100_over_72x48"]
    rng = np.random.default_rng(80)
    K, A, E, depth_np, pair_idx = ADV.near_plane_case(rng, hw, (1e-4, 1e-6, 1e-7, 1e-9), per_delta=5, dhw=dhw)
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    refs = [O.frame_pair(depth_np[a], depth_np[b], K, E[a], E[b], A, hw) for a, b in pair_idx]
    for sname in ("corr", "minimal"):
        res, kern = launch(depth, mats, None, pairs, hw, SETS[sname], _lib.PAIR_FAST)
        assert kern == _lib.KERNEL_PAIR_FAST_RECT
        for n, ref in enumerate(refs):
            check_integers(res, n, ref, hw)
    out = poisoned_outputs(len(pair_idx), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    for n, ref in enumerate(refs):
        check_compact_pair(out_np, n, ref, hw)


@pytest.mark.gpu
def test_depth_grid_larger_than_colour_takes_the_reference_order_kernel():
    """A depth grid finer than the colour grid (sx, sy > 1; no dataset has one): the fast kernels' guard band is only argued for
    sx, sy <= 1, so MSPA_PAIR_FAST is ignored there -- the exact kernel runs, the compacted set goes through the dense table --
    and the integers match the oracle."""
    hw, dhw = (48, 64), (96, 128)
    sc, ids, depth, mats = scene(hw, dhw, 3031)
    pair_idx = [(0, 1), (1, 0), (2, 2), (4, 0)]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    refs = [O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, hw) for a, b in pair_idx]
    assert sum(r["n_vis"] for r in refs) > 200
    lib = _lib.load()
    assert lib.mspa_pair_correspondences_workspace_bytes(len(pair_idx), dhw[0], dhw[1], hw[0], hw[1], _lib.PAIR_FAST) > 0
    for sname in ("corr", "minimal"):
        res, kern = launch(depth, mats, None, pairs, hw, SETS[sname], _lib.PAIR_FAST)
        assert kern == _lib.KERNEL_PAIR_EXACT
        for n, ref in enumerate(refs):
            check_integers(res, n, ref, hw)
    out = poisoned_outputs(len(pair_idx), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    for n, ref in enumerate(refs):
        check_compact_pair(out_np, n, ref, hw)


@pytest.mark.gpu
def test_unfilled_or_nan_bound_slots_lose_speed_not_correctness():
    """A C caller that moves to the 8-slot frame records but never calls mspa_frame_bounds_host leaves slot MSPA_MAT_BOUNDS
    zeroed; a NaN may sit there as well.  The guard then trusts nothing (every lane takes the exact chain) and nothing is
    culled: integers still equal the oracle, on the whole-tile kernel and on the rectangular one."""
    for hw, dhw, want in (((96, 128), (96, 128), _lib.KERNEL_PAIR_FAST_TIGHT), ((100, 144), (48, 72), _lib.KERNEL_PAIR_FAST_RECT)):
        sc, ids, depth, mats = scene(hw, dhw, 3032)
        pair_idx = [(0, 1), (1, 0), (0, 4), (3, 3)]
        pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
        refs = [O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, hw) for a, b in pair_idx]
        for fill in (0.0, float("nan")):
            bad = mats.clone()
            bad[:, _lib.MAT_BOUNDS, :] = fill
            for sname in ("corr", "minimal"):
                res, kern = launch(depth, bad, None, pairs, hw, SETS[sname], _lib.PAIR_FAST)
                assert kern == want
                for n, ref in enumerate(refs):
                    check_integers(res, n, ref, hw)
            out = poisoned_outputs(len(pair_idx), hw)
            engine.pair_correspondences(depth, bad, pairs, hw, out, flags=_lib.PAIR_FAST)
            torch.cuda.synchronize()
            out_np = {k: v.cpu().numpy() for k, v in out.items()}
            for n, ref in enumerate(refs):
                check_compact_pair(out_np, n, ref, hw)
