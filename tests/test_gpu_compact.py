"""Compacted correspondence output (include/mspa.h, mspa_pair_correspondences): visibility bitset + per 64 x 48 tile the
(xi, yi) of its visible pixels in (row, column) order.

  * the fused tight kernel (MSPA_PAIR_FAST, whole-tile shapes) against the NumPy oracle, entry by entry, at 96x128 and at the
    BASELINE shape 640x480, with and without the streaming hint -- asserting the tight kernel is what ran;
  * against the exact kernel + the stand-alone compaction step on 240 adversarial pairs incl. identity pairs (every pixel of
    those sits on a guard boundary: the fused kernel's rewrite path);
  * the dense-table route (ragged tiles, colour grid over a smaller depth grid, reference-order mode) against the oracle;
  * the flat np.nonzero-order view the host layer offers.
"""
import numpy as np
import pytest
import torch

from mspa import engine, synth, _lib
from oracle import np_oracle as O
from test_gpu_tight import adversarial_pairs, unpack_bits

DEV = "cuda"
TW, TH, CAP = 64, 48, 64 * 48


def expected_segments(vis, xi, yi, hw):
    """The oracle's visibility mask and pixel indices cut into the tile segments of the output format."""
    H, W = hw
    ns, nb = (W + TW - 1) // TW, (H + TH - 1) // TH
    v2, x2, y2 = vis.reshape(H, W), xi.reshape(H, W), yi.reshape(H, W)
    segs, counts = [], np.zeros(ns * nb, dtype=np.int32)
    for b in range(nb):
        for s in range(ns):
            m = v2[b * TH:(b + 1) * TH, s * TW:(s + 1) * TW]
            xs = x2[b * TH:(b + 1) * TH, s * TW:(s + 1) * TW][m]          # boolean indexing = (row, column) order
            ys = y2[b * TH:(b + 1) * TH, s * TW:(s + 1) * TW][m]
            segs.append(np.stack([xs, ys], 1).astype(np.int16))
            counts[b * ns + s] = m.sum()
    return segs, counts


def poisoned_outputs(n, hw):
    out = engine.alloc_pair_correspondences(n, hw, DEV)
    out["vis_bits"].fill_(0x5A5A5A5A5A5A5A5A)
    out["cpix"].fill_(-7)
    out["tile_counts"].fill_(-3)
    out["counts"].fill_(-3)
    return out


def check_pair(out_np, n, ref, hw):
    P = hw[0] * hw[1]
    assert np.array_equal(unpack_bits(out_np["vis_bits"][n], P), ref["vis"])
    assert tuple(out_np["counts"][n]) == (ref["n_valid"], ref["n_vis"])
    segs, counts = expected_segments(ref["vis"], ref["xi"], ref["yi"], hw)
    assert np.array_equal(out_np["tile_counts"][n], counts)
    for t, seg in enumerate(segs):
        assert np.array_equal(out_np["cpix"][n, t, :len(seg)], seg), f"pair {n} tile {t}"


def scene_inputs(hw, seed=1010, frames=5):
    sc = synth.make_scene(seed, n_points=64, n_frames=frames, color_hw=hw, depth_hw=hw, invalid_pose_frac=0.0, with_color=False,
                          trajectory="sweep", walk_step=0.08)
    ids = sc.valid_image_ids
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])).to(DEV)
    return sc, ids, depth, mats


@pytest.mark.gpu
@pytest.mark.parametrize("stream", [False, True], ids=["plain", "stream"])
@pytest.mark.parametrize("hw", [(96, 128), (480, 640)], ids=["96x128", "640x480"])
def test_fused_compact_vs_oracle(hw, stream):
    sc, ids, depth, mats = scene_inputs(hw)
    pair_idx = [(0, 1), (1, 0), (0, 4), (3, 3), (4, 2)] if hw[0] < 200 else [(0, 1), (4, 0), (2, 2)]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    out = poisoned_outputs(len(pair_idx), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0))
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_TIGHT, "the fused kernel must be what ran"
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    seen = 0
    for n, (a, b) in enumerate(pair_idx):
        ref = O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, hw)
        check_pair(out_np, n, ref, hw)
        seen += ref["n_vis"]
        # the flat view: np.nonzero order of the visibility mask
        i, xi, yi = engine.correspondences_rowmajor(out, hw, n)
        nz = np.nonzero(ref["vis"])[0]
        assert np.array_equal(i.cpu().numpy(), nz)
        assert np.array_equal(xi.cpu().numpy(), ref["xi"][nz]) and np.array_equal(yi.cpu().numpy(), ref["yi"][nz])
    assert seen > 0
    # entries past a tile's count are never written by the fused kernel (the traffic claim): the poison is still there
    cp, tc = out_np["cpix"], out_np["tile_counts"]
    for n, (a, b) in enumerate(pair_idx):
        if a == b:
            continue                      # identity pairs go through the rewrite path, which may leave a stale entry behind
        for t in range(cp.shape[1]):
            tail = cp[n, t, ((int(tc[n, t]) + 3) // 4) * 4:]               # the last 16-byte piece may be partly unspecified
            assert (tail == -7).all()


@pytest.mark.gpu
@pytest.mark.parametrize("stream", [False, True], ids=["plain", "stream"])
def test_fused_compact_equals_exact_route_on_adversarial_poses(stream):
    """240 adversarial pairs at 96x128: fused kernel == exact kernel + stand-alone compaction, every integer."""
    hw = (96, 128)
    rng = np.random.default_rng(77)
    K, A, E = adversarial_pairs(rng, 24, hw)
    boxes = synth._make_boxes(rng)
    depth_np = []
    for e in E:
        z = synth.render_depth(A @ e, K, hw, boxes)
        mm = np.clip(np.rint(z * 1000.0 + rng.normal(0, 4.0, z.shape)), 0, 65535).astype(np.uint16)
        mm[rng.random(mm.shape) < 0.07] = 0
        depth_np.append(mm)
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    pair_np = np.stack([rng.integers(0, len(E), 240), rng.integers(0, len(E), 240)], 1).astype(np.int32)
    pair_np[:24] = np.arange(24)[:, None]                 # identity pairs: everything lands on exact integers
    pairs = torch.from_numpy(pair_np).to(DEV)
    fused = poisoned_outputs(len(pair_np), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, fused, flags=_lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0))
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_TIGHT
    exact = poisoned_outputs(len(pair_np), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, exact, flags=0)       # dense table in a workspace + compaction
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_EXACT
    torch.cuda.synchronize()
    for k in ("vis_bits", "counts", "tile_counts"):
        assert torch.equal(fused[k], exact[k]), k
    tc = exact["tile_counts"].cpu().numpy()
    f, e = fused["cpix"].cpu().numpy(), exact["cpix"].cpu().numpy()
    for n in range(len(pair_np)):
        for t in range(tc.shape[1]):
            assert np.array_equal(f[n, t, :tc[n, t]], e[n, t, :tc[n, t]]), f"pair {n} tile {t}"
    assert int(exact["counts"][:, 1].sum()) > 1000
    assert int(exact["counts"][:24, 1].sum()) > 1000, "identity pairs must see something (they exercise the rewrite path)"


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ragged", "scaled", "ragged_exact"])
def test_dense_table_route_vs_oracle(case):
    """Shapes the fused kernel does not take: tiles ragged at both edges; a colour grid over a smaller depth grid; and the
    reference-order mode.  Same format, same integers."""
    hw, dhw = ((100, 150), (100, 150)) if case.startswith("ragged") else ((121, 162), (60, 80))
    sc = synth.make_scene(2020, n_points=64, n_frames=4, color_hw=hw, depth_hw=dhw, invalid_pose_frac=0.0, with_color=False,
                          trajectory="sweep", walk_step=0.08)
    ids = sc.valid_image_ids
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])).to(DEV)
    pair_idx = [(0, 1), (2, 0), (3, 3)]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    out = poisoned_outputs(len(pair_idx), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=0 if case == "ragged_exact" else _lib.PAIR_FAST)
    assert _lib.load().mspa_pair_reproject_last_kernel() != _lib.KERNEL_PAIR_FAST_TIGHT
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    for n, (a, b) in enumerate(pair_idx):
        ref = O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, hw)
        check_pair(out_np, n, ref, hw)


@pytest.mark.gpu
def test_compact_argument_checks():
    hw = (96, 128)
    sc, ids, depth, mats = scene_inputs(hw, frames=2)
    pairs = torch.tensor([[0, 1]], dtype=torch.int32, device=DEV)
    out = engine.alloc_pair_correspondences(1, hw, DEV)
    bad = dict(out)
    bad["cpix"] = out["cpix"][:, :1]
    with pytest.raises(ValueError):
        engine.pair_correspondences(depth, mats, pairs, hw, bad)
    lib = _lib.load()
    # the dense-table route without a workspace is refused, with a message that says what to pass
    rc = lib.mspa_pair_correspondences(depth.data_ptr(), mats.data_ptr(), 2, pairs.data_ptr(), 1, 96, 128, 96, 128,
                                       out["vis_bits"].data_ptr(), out["cpix"].data_ptr(), out["tile_counts"].data_ptr(), None,
                                       None, 0, 0, None)
    assert rc == _lib.MSPA_EINVAL and b"workspace" in lib.mspa_last_error_string()
    # a depth table at an odd 2-byte offset cannot feed the fused kernel's 4-byte LDS-DMA: dense route, i.e. a workspace is
    # needed although workspace_bytes (which cannot see the pointer) said 0 -- the error says so; with one, same results
    F = depth.shape[0]
    buf = torch.zeros(F * 96 * 128 + 1, dtype=depth.dtype, device=DEV)
    odd = buf[1:].view(F, 96, 128)
    odd.copy_(depth)
    assert odd.data_ptr() % 4 == 2
    fused = engine.pair_correspondences(depth, mats, pairs, hw, flags=_lib.PAIR_FAST)
    assert lib.mspa_pair_correspondences_workspace_bytes(1, 96, 128, 96, 128, _lib.PAIR_FAST) == 0
    args = (mats.data_ptr(), 2, pairs.data_ptr(), 1, 96, 128, 96, 128, out["vis_bits"].data_ptr(), out["cpix"].data_ptr(),
            out["tile_counts"].data_ptr(), out["counts"].data_ptr())
    rc = lib.mspa_pair_correspondences(odd.data_ptr(), *args, None, 0, _lib.PAIR_FAST, None)
    assert rc == _lib.MSPA_EINVAL and b"4-byte aligned" in lib.mspa_last_error_string()
    ws = torch.empty(96 * 128, dtype=torch.int32, device=DEV)
    rc = lib.mspa_pair_correspondences(odd.data_ptr(), *args, ws.data_ptr(), ws.numel() * 4, _lib.PAIR_FAST, None)
    assert rc == 0 and lib.mspa_pair_reproject_last_kernel() != _lib.KERNEL_PAIR_FAST_TIGHT
    torch.cuda.synchronize()
    for k in ("vis_bits", "tile_counts", "counts"):
        assert torch.equal(out[k], fused[k]), k
    n0 = fused["tile_counts"][0].cpu().numpy()
    for t in range(n0.size):
        assert torch.equal(out["cpix"][0, t, :n0[t]], fused["cpix"][0, t, :n0[t]])
    # the Python wrapper allocates that workspace itself, as its docstring promises ("allocated here when not given")
    via_wrapper = engine.pair_correspondences(odd, mats, pairs, hw, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    for k in ("vis_bits", "tile_counts", "counts"):
        assert torch.equal(via_wrapper[k], fused[k]), k
    # zero pairs: nothing to do
    empty = engine.alloc_pair_correspondences(0, hw, DEV)
    engine.pair_correspondences(depth, mats, pairs[:0], hw, empty)


@pytest.mark.gpu
@pytest.mark.parametrize("hw,n_pairs", [((48, 64), 1), ((48, 64), 3), ((96, 64), 9), ((48, 192), 17)],
                         ids=["one-tile-1", "one-tile-3", "two-bands-9", "three-stripes-17"])
def test_fused_compact_small_shapes_and_odd_batch_sizes(hw, n_pairs):
    """A single tile, batches that are not a multiple of the XCD count, and no counter output."""
    sc, ids, depth, mats = scene_inputs(hw, seed=1717, frames=4)
    rng = np.random.default_rng(n_pairs)
    pair_idx = [(int(a), int(b)) for a, b in rng.integers(0, len(ids), (n_pairs, 2))]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    out = engine.alloc_pair_correspondences(n_pairs, hw, DEV, counts=False)
    for t in out.values():
        t.fill_(-9)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST)
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_TIGHT
    assert "counts" not in out
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    P = hw[0] * hw[1]
    for n, (a, b) in enumerate(pair_idx):
        ref = O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, hw)
        assert np.array_equal(unpack_bits(out_np["vis_bits"][n], P), ref["vis"])
        segs, counts = expected_segments(ref["vis"], ref["xi"], ref["yi"], hw)
        assert np.array_equal(out_np["tile_counts"][n], counts)
        for t, seg in enumerate(segs):
            assert np.array_equal(out_np["cpix"][n, t, :len(seg)], seg)


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(96, 128), (480, 640)], ids=["96x128", "640x480"])
def test_guarded_lanes_that_stay_visible_are_patched_in_place(hw):
    """Fronto-parallel plane, camera 2 shifted sideways by a whole number of pixels: every projection sits on an integer
    (the "u at an integer" guard), so EVERY lane is re-evaluated with the reference chain, while the depth test has a 5 mm
    margin and no lane changes visibility -- each entry of each tile goes through the rewrite-at-its-rank path (a wrong
    rank would clobber a neighbour).  The second pair adds holes so that ranks differ from columns."""
    H, W = hw
    K = np.array([[500.0, 0, W / 2 - 0.5, 0], [0, 500.0, H / 2 - 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    A = np.eye(4)
    shift_px, d_mm = 8, 2000
    E0 = np.eye(4)
    E1 = np.eye(4)
    E1[0, 3] = shift_px * (d_mm * 0.001) / 500.0            # camera-to-world: camera 2 sits to the right
    d0 = np.full((H, W), d_mm, dtype=np.uint16)
    d0_holes = d0.copy()
    rng = np.random.default_rng(5)
    d0_holes[rng.random((H, W)) < 0.3] = 0
    d1 = np.full((H, W), d_mm + 5, dtype=np.uint16)
    frames = [d0, d1, d0_holes]
    Es = [E0, E1, E0]
    depth = engine.depth_to_device(np.stack(frames), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, Es)).to(DEV)
    pair_idx = [(0, 1), (2, 1)]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    out = poisoned_outputs(len(pair_idx), hw)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST)
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_TIGHT
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    for n, (a, b) in enumerate(pair_idx):
        ref = O.frame_pair(frames[a], frames[b], K, Es[a], Es[b], A, hw)
        assert ref["n_vis"] > 0.5 * (frames[a] > 0).sum()               # most of the plane lands in frame 2
        check_pair(out_np, n, ref, hw)
    # nothing past a tile's count was written.  (Not the first stripe and the first band: column 8 projects onto u = 0 and
    # row 0 onto v = 0 up to rounding, where fast and exact chain may disagree about "in view" -- a change of visibility: the
    # tile is rebuilt and may keep stale entries behind its count.)
    cp, tc = out_np["cpix"], out_np["tile_counts"]
    n_stripes = W // TW
    checked = 0
    for n in range(len(pair_idx)):
        for t in range(n_stripes, cp.shape[1]):
            if t % n_stripes:
                assert (cp[n, t, int(tc[n, t]):] == -7).all()
                checked += 1
    assert checked > 0


@pytest.mark.gpu
def test_scene_on_device_pair_correspondences():
    """The resident-scene entry point (mspa/scene.py): image-id pairs in, compacted correspondences out, flat view in
    np.nonzero order -- against the oracle."""
    from mspa.scene import SceneOnDevice
    hw = (96, 128)
    sc = synth.make_scene(1011, n_points=64, n_frames=4, color_hw=hw, depth_hw=hw, invalid_pose_frac=0.0, with_color=False,
                          trajectory="sweep", walk_step=0.08)
    scene = SceneOnDevice(sc.K, sc.A, sc.E, sc.depth, sc.color_hw, sc.points, DEV)
    ids = sc.valid_image_ids
    pair_ids = [(ids[0], ids[1]), (ids[2], ids[0]), (ids[3], ids[3])]
    out = scene.pair_correspondences(pair_ids)
    torch.cuda.synchronize()
    out_np = {k: v.cpu().numpy() for k, v in out.items()}
    for n, (a, b) in enumerate(pair_ids):
        ref = O.frame_pair(sc.depth[a], sc.depth[b], sc.K, sc.E[a], sc.E[b], sc.A, hw)
        check_pair(out_np, n, ref, hw)
        i, xi, yi = engine.correspondences_rowmajor(out, hw, n)
        nz = np.nonzero(ref["vis"])[0]
        assert np.array_equal(i.cpu().numpy(), nz) and np.array_equal(xi.cpu().numpy(), ref["xi"][nz])
        assert np.array_equal(yi.cpu().numpy(), ref["yi"][nz])
    assert scene.pair_correspondences([])["tile_counts"].shape[0] == 0
