"""The specialised fast kernels the benchmark times, one test per instantiation.

``pair_fast_tight_kernel<SET, STREAM>`` only runs when the output set is EXACTLY corr / dense / minimal on a whole-tile
image; a test that asks for one output more silently measures the generic kernel instead.  Every case here asserts through
``mspa_pair_reproject_last_kernel`` that the instantiation it means to check is the one that was launched, and compares
every output with the C and NumPy oracles (bit-exact integers, float32 points to 2e-7 relative) -- at 96x128 on seeded and
on adversarial poses, and at the BASELINE shape 640x480 against the reference's own frozen outputs
(tests/golden/k3_640x480.npz, oracle/gen_golden.py golden_k3_640x480).
"""
import hashlib

import numpy as np
import pytest
import torch

from golden_util import GoldenScene, same_f64
from mspa import engine, synth, _lib
from oracle import c_oracle as C
from oracle import np_oracle as O

DEV = "cuda"

SETS = {
    "corr": ("vis_bits", "pix_i16", "counts"),
    "dense": ("vis_u8", "pix_i16", "xyz_f32", "rgba", "counts"),
    "dense_xyz": ("vis_u8", "pix_i16", "xyz_f32", "counts"),          # SURVEY 8d's dense with rgb = 0: no colour words
    "minimal": ("vis_bits", "counts"),
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def stable_color_image(hw):
    H, W = hw
    return ((np.arange(H * W * 3, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(7)).astype(np.uint8).reshape(H, W, 3)


def unpack_bits(words, n):
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n].astype(bool)


def launch(depth, mats, rgb, pairs, hw, outputs, flags):
    out = engine.alloc_pair_outputs(pairs.shape[0], hw, outputs, DEV)
    for t in out.values():
        t.fill_(0x5A if t.dtype == torch.uint8 else 0x5A5A5A5A if t.dtype == torch.int32 else 23)   # poison
    engine.pair_reproject(depth, mats, pairs, hw, out, rgb=rgb if "rgba" in outputs else None, flags=flags)
    kern = _lib.load().mspa_pair_reproject_last_kernel()
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}, kern


def check_set(res, n, name, ref_np, ref_c, hw, color):
    """Every output of one pair of one output set against the oracles."""
    P = hw[0] * hw[1]
    valid = ref_np["valid"]
    assert np.array_equal(ref_np["vis"], ref_c["vis"])
    assert tuple(res["counts"][n]) == (ref_np["n_valid"], ref_np["n_vis"])
    if "vis_bits" in res:
        assert np.array_equal(unpack_bits(res["vis_bits"][n], P), ref_np["vis"])
    if "vis_u8" in res:
        assert np.array_equal(res["vis_u8"][n], ref_np["vis"].astype(np.uint8))          # 0 / 1 exactly, no poison left
    with np.errstate(invalid="ignore"):
        inview = valid & O.check_point_in_image_boundary(ref_np["uv2"], hw) & (ref_np["depth2"] > 0)
    if "pix_i16" in res:
        pix = res["pix_i16"][n]
        assert np.array_equal(pix[inview, 0], ref_np["xi"][inview]) and np.array_equal(pix[inview, 1], ref_np["yi"][inview])
        assert (pix[~inview] == -1).all()
    if "xyz_f32" in res:
        f32 = res["xyz_f32"][n]
        assert np.isnan(f32[~valid]).all() and np.isfinite(f32[valid]).all()
        assert np.allclose(f32[valid], ref_c["xyz"][valid], rtol=2e-7, atol=1e-7)
    if "rgba" in res:
        rgba = res["rgba"][n].view(np.uint32)
        exp = color.reshape(-1, 3).astype(np.uint32)
        exp = exp[:, 0] | (exp[:, 1] << 8) | (exp[:, 2] << 16) | np.where(valid, 0xFF000000, 0).astype(np.uint32)
        assert np.array_equal(rgba, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("stream", [False, True], ids=["plain", "stream"])
@pytest.mark.parametrize("name", list(SETS))
@pytest.mark.parametrize("hw", [(96, 128), (480, 640)], ids=["96x128", "640x480"])
def test_tight_instantiation_vs_oracle(hw, name, stream):
    """Seeded scene, neighbouring + distant + identity pairs: the tight kernel of every output set, with and without the
    streaming hint, against the oracles, output by output."""
    sc = synth.make_scene(1010, n_points=64, n_frames=5, color_hw=hw, depth_hw=hw, invalid_pose_frac=0.0, with_color=True,
                          trajectory="sweep", walk_step=0.08)
    ids = sc.valid_image_ids
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])).to(DEV)
    rgb = torch.from_numpy(np.stack([sc.color[i] for i in ids])).to(DEV)
    pair_idx = [(0, 1), (1, 0), (0, 4), (3, 3), (4, 2)] if hw[0] < 200 else [(0, 1), (4, 0), (2, 2)]
    pairs = torch.tensor(pair_idx, dtype=torch.int32, device=DEV)
    flags = _lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0)
    res, kern = launch(depth, mats, rgb, pairs, hw, SETS[name], flags)
    assert kern == _lib.KERNEL_PAIR_FAST_TIGHT, "the tight instantiation must be the kernel under test"
    seen_vis = 0
    for n, (a, b) in enumerate(pair_idx):
        ia, ib = ids[a], ids[b]
        ref_np = O.frame_pair(sc.depth[ia], sc.depth[ib], sc.K, sc.E[ia], sc.E[ib], sc.A, hw, sc.color[ia])
        ref_c = C.frame_pair(sc.depth[ia], sc.depth[ib], sc.K, sc.E[ia], sc.E[ib], sc.A, hw)
        check_set(res, n, name, ref_np, ref_c, hw, sc.color[ia])
        seen_vis += ref_np["n_vis"]
    assert seen_vis > 0
    # one output more and it is a different kernel: the generic one (what round 1's tests exercised)
    _, kern2 = launch(depth, mats, rgb, pairs, hw, SETS[name] + ("valid_u8",), flags)
    assert kern2 == _lib.KERNEL_PAIR_FAST


def adversarial_pairs(rng, n, hw):
    """Camera pairs that stress the culling and guard logic: looking away, coincident, grazing, very close, exact
    half-pixel shifts (same recipe as test_fast_equals_exact_random_poses in test_gpu_heads / test_gpu_parity)."""
    K = synth.intrinsics_for(hw)
    A = np.eye(4)
    E = []
    for k in range(n):
        kind = k % 6
        eye = rng.uniform([1, 1, 1.2], [5, 5, 1.9])
        tgt = synth.ROOM / 2 + rng.normal(0, 1.0, 3) * [1, 1, 0.4]
        if kind == 1:
            tgt = eye + (eye - tgt)                       # looking the other way
        e = synth._look_at(eye, tgt)
        if kind == 2 and E:
            e = E[-1].copy()                              # coincident with the previous camera
        if kind == 3 and E:
            e = E[-1].copy()
            e[:3, 3] += e[:3, 0] * (0.5 / K[0, 0]) * 2.0   # exact half-pixel shift at z = 2
        if kind == 4:
            e[:3, 3] = rng.uniform([0.05, 0.05, 0.1], [0.3, 0.3, 0.4])   # in a corner, grazing the walls
        E.append(synth._roundtrip_f(e))
    return K, A, E


@pytest.mark.gpu
@pytest.mark.parametrize("stream", [False, True], ids=["plain", "stream"])
@pytest.mark.parametrize("name", list(SETS))
def test_tight_equals_exact_on_adversarial_poses(name, stream):
    """240 adversarial pairs at 96x128: every integer output of every tight instantiation equals the exact kernel's
    (which is bit-identical to the C oracle, test_gpu_parity.py), float32 points within 2e-7."""
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
    rgb = torch.from_numpy(rng.integers(0, 256, (len(E),) + hw + (3,), dtype=np.uint8)).to(DEV)
    pair_np = np.stack([rng.integers(0, len(E), 240), rng.integers(0, len(E), 240)], 1).astype(np.int32)
    pair_np[:24] = np.arange(24)[:, None]                 # identity pairs: everything lands on exact integers
    pairs = torch.from_numpy(pair_np).to(DEV)
    flags = _lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0)
    fast, kf = launch(depth, mats, rgb, pairs, hw, SETS[name], flags)
    exact, ke = launch(depth, mats, rgb, pairs, hw, SETS[name], 0)
    assert kf == _lib.KERNEL_PAIR_FAST_TIGHT and ke == _lib.KERNEL_PAIR_EXACT
    for k in SETS[name]:
        if k == "xyz_f32":
            assert np.allclose(fast[k], exact[k], rtol=2e-7, atol=1e-7, equal_nan=True)
        else:
            assert np.array_equal(fast[k], exact[k]), f"{name}/{'stream' if stream else 'plain'}: {k} differs from the exact kernel"
    assert int(exact["counts"][:, 1].sum()) > 1000


@pytest.mark.gpu
def test_tight_equals_exact_on_adversarial_poses_at_640x480():
    """The same at the BASELINE shape itself: 72 adversarial pairs of 640x480 frames (8 identity pairs among them), every
    tight instantiation and the fused compacted set against the exact kernel, every integer output."""
    hw = (480, 640)
    rng = np.random.default_rng(4242)
    K, A, E = adversarial_pairs(rng, 16, hw)
    boxes = synth._make_boxes(rng)
    depth_np = []
    for e in E:
        z = synth.render_depth(A @ e, K, hw, boxes)
        mm = np.clip(np.rint(z * 1000.0 + rng.normal(0, 4.0, z.shape)), 0, 65535).astype(np.uint16)
        mm[rng.random(mm.shape) < 0.07] = 0
        depth_np.append(mm)
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    rgb = torch.from_numpy(rng.integers(0, 256, (len(E),) + hw + (3,), dtype=np.uint8)).to(DEV)
    pair_np = np.stack([rng.integers(0, len(E), 72), rng.integers(0, len(E), 72)], 1).astype(np.int32)
    pair_np[:8] = np.arange(8)[:, None]
    pairs = torch.from_numpy(pair_np).to(DEV)
    for name in SETS:
        fast, kf = launch(depth, mats, rgb, pairs, hw, SETS[name], _lib.PAIR_FAST | _lib.PAIR_STREAM)
        exact, ke = launch(depth, mats, rgb, pairs, hw, SETS[name], 0)
        assert kf == _lib.KERNEL_PAIR_FAST_TIGHT and ke == _lib.KERNEL_PAIR_EXACT
        for k in SETS[name]:
            if k == "xyz_f32":
                assert np.allclose(fast[k], exact[k], rtol=2e-7, atol=1e-7, equal_nan=True)
            else:
                assert np.array_equal(fast[k], exact[k]), f"{name}: {k} differs from the exact kernel at 640x480"
        assert int(exact["counts"][:, 1].sum()) > 100000
        del fast, exact
    fused = engine.pair_correspondences(depth, mats, pairs, hw, flags=_lib.PAIR_FAST | _lib.PAIR_STREAM)
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_TIGHT
    route = engine.pair_correspondences(depth, mats, pairs, hw, flags=0)
    torch.cuda.synchronize()
    for k in ("vis_bits", "counts", "tile_counts"):
        assert torch.equal(fused[k], route[k]), k
    tc = route["tile_counts"].cpu().numpy()
    f, e = fused["cpix"].cpu().numpy(), route["cpix"].cpu().numpy()
    keep = np.arange(f.shape[2])[None, None, :] < tc[:, :, None]
    assert np.array_equal(f[keep], e[keep])


# ---- the reference's own outputs at the BASELINE shape -------------------------------------------------
@pytest.fixture(scope="module")
def g640():
    return GoldenScene("k3_640x480")


def test_oracle_reproduces_reference_at_640x480(g640):
    """CPU: the NumPy oracle against the frozen reference digests (pins the oracle at the BASELINE shape)."""
    g = g640
    hw = g.color_hw
    assert hw == (480, 640) == g.depth_hw
    color = stable_color_image(hw)
    for n, (f0, f1) in enumerate(g["pair_ids"]):
        f0, f1 = str(f0), str(f1)
        a7 = O.project_mask_to_3d(g.depth[f0], g.K, g.E[f0], None, g.A, color)
        assert a7.shape[0] == int(g[f"pair{n}_rows"]) and same_f64(a7[:64, :3], g[f"pair{n}_xyz_head"])
        assert sha(a7[:, :3]) == str(g[f"pair{n}_sha_xyz"]) and sha(a7[:, 3:]) == str(g[f"pair{n}_sha_rgb"])
        uv, d = O.project_3d_point_to_image(a7[:, :3], g.K, O.aligned_extrinsic(g.A, g.E[f1]))
        assert sha(uv) == str(g[f"pair{n}_sha_uv"]) and sha(d) == str(g[f"pair{n}_sha_depth"])
        r = O.frame_pair(g.depth[f0], g.depth[f1], g.K, g.E[f0], g.E[f1], g.A, hw, color)
        assert np.array_equal(np.packbits(r["vis"], bitorder="little"), g[f"pair{n}_vis_bits"])
        assert r["n_vis"] == int(g[f"pair{n}_n_vis"])


@pytest.mark.gpu
def test_kernels_reproduce_reference_at_640x480(g640):
    """GPU: the exact kernel reproduces the reference's float64 bytes; all six tight instantiations reproduce its visibility
    bitset, counters and pixel-index table (SHA-256 of the [P, 2] int16 array as K3 lays it out)."""
    g = g640
    hw = g.color_hw
    P = hw[0] * hw[1]
    ids = g.valid_image_ids
    fidx = {i: k for k, i in enumerate(ids)}
    depth = engine.depth_to_device(np.stack([g.depth[i] for i in ids]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(g.K, g.A, [g.E[i] for i in ids])).to(DEV)
    color = stable_color_image(hw)
    rgb = torch.from_numpy(np.stack([color] * len(ids))).to(DEV)
    pair_ids = [(str(a), str(b)) for a, b in g["pair_ids"]]
    pairs = torch.tensor([[fidx[a], fidx[b]] for a, b in pair_ids], dtype=torch.int32, device=DEV)
    ex, kern = launch(depth, mats, rgb, pairs, hw, ("valid_u8", "vis_bits", "pix_i16", "xyz_f64", "uv_f64", "depth_f64", "rgba",
                                                     "counts"), 0)
    assert kern == _lib.KERNEL_PAIR_EXACT
    for n in range(len(pair_ids)):
        valid = ex["valid_u8"][n].astype(bool)
        assert int(valid.sum()) == int(g[f"pair{n}_rows"]) == int(ex["counts"][n, 0])
        assert sha(ex["xyz_f64"][n][valid]) == str(g[f"pair{n}_sha_xyz"])
        assert sha(ex["uv_f64"][n][valid]) == str(g[f"pair{n}_sha_uv"])
        assert sha(ex["depth_f64"][n][valid]) == str(g[f"pair{n}_sha_depth"])
        rgba = ex["rgba"][n].view(np.uint32)[valid]
        cols = np.stack([rgba & 0xFF, (rgba >> 8) & 0xFF, (rgba >> 16) & 0xFF], axis=1).astype(np.float64)
        assert sha(cols) == str(g[f"pair{n}_sha_rgb"])
        assert np.array_equal(ex["vis_bits"][n].view(np.uint8)[:P // 8], g[f"pair{n}_vis_bits"])
        assert sha(ex["pix_i16"][n]) == str(g[f"pair{n}_sha_pix"])
    for name, outs in SETS.items():
        for stream in (False, True):
            res, kern = launch(depth, mats, rgb, pairs, hw, outs, _lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0))
            assert kern == _lib.KERNEL_PAIR_FAST_TIGHT
            for n in range(len(pair_ids)):
                tag = f"{name}/{'stream' if stream else 'plain'} pair {n}"
                assert tuple(res["counts"][n]) == (int(g[f"pair{n}_rows"]), int(g[f"pair{n}_n_vis"])), tag
                if "vis_bits" in res:
                    assert np.array_equal(res["vis_bits"][n].view(np.uint8)[:P // 8], g[f"pair{n}_vis_bits"]), tag
                if "vis_u8" in res:
                    assert np.array_equal(np.packbits(res["vis_u8"][n], bitorder="little"), g[f"pair{n}_vis_bits"]), tag
                if "pix_i16" in res:
                    assert sha(res["pix_i16"][n]) == str(g[f"pair{n}_sha_pix"]), tag


# ---- ScanNet's own shape: 1296 x 968 colour over 640 x 480 depth (pair_fast_scaled_kernel) ---------------------
SCANNET_HW, SCANNET_DHW = (968, 1296), (480, 640)


@pytest.mark.gpu
@pytest.mark.parametrize("stream", [False, True], ids=["plain", "stream"])
@pytest.mark.parametrize("name", ["corr", "minimal"])
def test_scaled_kernel_equals_exact_on_adversarial_poses(name, stream):
    """The kernel specialised for ScanNet's shape (wobbling stripes, row-straddling words in the last stripe, both grid
    scalings) against the exact kernel: 96 adversarial pairs + identity pairs, every integer output."""
    rng = np.random.default_rng(99)
    K, A, E = adversarial_pairs(rng, 12, SCANNET_HW)
    Kd = K.copy()
    Kd[0] *= SCANNET_DHW[1] / SCANNET_HW[1]
    Kd[1] *= SCANNET_DHW[0] / SCANNET_HW[0]
    boxes = synth._make_boxes(rng)
    depth_np = []
    for e in E:
        z = synth.render_depth(A @ e, Kd, SCANNET_DHW, boxes)
        mm = np.clip(np.rint(z * 1000.0 + rng.normal(0, 4.0, z.shape)), 0, 65535).astype(np.uint16)
        mm[rng.random(mm.shape) < 0.07] = 0
        depth_np.append(mm)
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    pair_np = np.stack([rng.integers(0, len(E), 96), rng.integers(0, len(E), 96)], 1).astype(np.int32)
    pair_np[:12] = np.arange(12)[:, None]
    pairs = torch.from_numpy(pair_np).to(DEV)
    flags = _lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0)
    outs = SETS[name]
    exact = engine.alloc_pair_outputs(len(pair_np), SCANNET_HW, outs, DEV)
    engine.pair_reproject(depth, mats, pairs, SCANNET_HW, exact, flags=0)
    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_EXACT
    # round 4: the rectangular-tile kernel is the default at this shape; the wobbling-stripe kernel under MSPA_PAIR_WORD_STRIPES
    for extra, want in ((_lib.PAIR_WORD_STRIPES, _lib.KERNEL_PAIR_FAST_SCALED), (0, _lib.KERNEL_PAIR_FAST_RECT)):
        fast = engine.alloc_pair_outputs(len(pair_np), SCANNET_HW, outs, DEV)
        for t in fast.values():
            t.fill_(23)
        engine.pair_reproject(depth, mats, pairs, SCANNET_HW, fast, flags=flags | extra)
        assert _lib.load().mspa_pair_reproject_last_kernel() == want
        torch.cuda.synchronize()
        for k in outs:
            if not torch.equal(fast[k], exact[k]):
                bad = (fast[k] != exact[k]).reshape(len(pair_np), -1).any(dim=1).nonzero().flatten().tolist()
                raise AssertionError(f"{name}/{'stream' if stream else 'plain'}/kernel {want}: {k} differs from the exact kernel in pairs {bad[:8]}")
    assert int(exact["counts"][:, 1].sum()) > 100000


@pytest.mark.gpu
def test_scaled_kernel_reproduces_reference_digest():
    """tests/golden/scannet_shape.npz: the reference's visibility mask of the full-frame pair (SHA-256 over the valid rows)."""
    g = GoldenScene("scannet_shape")
    f0, f1 = (str(x) for x in g["pair_ids"][0])
    depth = engine.depth_to_device(np.stack([g.depth[f0], g.depth[f1]]), DEV)
    mats = torch.from_numpy(engine.frame_matrices(g.K, g.A, [g.E[f0], g.E[f1]])).to(DEV)
    pairs = torch.tensor([[0, 1], [1, 0]], dtype=torch.int32, device=DEV)
    P = g.color_hw[0] * g.color_hw[1]
    ex, _ = launch(depth, mats, None, pairs, g.color_hw, ("valid_u8", "vis_bits", "pix_i16", "counts"), 0)
    for stream, extra, want in ((False, _lib.PAIR_WORD_STRIPES, _lib.KERNEL_PAIR_FAST_SCALED), (True, _lib.PAIR_WORD_STRIPES, _lib.KERNEL_PAIR_FAST_SCALED),
                                (False, 0, _lib.KERNEL_PAIR_FAST_RECT), (True, 0, _lib.KERNEL_PAIR_FAST_RECT)):
        res, kern = launch(depth, mats, None, pairs, g.color_hw, SETS["corr"], _lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0) | extra)
        assert kern == want
        valid = ex["valid_u8"][0].astype(bool)
        vis = unpack_bits(res["vis_bits"][0], P)
        assert sha(vis[valid]) == str(g["pair_sha_vis"]) and int(vis.sum()) == int(g["pair_n_vis"]) == int(res["counts"][0, 1])
        for k in ("vis_bits", "pix_i16", "counts"):
            assert np.array_equal(res[k], ex[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("hw,dhw,kernel", [((480, 640), (480, 640), "tight"), (SCANNET_HW, SCANNET_DHW, "scaled"), (SCANNET_HW, SCANNET_DHW, "rect")],
                         ids=["640x480", "scannet-word-stripes", "scannet"])
def test_tile_culling_with_large_holes_and_distant_views(hw, dhw, kernel):
    """Tile-level culling: tiles whose depth box holds no valid sample at all (a quarter of frame 0 is a hole), tiles
    that cannot land in the other view (cameras back to back, side by side looking apart) and tiles that straddle the edge
    of the other view -- every integer output of the fast kernel equals the exact kernel's."""
    rng = np.random.default_rng(17)
    H, W = hw
    K = np.array([[1.1 * W / 1.3, 0, W / 2 - 0.5, 0], [0, 1.1 * W / 1.3, H / 2 - 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    Kd = K.copy()
    Kd[0] *= dhw[1] / W
    Kd[1] *= dhw[0] / H
    A = np.eye(4)

    def pose(yaw_deg, tx):
        c, s = np.cos(np.radians(yaw_deg)), np.sin(np.radians(yaw_deg))
        E = np.eye(4)
        E[:3, :3] = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
        E[0, 3] = tx
        return E
    E = [pose(0, 0), pose(35, 0.3), pose(180, 0), pose(-70, -0.5), pose(8, 0.05)]
    boxes = synth._make_boxes(rng)
    depth_np = []
    for k, e in enumerate(E):
        z = synth.render_depth(A @ e, Kd, dhw, boxes)
        mm = np.clip(np.rint(z * 1000.0), 0, 65535).astype(np.uint16)
        if k == 0:
            mm[:dhw[0] // 2, :dhw[1] // 2] = 0                      # a hole of whole tiles
        if k == 4:
            mm[rng.random(mm.shape) < 0.5] = 0
        depth_np.append(mm)
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    pair_np = np.array([(a, b) for a in range(5) for b in range(5)], dtype=np.int32)
    pairs = torch.from_numpy(pair_np).to(DEV)
    outs = SETS["corr"]
    for stream in (False, True):
        fast = engine.alloc_pair_outputs(len(pair_np), hw, outs, DEV)
        exact = engine.alloc_pair_outputs(len(pair_np), hw, outs, DEV)
        for t in fast.values():
            t.fill_(23)
        engine.pair_reproject(depth, mats, pairs, hw, fast, flags=_lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0) |
                              (_lib.PAIR_WORD_STRIPES if kernel == "scaled" else 0))
        assert _lib.load().mspa_pair_reproject_last_kernel() == {"tight": _lib.KERNEL_PAIR_FAST_TIGHT, "scaled": _lib.KERNEL_PAIR_FAST_SCALED,
                                                                "rect": _lib.KERNEL_PAIR_FAST_RECT}[kernel]
        engine.pair_reproject(depth, mats, pairs, hw, exact, flags=0)
        torch.cuda.synchronize()
        for k in outs:
            if not torch.equal(fast[k], exact[k]):
                bad = (fast[k] != exact[k]).reshape(len(pair_np), -1).any(dim=1).nonzero().flatten().tolist()
                raise AssertionError(f"{k} differs from the exact kernel in pairs {[tuple(pair_np[i]) for i in bad[:8]]}")
        c = exact["counts"].cpu().numpy()
        assert c[5 * 0 + 2, 1] == 0 and c[5 * 2 + 0, 1] == 0          # back to back: nothing visible
        assert c[:, 1].sum() > 100000
