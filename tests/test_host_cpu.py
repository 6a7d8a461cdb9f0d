"""Host-side logic that needs no GPU: the C-ABI surface, matrix preparation, sharding + collation."""
import ctypes
import os
import re
import socket

import numpy as np
import pytest
import torch

from mspa import _lib, engine, shard, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mspa.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mspa_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed from include/mspa.h"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mspa.h but not exported by libmspa.so"
    assert sorted(_lib.exported_symbols()) == declared, "mspa/_lib.py signatures out of sync with the header"
    assert _lib.version() == int(re.search(r"#define MSPA_VERSION (\d+)", header).group(1))


def test_error_reporting_without_gpu_calls():
    lib = _lib.load()
    # argument validation happens before any HIP call, so it is checkable on a CPU-only box
    rc = lib.mspa_pair_overlap(None, 1, 1, None, 0, None, None, None, None)
    assert rc == _lib.MSPA_EINVAL and b"null pointer" in lib.mspa_last_error_string()
    rc = lib.mspa_pair_reproject(None, None, None, 1, None, 0, 480, 640, 480, 640, *([None] * 10), 0, None)
    assert rc == _lib.MSPA_EINVAL
    dummy = ctypes.c_void_p(64)
    rc = lib.mspa_pair_reproject(dummy, None, dummy, 1, dummy, 1, 480, 640, 480, 640, *([None] * 10), 4, None)   # unknown flag bit
    assert rc == _lib.MSPA_EINVAL and b"unknown flag" in lib.mspa_last_error_string()
    with pytest.raises(_lib.MspaError):
        _lib.check(rc)


def test_missing_library_fails_loudly():
    """No libmspa.so -> ImportError naming the build command; never a silent CPU path."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mspa import _lib\n"
            "try:\n    _lib.load()\nexcept ImportError as e:\n    print('IMPORTERROR', 'no CPU fallback' in str(e)); sys.exit(0)\n"
            "print('LOADED'); sys.exit(1)\n") % os.path.join(ROOT, "multi-spatialmllm_amd")
    env = dict(os.environ, MSPA_LIB="/nonexistent/libmspa.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "IMPORTERROR True" in out.stdout, out.stdout + out.stderr


def test_size_limits_are_rejected():
    lib = _lib.load()
    dummy = ctypes.c_void_p(64)       # never dereferenced: validation comes first
    # width beyond int16 pixel indices; H*W*W >= 2^32
    for (dh, dw, H, W) in ((480, 640, 480, 40000), (480, 640, 2000, 2000), (1, 640, 480, 640)):
        rc = lib.mspa_pair_reproject(dummy, None, dummy, 1, dummy, 1, dh, dw, H, W, *([None] * 10), 0, None)
        assert rc == _lib.MSPA_EINVAL, (dh, dw, H, W)
    rc = lib.mspa_vertex_visibility(dummy, 10, 3, 1, dummy, 1, dummy, 480, 640, 480, 70000, *([None] * 5), None)
    assert rc == _lib.MSPA_EINVAL
    rc = lib.mspa_pair_overlap(dummy, 2, 1 << 26, dummy, 1, dummy, None, None, None)
    assert rc == _lib.MSPA_EINVAL and b"too long" in lib.mspa_last_error_string()


def test_tiled_overlap_entry_points_validate_before_launching():
    lib = _lib.load()
    dummy = ctypes.c_void_p(64)
    need = lib.mspa_overlap_workspace_bytes(320, 320, 2048)
    assert need >= 100 * 32 * 4096 and lib.mspa_overlap_workspace_bytes(0, 5, 7) == 0
    assert lib.mspa_overlap_workspace_bytes(40, 320, 2048) > 0
    assert lib.mspa_scene_overlap(dummy, 320, 2048, dummy, need - 1, dummy, None, None, None) == _lib.MSPA_EINVAL
    assert b"workspace smaller" in lib.mspa_last_error_string()
    assert lib.mspa_scene_overlap(dummy, 320, 2048, None, need, dummy, None, None, None) == _lib.MSPA_EINVAL
    assert lib.mspa_scene_overlap(None, 1, 2048, None, 0, None, None, None, None) == _lib.MSPA_OK       # no pair
    assert lib.mspa_scene_overlap(dummy, 320, 1 << 26, dummy, 1 << 40, dummy, None, None, None) == _lib.MSPA_EINVAL
    assert lib.mspa_overlap_matrix(dummy, 0, dummy, 5, 7, None, 0, None, None) == _lib.MSPA_OK
    assert lib.mspa_overlap_matrix(dummy, 3, dummy, 5, 7, dummy, 8, dummy, None) == _lib.MSPA_EINVAL
    assert lib.mspa_overlap_matrix(dummy, 3, None, 5, 7, dummy, 1 << 20, dummy, None) == _lib.MSPA_EINVAL


def test_new_entry_points_validate_before_launching():
    """K5c / K7 / K8: bad sizes and null pointers come back as MSPA_EINVAL, empty inputs as MSPA_OK -- no HIP call either way."""
    lib = _lib.load()
    dummy = ctypes.c_void_p(64)
    assert lib.mspa_track_rigidity_loss(dummy, -1, 4, 0.01, dummy, None) == _lib.MSPA_EINVAL
    assert lib.mspa_track_rigidity_loss(None, 3, 4, 0.01, dummy, None) == _lib.MSPA_EINVAL
    assert lib.mspa_track_rigidity_loss(None, 3, 0, 0.01, None, None) == _lib.MSPA_OK            # no points: nothing to do
    assert lib.mspa_track_rigidity_loss(dummy, 3, 1 << 20, 0.01, dummy, None) == _lib.MSPA_EINVAL   # P*P/256 blocks > 2^31
    assert lib.mspa_object_extents(dummy, 4, 1, dummy, 100, dummy, dummy, 5, 2, dummy, dummy, dummy, None) == _lib.MSPA_EINVAL
    assert b"shorter than the vertex count" in lib.mspa_last_error_string()
    assert lib.mspa_object_extents(None, 4, 2, dummy, 100, dummy, dummy, 5, 2, dummy, dummy, dummy, None) == _lib.MSPA_EINVAL
    assert lib.mspa_object_extents(None, 0, 2, None, 100, None, None, 0, 2, None, None, None, None) == _lib.MSPA_OK
    assert lib.mspa_object_extents(dummy, 65536 * 64, 2, dummy, 100, dummy, dummy, 5, 2, dummy, dummy, dummy, None) == _lib.MSPA_EINVAL
    assert lib.mspa_track_pair_distances(dummy, 3, 4, dummy, dummy, dummy, -1, 3, dummy, dummy, None) == _lib.MSPA_EINVAL
    assert lib.mspa_track_pair_distances(None, 3, 4, dummy, dummy, dummy, 2, 3, dummy, dummy, None) == _lib.MSPA_EINVAL
    assert lib.mspa_track_pair_distances(None, 3, 4, None, None, None, 0, 0, None, None, None) == _lib.MSPA_OK
    assert lib.mspa_track_pair_distances(dummy, 3, 4, dummy, dummy, dummy, 70000, 3, dummy, dummy, None) == _lib.MSPA_EINVAL


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_refuses_to_run_without_gpu():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.pair_overlap(torch.zeros((2, 2), dtype=torch.int64), torch.zeros((1, 2), dtype=torch.int32))


def test_matrix_preparation_matches_numpy():
    sc = synth.make_scene(5, n_points=16, n_frames=3, color_hw=(24, 32), depth_hw=(24, 32), invalid_pose_frac=0,
                          with_color=False)
    E = [sc.E[i] for i in sc.valid_image_ids]
    m = engine.frame_matrices(sc.K, sc.A, E)
    assert m.shape == (3, _lib.FRAME_MATS, 16) and m.dtype == np.float64
    for f in range(3):
        assert np.array_equal(m[f, _lib.MAT_KINV].reshape(4, 4), np.linalg.inv(sc.K))
        assert np.array_equal(m[f, _lib.MAT_E].reshape(4, 4), E[f])
        assert np.array_equal(m[f, _lib.MAT_EINV_ALIGNED].reshape(4, 4), np.linalg.inv(sc.A @ E[f]))
        assert np.allclose(m[f, _lib.MAT_UNPROJ].reshape(4, 4), sc.A @ E[f] @ np.linalg.inv(sc.K), rtol=1e-15)
        assert np.allclose(m[f, _lib.MAT_REPROJ].reshape(4, 4), sc.K @ np.linalg.inv(sc.A @ E[f]), rtol=1e-15)
    # slot MSPA_MAT_BOUNDS: the guard-bound coefficients, NumPy form == the library's host helper (up to the summation order
    # of the magnitudes), and what they claim -- magnitudes of the chain's two halves -- holds
    m2 = m.copy()
    m2[:, _lib.MAT_BOUNDS] = -1.0
    _lib.check(_lib.load().mspa_frame_bounds_host(m2.ctypes.data, m2.shape[0]))
    assert np.allclose(m2[:, _lib.MAT_BOUNDS], m[:, _lib.MAT_BOUNDS], rtol=1e-14, atol=0)
    assert np.array_equal(m2[:, :_lib.MAT_BOUNDS], m[:, :_lib.MAT_BOUNDS])
    c = _lib.GUARD_C * 2.0 ** -53
    for f in range(3):
        b = m[f, _lib.MAT_BOUNDS]
        Ua = np.abs(sc.A) @ np.abs(E[f]) @ np.abs(np.linalg.inv(sc.K))
        Na = np.abs(sc.K) @ np.abs(np.linalg.inv(sc.A @ E[f]))
        assert np.allclose(b[:3], Ua[:3, :3].max(axis=0), rtol=1e-14) and np.isclose(b[3], 1000.0 * Ua[:3, 3].max(), rtol=1e-14)
        assert np.allclose(b[4:7], c * Na[:3, :3].sum(axis=1), rtol=1e-14) and np.allclose(b[8:11], c * 1000.0 * Na[:3, 3], rtol=1e-14)
        assert b[7] == 0 and (b[11:] == 0).all()
    assert _lib.load().mspa_frame_bounds_host(None, 3) == _lib.MSPA_EINVAL
    assert engine.fast_path_ok(sc.K)
    bad_K = sc.K.copy()
    bad_K[2, 3] = 0.5
    assert not engine.fast_path_ok(bad_K)
    for f in range(3):
        pass
    c = engine.camera_matrices(sc.K, [sc.A @ e for e in E])
    assert c.shape == (3, _lib.CAM_MATS, 16)
    assert np.array_equal(c[1, 0].reshape(4, 4), np.linalg.inv(sc.A @ E[1]))
    c2 = c.copy()
    c2[:, _lib.CAM_BOUNDS] = -1.0                    # slot MSPA_CAM_BOUNDS: NumPy form == the library's host helper
    _lib.check(_lib.load().mspa_camera_bounds_host(c2.ctypes.data, c2.shape[0]))
    assert np.allclose(c2[:, _lib.CAM_BOUNDS], c[:, _lib.CAM_BOUNDS], rtol=1e-14, atol=0) and (c[:, _lib.CAM_BOUNDS, :4] > 0).all()
    assert np.array_equal(c2[:, :_lib.CAM_BOUNDS], c[:, :_lib.CAM_BOUNDS])
    with pytest.raises(ValueError):
        engine.frame_matrices(sc.K, sc.A, [np.full((4, 4), np.nan)])


def test_partition_and_lpt():
    for n in (0, 1, 7, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard.partition(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    costs = [shard.scene_cost(f, 131072) for f in (300, 20, 64, 64, 500, 90, 10, 250, 250)]
    bins = shard.lpt_assign(costs, 3)
    assert sorted(i for b in bins for i in b) == list(range(len(costs)))
    loads = [sum(costs[i] for i in b) for b in bins]
    assert max(loads) <= sum(costs) / 3 + max(costs)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mspa import shard as S, synth as Y
    from oracle import np_oracle as O
    ctx = S.init_distributed(torch.device("cpu"))
    sc = Y.make_scene(31, n_points=16, n_frames=5, color_hw=(24, 32), depth_hw=(24, 32), invalid_pose_frac=0,
                      with_color=False)
    ids = sc.valid_image_ids
    all_pairs = [(i, j) for i in range(len(ids)) for j in range(len(ids)) if i != j][:13]   # 13: uneven split
    lo, hi = S.partition(len(all_pairs), world, rank)
    recs = []
    col = np.zeros((24, 32, 3), np.uint8)
    for (i, j) in all_pairs[lo:hi]:      # the CPU stand-in for the per-rank kernel launch (tests only)
        r = O.frame_pair(sc.depth[ids[i]], sc.depth[ids[j]], sc.K, sc.E[ids[i]], sc.E[ids[j]], sc.A, (24, 32), col)
        recs.append([i, j, r["n_valid"], r["n_vis"]])
    local = torch.tensor(recs, dtype=torch.int32).reshape(-1, 4)
    full = S.collate_records(local, ctx)
    # fixed-size asynchronous form used by bench.py (one all_gather, no count exchange)
    fixed = torch.full((5, 2), rank + 1, dtype=torch.int32)
    g, work = S.collate_records_async(fixed, ctx)
    work.wait()
    assert g.shape == (5 * world, 2) and all(int(g[5 * r, 0]) == r + 1 for r in range(world))
    t = ctx.max_over_ranks(float(rank + 1))
    # ragged float64 rows towards rank 0 only, as the scene sweeps collate a window (mspa/sweep.py): a rank may have none
    rows = torch.arange(7 * (3 + 5 * rank), dtype=torch.float64).reshape(-1, 7) * 0.1 + rank if rank else torch.zeros((0, 7), dtype=torch.float64)
    to0 = S.collate_records(rows, ctx, dst=0).numpy()
    everywhere = S.collate_records(rows, ctx).numpy()
    assert np.array_equal(to0, everywhere) if rank == 0 else to0.shape == (0, 7)
    assert everywhere.shape == (8, 7) and everywhere[0, 0] == 1.0
    # the pipeline's exchange of FINISHED records (mspa/pipeline.py): per-rank JSON lines + sort keys as bytes, to rank 0
    from mspa import pipeline as PL
    mine = {"head_a": [{"id": f"a{rank}_{k}", "v": [rank, k, 0.1 * k]} for k in range(2 + rank)]}
    if rank == 1:
        mine["head_b"] = [{"id": 7, "text": "caf\u00e9 \n two"}]            # a file only one rank contributes to; non-ASCII, newline
    parts = S.gather_bytes(PL._pack_outputs(mine), ctx, dst=0)
    got = None
    if rank == 0:
        merged = {}
        for p in parts:
            PL._unpack_outputs(p, merged)
        got = {name: sorted((k, ln.decode()) for k, ln in v) for name, v in merged.items()}
    # a generator state from rank 0 to everybody (the object-movement builders' writer keeps the ranks' `random` in step)
    import random
    random.seed(100 + rank)
    random.setstate(S.broadcast_object(random.getstate(), ctx, src=0))
    drawn = random.random()
    # a rank-local failure reaches every rank before the next collective: the failing rank raises its own error
    raised = []
    for failing in (None, 1):
        try:
            S.raise_together(ctx, KeyError("frame 00015") if rank == failing else None, "pass 1")
            raised.append(None)
        except Exception as e:
            raised.append(type(e).__name__)
    ctx.barrier()
    q.put((rank, full.numpy().tolist(), t, got, drawn, raised))
    ctx.close()


def test_shard_and_collate_gloo_world2():
    import torch.multiprocessing as mp
    from oracle import np_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = synth.make_scene(31, n_points=16, n_frames=5, color_hw=(24, 32), depth_hw=(24, 32), invalid_pose_frac=0,
                          with_color=False)
    ids = sc.valid_image_ids
    all_pairs = [(i, j) for i in range(len(ids)) for j in range(len(ids)) if i != j][:13]
    col = np.zeros((24, 32, 3), np.uint8)
    expect = []
    for (i, j) in all_pairs:
        r = O.frame_pair(sc.depth[ids[i]], sc.depth[ids[j]], sc.K, sc.E[ids[i]], sc.E[ids[j]], sc.A, (24, 32), col)
        expect.append([i, j, r["n_valid"], r["n_vis"]])
    import random
    random.seed(100)
    want_draw = random.random()
    for rank, full, t, tapes, drawn, raised in results:
        assert full == expect, f"rank {rank}: collated records differ from the single-process table"
        assert t == 2.0 and drawn == want_draw
        assert raised == [None, "KeyError" if rank == 1 else "RuntimeError"]
        if rank == 0:
            import json
            want_a = sorted((f"a{r}_{k}", json.dumps({"id": f"a{r}_{k}", "v": [r, k, 0.1 * k]})) for r in range(2) for k in range(2 + r))
            assert tapes == {"head_a": want_a, "head_b": [("7", json.dumps({"id": 7, "text": "caf\u00e9 \n two"}))]}
        else:
            assert tapes is None


def test_correspondences_rowmajor_view_on_host_tensors():
    """engine.correspondences_rowmajor is indexing only (torch as plumbing): exercised here on CPU tensors laid out the way
    mspa_pair_correspondences lays them out, ragged tiles included."""
    from mspa import engine, _lib
    rng = np.random.default_rng(3)
    H, W = 100, 150                                           # 3 stripes x 3 bands, ragged at both edges
    ns, nb = engine.corr_tiles((H, W))
    assert (ns, nb) == (3, 3)
    vis = rng.random((H, W)) < 0.2
    xi = rng.integers(0, W, (H, W)).astype(np.int16)
    yi = rng.integers(0, H, (H, W)).astype(np.int16)
    bits = np.packbits(np.concatenate([vis.reshape(-1), np.zeros((-H * W) % 64, bool)]), bitorder="little").view(np.int64)
    cpix = np.full((1, ns * nb, _lib.CORR_TILE_CAP, 2), -7, dtype=np.int16)
    counts = np.zeros((1, ns * nb), dtype=np.int32)
    for b in range(nb):
        for s_ in range(ns):
            sl = (slice(b * 48, (b + 1) * 48), slice(s_ * 64, (s_ + 1) * 64))
            m = vis[sl]
            t = b * ns + s_
            counts[0, t] = m.sum()
            cpix[0, t, :m.sum(), 0], cpix[0, t, :m.sum(), 1] = xi[sl][m], yi[sl][m]
    out = {"vis_bits": torch.from_numpy(bits[None]), "cpix": torch.from_numpy(cpix), "tile_counts": torch.from_numpy(counts)}
    i, gx, gy = engine.correspondences_rowmajor(out, (H, W), 0)
    nz = np.nonzero(vis.reshape(-1))[0]
    assert np.array_equal(i.numpy(), nz) and np.array_equal(gx.numpy(), xi.reshape(-1)[nz]) and np.array_equal(gy.numpy(), yi.reshape(-1)[nz])


def test_batched_matrix_preparation_is_bit_identical_to_per_frame_numpy():
    """engine.frame_matrices / camera_matrices batch their LAPACK / BLAS calls; every entry must equal what the reference's
    per-frame expressions give (np.linalg.inv(A @ E), IH:57, 113-124; inv(K), OPS:313), bit for bit."""
    rng = np.random.default_rng(5)
    sc = synth.make_scene(777, n_points=8, n_frames=40, color_hw=(12, 16), depth_hw=(12, 16), invalid_pose_frac=0.0,
                          with_color=False)
    E = [sc.E[i] for i in sc.image_ids] + [np.eye(4)]
    fm = engine.frame_matrices(sc.K, sc.A, E)
    cm = engine.camera_matrices(sc.K, [sc.A @ e for e in E])
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    Kinv = np.linalg.inv(sc.K)
    for f, e in enumerate(E):
        AE = sc.A @ e
        inv = np.linalg.inv(AE)
        assert np.array_equal(bits(fm[f, _lib.MAT_EINV_ALIGNED]), bits(inv.reshape(16)))
        assert np.array_equal(bits(fm[f, _lib.MAT_KINV]), bits(Kinv.reshape(16)))
        assert np.array_equal(bits(fm[f, _lib.MAT_E]), bits(np.asarray(e, np.float64).reshape(16)))
        assert np.array_equal(bits(fm[f, _lib.MAT_UNPROJ]), bits((sc.A @ e @ Kinv).reshape(16)))
        assert np.array_equal(bits(fm[f, _lib.MAT_REPROJ]), bits((sc.K @ inv).reshape(16)))
        assert np.array_equal(bits(cm[f, 0]), bits(inv.reshape(16)))
    bad = [e.copy() for e in E]
    bad[3][3, 1] = 1e-9
    with pytest.raises(ValueError, match=r"E\[3\]"):
        engine.frame_matrices(sc.K, sc.A, bad)
    assert engine.frame_matrices(sc.K, sc.A, []).shape == (0, _lib.FRAME_MATS, 16)


def test_gather_blocks_host_matches_per_frame_copies():
    """mspa_gather_blocks_host (the staging copy of mspa/upload.py) against np.copyto frame by frame: any thread count, more
    threads than blocks, repeated source blocks, zero blocks; wrong-sized and non-contiguous blocks are refused."""
    from mspa import engine
    rng = np.random.default_rng(11)
    src = [rng.integers(0, 65535, (48, 64), dtype=np.uint16) for _ in range(5)]
    blocks = [src[k % 5] for k in range(23)]
    for nt in (1, 3, 8, 64):
        dst = np.zeros((25, 48, 64), np.uint16)
        engine.gather_blocks_host(blocks, dst, nt)
        assert all((dst[k] == blocks[k]).all() for k in range(23)) and not dst[23:].any()
    engine.gather_blocks_host([], np.zeros((0, 48, 64), np.uint16), 4)
    as_i16 = np.zeros((23, 48, 64), np.int16)                      # the pinned stack is int16 holding the same bits
    engine.gather_blocks_host(blocks, as_i16, 2)
    assert (as_i16.view(np.uint16)[7] == blocks[7]).all()
    with pytest.raises(ValueError):
        engine.gather_blocks_host([src[0][:10]], np.zeros((1, 48, 64), np.uint16))
    with pytest.raises(ValueError):
        engine.gather_blocks_host([src[0].T.copy().T], np.zeros((1, 48, 64), np.uint16))
    with pytest.raises(ValueError):
        engine.gather_blocks_host(blocks, np.zeros((3, 48, 64), np.uint16))


def _pose_cases(seed=21, n=400):
    rng = np.random.default_rng(seed)
    Es = []
    for _ in range(n):
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        e = np.eye(4)
        e[:3, :3] = q * rng.uniform(0.5, 2.0)          # not exactly orthonormal: the norm matters
        e[:3, 3] = rng.normal(size=3) * 3
        Es.append(e)
    for z in ([0, 0, 1], [0, 0, -1], [1, 0, 0], [0, -1, 0], [1e-300, 0, 1], [-1, 1e-17, 0]):
        e = np.eye(4)
        e[:3, 2] = z
        Es.append(e)
    return Es


def check_host_pose_prep_bitwise():
    """The vectorised host preparation (angles over all frames at once, one batched A @ E, one isfinite pass) against the
    oracle's literal per-frame forms, bit for bit."""
    from oracle import np_oracle as O
    from mspa.scene import valid_image_ids
    Es = _pose_cases()
    bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.int64)
    yaw, pitch = engine.extract_yaw_pitch_host(Es)
    ref = [O.extract_yaw_pitch(e) for e in Es]
    assert np.array_equal(bits(yaw), bits(np.array([r[0] for r in ref])))
    assert np.array_equal(bits(pitch), bits(np.array([r[1] for r in ref])))
    assert engine.extract_yaw_pitch_host([])[0].shape == (0,)
    A = np.eye(4)
    A[:3, :3] = np.linalg.qr(np.random.default_rng(2).normal(size=(3, 3)))[0]
    A[:3, 3] = [0.3, -1.2, 0.05]
    batched = np.matmul(A, np.stack(Es))
    assert all(np.array_equal(bits(batched[k]), bits(A @ Es[k])) for k in range(len(Es)))
    E = {f"f{k}": e.copy() for k, e in enumerate(Es[:50])}
    E["f7"][1, 2] = np.inf
    E["f31"][0, 0] = np.nan
    E["f40"] = E["f40"].astype(np.float32)
    assert valid_image_ids(E) == O.valid_image_ids(E) and len(valid_image_ids(E)) == 48
    assert valid_image_ids({}) == []


def test_host_pose_prep_is_bit_identical_to_per_frame_numpy():
    check_host_pose_prep_bitwise()
