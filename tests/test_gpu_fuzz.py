"""Randomised parity sweep under the driver's eyes and against the ORACLE.

``tools/fuzz_parity.py`` (78 800 pairs in round 3) compares the fast kernels with the exact kernel -- HIP with HIP, outside the
suite.  This is the same generator as a time-boxed ``-m gpu`` test: per seed, fresh adversarial + hand-held poses at the
BASELINE shape (640x480) and at ScanNet's own shape (1296x968 colour over 640x480 depth), every ordered pair through every
fast output set (plain and streaming) and the compacted set; a seeded SUBSAMPLE of the pairs is compared with
``oracle.np_oracle.frame_pair`` output by output (the NumPy oracle takes 0.3 - 1.5 s per pair at these shapes), all of them
with the exact kernel.  Seeds keep coming until the time box (MSPA_FUZZ_SECONDS, default 45 s) is used up; at least one seed
of each shape always runs.
"""
import os
import time

import numpy as np
import pytest
import torch

import adversarial as ADV
from mspa import engine, _lib
from oracle import np_oracle as O
from test_gpu_compact import check_pair as check_compact_pair
from test_gpu_guard import check_integers
from test_gpu_tight import SETS

DEV = "cuda"
BUDGET = float(os.environ.get("MSPA_FUZZ_SECONDS", "45"))


def one_seed(seed, hw, dhw, n_frames, n_oracle):
    K, A, E, boxes = ADV.fuzz_poses(seed, n_frames, hw)
    Kd = ADV.depth_intrinsics(K, hw, dhw)
    depth_np = [ADV.fuzz_depth((seed, k, A @ e, Kd, dhw, boxes)) for k, e in enumerate(E)]
    depth = engine.depth_to_device(np.stack(depth_np), DEV)
    mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    rgb = torch.randint(0, 256, (n_frames,) + tuple(hw) + (3,), generator=g, device=DEV, dtype=torch.uint8)
    idx = torch.arange(n_frames, device=DEV, dtype=torch.int32)
    pairs = torch.stack([idx.repeat_interleave(n_frames), idx.repeat(n_frames)], 1).contiguous()
    n = pairs.shape[0]
    rng = np.random.default_rng(seed)
    pick = sorted(rng.choice(n, size=min(n_oracle, n), replace=False).tolist())
    refs = {k: O.frame_pair(depth_np[k // n_frames], depth_np[k % n_frames], K, E[k // n_frames], E[k % n_frames], A, hw) for k in pick}
    tight = tuple(hw) == tuple(dhw)
    sets = list(SETS) if tight else ["corr", "minimal"]
    want = _lib.KERNEL_PAIR_FAST_TIGHT if tight else _lib.KERNEL_PAIR_FAST_RECT
    for name in sets:
        outs = SETS[name]
        exact = engine.alloc_pair_outputs(n, hw, outs, DEV)
        engine.pair_reproject(depth, mats, pairs, hw, exact, rgb=rgb if "rgba" in outs else None, flags=0)
        for stream in (0, _lib.PAIR_STREAM):
            fast = engine.alloc_pair_outputs(n, hw, outs, DEV)
            for t in fast.values():
                t.fill_(23)
            engine.pair_reproject(depth, mats, pairs, hw, fast, rgb=rgb if "rgba" in outs else None, flags=_lib.PAIR_FAST | stream)
            assert _lib.load().mspa_pair_reproject_last_kernel() == want
            for k in outs:
                if k == "xyz_f32":
                    fe, ff = exact[k], fast[k]
                    ok = (torch.isnan(fe) == torch.isnan(ff)) & (torch.isnan(fe) | ((fe - ff).abs() <= 2e-7 * fe.abs() + 1e-7))
                    assert bool(ok.all()), f"seed {seed} {name}: float32 points differ from the exact kernel"
                else:
                    bad = (fast[k] != exact[k]).reshape(n, -1).any(1)
                    assert not bool(bad.any()), f"seed {seed} {name}/{k}: pairs {bad.nonzero().flatten()[:6].tolist()} differ from the exact kernel"
            res = {k: fast[k][pick].cpu().numpy() for k in outs}
            for j, k in enumerate(pick):
                check_integers(res, j, refs[k], hw)
            del fast
        del exact
    # the fused compacted set: the tight kernel at 640x480, its rectangular-tile form at ScanNet's shape (no dense table)
    out = engine.alloc_pair_correspondences(n, hw, DEV)
    out["cpix"].fill_(-7)
    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST | _lib.PAIR_STREAM)
    assert _lib.load().mspa_pair_reproject_last_kernel() == (_lib.KERNEL_PAIR_FAST_TIGHT if tight else _lib.KERNEL_PAIR_FAST_RECT)
    torch.cuda.synchronize()
    out_np = {k: v[pick].cpu().numpy() for k, v in out.items()}
    for j, k in enumerate(pick):
        check_compact_pair(out_np, j, refs[k], hw)
    del out
    torch.cuda.synchronize()
    return n, len(pick)


@pytest.mark.gpu
def test_randomised_sweep_against_oracle_time_boxed():
    t_end = time.time() + BUDGET
    seed = int(os.environ.get("MSPA_FUZZ_SEED0", "9001"))
    done = {"640x480": [0, 0], "scannet": [0, 0]}
    first = True
    while first or time.time() < t_end:
        n, k = one_seed(seed, (480, 640), (480, 640), 10, 10)
        done["640x480"][0] += n
        done["640x480"][1] += k
        if first or time.time() < t_end:
            n, k = one_seed(seed, (968, 1296), (480, 640), 5, 3)
            done["scannet"][0] += n
            done["scannet"][1] += k
        first = False
        seed += 1
    print(f"fuzz: {done} (pairs through every fast set vs the exact kernel, pairs vs the NumPy oracle)")
    assert done["640x480"][1] >= 10 and done["scannet"][1] >= 3
