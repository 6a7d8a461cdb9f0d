#!/usr/bin/env python3
"""Randomised parity sweep on the GPU, outside the test suite: fast kernels (every output set, plain and streaming, the fused
compacted set) against the exact kernel -- which the suite pins bit-for-bit to the C / NumPy oracles and the reference's frozen
outputs -- on ALL ordered pairs of freshly drawn adversarial + hand-held poses per seed, at the BASELINE shape (640x480) and at
ScanNet's own shape (1296x968 colour over 640x480 depth).

    python tools/fuzz_parity.py [--seconds 150] [--frames 20] [--seed0 1]
Prints a markdown summary (committed as profiles/r03_fuzz_parity.md): pairs and pixels compared, mismatching pairs (must be 0).
"""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT, os.path.join(ROOT, "tests")]

SETS = {"corr": ("vis_bits", "pix_i16", "counts"), "minimal": ("vis_bits", "counts"),
        "dense_xyz": ("vis_u8", "pix_i16", "xyz_f32", "counts"), "dense": ("vis_u8", "pix_i16", "xyz_f32", "rgba", "counts")}


def _poses(seed, n, hw):
    import adversarial
    return adversarial.fuzz_poses(seed, n, hw)


def _render(job):
    import adversarial
    return adversarial.fuzz_depth(job)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=150.0)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--seed0", type=int, default=1)
    a = ap.parse_args()
    import torch
    from mspa import engine, _lib
    dev = "cuda"
    shapes = [("640x480", (480, 640), (480, 640), a.frames), ("scannet 1296x968 / 640x480", (968, 1296), (480, 640), max(6, a.frames // 2))]
    tot = {s[0]: {"pairs": 0, "pixels": 0, "launch_sets": 0, "bad": []} for s in shapes}
    t_end = time.time() + a.seconds
    seed = a.seed0
    pool = Pool(min(32, os.cpu_count() or 4))
    while time.time() < t_end:
        for name, hw, dhw, nf in shapes:
            K, A, E, boxes = _poses(seed, nf, hw)
            Kd = K.copy()
            Kd[0] *= dhw[1] / hw[1]
            Kd[1] *= dhw[0] / hw[0]
            depth_np = pool.map(_render, [(seed, k, A @ e, Kd, dhw, boxes) for k, e in enumerate(E)])
            depth = engine.depth_to_device(np.stack(depth_np), dev)
            mats = torch.from_numpy(engine.frame_matrices(K, A, E)).to(dev)
            g = torch.Generator(device=dev)
            g.manual_seed(seed)
            rgb = torch.randint(0, 256, (nf,) + hw + (3,), generator=g, device=dev, dtype=torch.uint8)
            idx = torch.arange(nf, device=dev, dtype=torch.int32)
            pairs = torch.stack([idx.repeat_interleave(nf), idx.repeat(nf)], 1).contiguous()
            n = pairs.shape[0]
            tight = hw == dhw
            sets = SETS if tight else {k: SETS[k] for k in ("corr", "minimal")}
            for sname, outs in sets.items():
                exact = engine.alloc_pair_outputs(n, hw, outs, dev)
                engine.pair_reproject(depth, mats, pairs, hw, exact, rgb=rgb if "rgba" in outs else None, flags=0)
                for stream in (0, _lib.PAIR_STREAM):
                    fast = engine.alloc_pair_outputs(n, hw, outs, dev)
                    for t in fast.values():
                        t.fill_(23)
                    engine.pair_reproject(depth, mats, pairs, hw, fast, rgb=rgb if "rgba" in outs else None, flags=_lib.PAIR_FAST | stream)
                    kern = _lib.load().mspa_pair_reproject_last_kernel()
                    assert kern == (_lib.KERNEL_PAIR_FAST_TIGHT if tight else _lib.KERNEL_PAIR_FAST_RECT), kern
                    for k in outs:
                        if k == "xyz_f32":
                            fe, ff = exact[k], fast[k]
                            ok = (torch.isnan(fe) == torch.isnan(ff)) & (torch.isnan(fe) | ((fe - ff).abs() <= 2e-7 * fe.abs() + 1e-7))
                            badp = (~ok).reshape(n, -1).any(1)
                        else:
                            badp = (fast[k] != exact[k]).reshape(n, -1).any(1)
                        if bool(badp.any()):
                            tot[name]["bad"].append((seed, sname, "stream" if stream else "plain", k, badp.nonzero().flatten()[:4].tolist()))
                    tot[name]["launch_sets"] += 1
                    del fast
                del exact
            if tight:                                  # the fused compacted set against exact kernel + stand-alone compaction
                ws_flags = 0
                ref = engine.alloc_pair_correspondences(n, hw, dev)
                engine.pair_correspondences(depth, mats, pairs, hw, ref, flags=ws_flags)
                for stream in (0, _lib.PAIR_STREAM):
                    out = engine.alloc_pair_correspondences(n, hw, dev)
                    out["cpix"].fill_(-7)
                    engine.pair_correspondences(depth, mats, pairs, hw, out, flags=_lib.PAIR_FAST | stream)
                    assert _lib.load().mspa_pair_reproject_last_kernel() == _lib.KERNEL_PAIR_FAST_TIGHT
                    for k in ("vis_bits", "tile_counts", "counts"):
                        badp = (out[k] != ref[k]).reshape(n, -1).any(1)
                        if bool(badp.any()):
                            tot[name]["bad"].append((seed, "compact", "stream" if stream else "plain", k, badp.nonzero().flatten()[:4].tolist()))
                    cap = out["cpix"].shape[2]
                    live = torch.arange(cap, device=dev)[None, None, :] < ref["tile_counts"][:, :, None].to(torch.int64)
                    badp = ((out["cpix"] != ref["cpix"]).any(-1) & live).reshape(n, -1).any(1)
                    if bool(badp.any()):
                        tot[name]["bad"].append((seed, "compact", "stream" if stream else "plain", "cpix", badp.nonzero().flatten()[:4].tolist()))
                    tot[name]["launch_sets"] += 1
                    del out
                del ref
            torch.cuda.synchronize()
            tot[name]["pairs"] += n
            tot[name]["pixels"] += n * hw[0] * hw[1]
        seed += 1
    pool.close()
    print("# Randomised parity sweep (tools/fuzz_parity.py): fast kernels vs the exact kernel on the GPU\n")
    print(f"seeds {a.seed0}..{seed - 1}, {a.frames} poses per seed at 640x480 ({a.frames // 2} adversarial: looking away, coincident, "
          "half-pixel shifts, grazing corners; the rest a hand-held walk), every ORDERED pair incl. identity pairs; depth = rendered "
          "room + 4 mm noise + 7 % invalid pixels + a large hole in every fifth frame.  Output sets: corr, minimal, dense_xyz, dense "
          "(plain and MSPA_PAIR_STREAM), the fused compacted set against exact kernel + mspa_compact_correspondences; ScanNet shape: "
          "corr, minimal.  Integers bit-exact, float32 points to 2e-7 relative.\n")
    print("| shape | distinct ordered pairs | pixels per set | set launches compared | mismatching (seed, set, mode, output, pairs) |")
    print("|---|---|---|---|---|")
    for name, t in tot.items():
        print(f"| {name} | {t['pairs']} | {t['pixels']:.3e} | {t['launch_sets']} | {t['bad'] if t['bad'] else 'none'} |")
    if any(t["bad"] for t in tot.values()):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
