#!/usr/bin/env python3
"""A/B timing of K3 builds inside ONE process on one GPU box: every tools/ab/libmspa_*.so plus the in-tree library is
loaded through its own ctypes handle, the bench's inputs (vc workload) are built once, and the output sets named on the
command line are timed interleaved, several rounds, HIP events around `--steps` back-to-back launches.

    python tools/ab_k3.py [--sets corr,compact,minimal,dense_xyz] [--steps 30] [--rounds 3] [--workload vc]
Prints one line per library: median kernel ms per set (and the spread over the rounds).
"""
import argparse
import ctypes
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="corr,compact,minimal")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--workload", default="vc")
    ap.add_argument("--repeat", type=int, default=1, help="the bench's 1 000 pairs this many times over in one launch (same mix of "
                    "overlaps: drawing MORE pairs from the sampler changes the mix); times are printed per 1 000 pairs")
    ap.add_argument("--libs", default="", help="comma list of library paths (default: tools/ab/libmspa_*.so + in-tree)")
    a = ap.parse_args()
    import torch
    from mspa import _lib, engine
    device = torch.device("cuda", 0)
    sys.argv = [sys.argv[0]]
    args = bench.parse_args()
    sc = bench.make_base_scene(args, 0)
    args.also = "dense:fast"                                   # so that rgb frames exist
    depth, mats, rgb, nb, reps = bench.build_inputs(args, 0, device, sc)
    overlap = bench.scene_overlap_table(sc, device)
    pairs_np, _, _ = bench.workload_pairs(overlap, nb, reps, args.pairs, a.workload, 0)
    pairs = torch.from_numpy(pairs_np).to(device).repeat(a.repeat, 1)
    n = pairs.shape[0]
    H, W = bench.H, bench.W
    sets = [s for s in a.sets.split(",") if s]
    outs = {}
    for s in sets:
        spec = bench.VARIANTS[s]
        outs[s] = (engine.alloc_pair_correspondences(n, (H, W), device) if spec.get("compact")
                   else engine.alloc_pair_outputs(n, (H, W), spec["outputs"], device))
    libs = [p for p in a.libs.split(",") if p] or sorted(glob.glob(os.path.join(ROOT, "tools/ab/libmspa_*.so"))) + [_lib.LIB_PATH]
    handles = {}
    for path in libs:
        h = ctypes.CDLL(path)
        for name in ("mspa_pair_reproject", "mspa_pair_correspondences"):
            if hasattr(h, name):
                fn = getattr(h, name)
                fn.restype, fn.argtypes = _lib._SIGNATURES[name]
        handles[os.path.basename(path)] = h
    flags = _lib.PAIR_FAST | _lib.PAIR_STREAM
    stream = torch.cuda.current_stream().cuda_stream
    F = depth.shape[0]

    def launch(h, s):
        o = outs[s]
        g = lambda k: o[k].data_ptr() if k in o else None
        if bench.VARIANTS[s].get("compact"):
            rc = h.mspa_pair_correspondences(depth.data_ptr(), mats.data_ptr(), F, pairs.data_ptr(), n, H, W, H, W, g("vis_bits"),
                                             g("cpix"), g("tile_counts"), g("counts"), None, 0, flags, stream)
        else:
            rc = h.mspa_pair_reproject(depth.data_ptr(), rgb.data_ptr() if "rgba" in o else None, mats.data_ptr(), F,
                                       pairs.data_ptr(), n, H, W, H, W, g("vis_bits"), g("vis_u8"), g("valid_u8"), g("pix_i16"),
                                       g("xyz_f32"), g("rgba"), g("xyz_f64"), g("uv_f64"), g("depth_f64"), g("counts"), flags, stream)
        if rc:
            raise RuntimeError(f"rc {rc}")

    res = {k: {s: [] for s in sets} for k in handles}
    for r in range(a.rounds):
        for s in sets:
            for k, h in handles.items():
                if bench.VARIANTS[s].get("compact") and not hasattr(h, "mspa_pair_correspondences"):
                    continue
                try:
                    for _ in range(3):
                        launch(h, s)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(a.steps):
                        launch(h, s)
                    e1.record()
                    torch.cuda.synchronize()
                    res[k][s].append(e0.elapsed_time(e1) / a.steps * 1000.0 / n)   # per 1 000 pairs; includes the 2 us counter memset per launch
                except Exception as ex:                                   # a timing-only build may refuse a set
                    res[k][s].append(float("nan"))
    for k in handles:
        cells = []
        for s in sets:
            v = np.array(res[k][s], dtype=float)
            cells.append(f"{s} {np.nanmedian(v):.4f} (+-{(np.nanmax(v) - np.nanmin(v)) / 2:.4f})" if len(v) and not np.all(np.isnan(v)) else f"{s} -")
        print(f"{k:<28s} " + "  ".join(cells), flush=True)


if __name__ == "__main__":
    main()
