#!/usr/bin/env python3
"""ScanNet's own shape (1296x968 colour over 640x480 depth), one process, interleaved rounds: the wobbling-stripe kernel
(correspondence / minimal sets, MSPA_PAIR_WORD_STRIPES), the rectangular-tile kernel on the same sets (the default) and the fused compacted
set (rectangular tiles), ms per 1 000 pairs -- the rectangular-tile legs once per library (tools/ab/libmspa_*.so + in-tree).
    python tools/ab_scannet.py [--pairs 200] [--steps 20] [--rounds 3]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=200)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--legs", default="", help="comma list of substrings: only the legs whose name contains one of them (PMC runs)")
    a = ap.parse_args()
    import torch
    from mspa import _lib, engine, synth, workload
    dev = torch.device("cuda", 0)
    CH, CW, DH, DW = 968, 1296, 480, 640
    sc = synth.make_scene(1000, n_points=32768, n_frames=16, color_hw=(CH, CW), depth_hw=(DH, DW), invalid_pose_frac=0.0,
                          with_color=False, trajectory="sweep", walk_step=0.08, target_step=0.25)
    ids = sc.valid_image_ids
    nb = len(ids)
    cam = torch.from_numpy(engine.camera_matrices(sc.K, [sc.A @ sc.E[i] for i in ids])).to(dev)
    d_base = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), dev)
    xyz = torch.from_numpy(np.ascontiguousarray(sc.points[:, :3])).to(dev)
    overlap = engine.scene_overlap(engine.vertex_visibility(xyz, cam, d_base, (CH, CW), ("bits",))["bits"]).cpu().numpy()
    base, info = workload.select_pairs(overlap, nb, a.pairs, "vc", seed=77)
    reps = 8
    depth = d_base.repeat(reps, 1, 1).contiguous()
    mats = torch.from_numpy(np.tile(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids]), (reps, 1, 1))).to(dev)
    rep = (np.arange(a.pairs) % reps).astype(np.int32)
    pairs = torch.from_numpy(np.stack([rep * nb + base[:, 0], rep * nb + base[:, 1]], 1).astype(np.int32)).to(dev)
    F = _lib.PAIR_FAST | _lib.PAIR_STREAM
    outs = {"corr": engine.alloc_pair_outputs(a.pairs, (CH, CW), ("vis_bits", "pix_i16", "counts"), dev),
            "minimal": engine.alloc_pair_outputs(a.pairs, (CH, CW), ("vis_bits", "counts"), dev)}
    comp = engine.alloc_pair_correspondences(a.pairs, (CH, CW), dev)
    import ctypes
    import glob
    libs = sorted(glob.glob(os.path.join(ROOT, "tools/ab/libmspa_*.so"))) + [_lib.LIB_PATH]
    handles = {}
    for path in libs:
        h = ctypes.CDLL(path)
        for name in ("mspa_pair_reproject", "mspa_pair_correspondences"):
            fn = getattr(h, name)
            fn.restype, fn.argtypes = _lib._SIGNATURES[name]
        handles[os.path.basename(path)[7:-3].lstrip("_") or "in-tree"] = h
    stream = torch.cuda.current_stream().cuda_stream
    nF = depth.shape[0]

    def launch(h, which, flags):
        if which == "compact":
            g = lambda k: comp[k].data_ptr() if k in comp else None
            rc = h.mspa_pair_correspondences(depth.data_ptr(), mats.data_ptr(), nF, pairs.data_ptr(), a.pairs, DH, DW, CH, CW, g("vis_bits"),
                                             g("cpix"), g("tile_counts"), g("counts"), None, 0, flags, stream)
        else:
            o = outs[which]
            g = lambda k: o[k].data_ptr() if k in o else None
            rc = h.mspa_pair_reproject(depth.data_ptr(), None, mats.data_ptr(), nF, pairs.data_ptr(), a.pairs, DH, DW, CH, CW,
                                       g("vis_bits"), None, None, g("pix_i16"), None, None, None, None, None, g("counts"), flags, stream)
        if rc:
            raise RuntimeError(f"rc {rc}")

    legs = {}
    for lname, h in handles.items():
        tag = "" if len(handles) == 1 else "@" + lname
        legs["corr:rect" + tag] = (h, "corr", F)
        legs["minimal:rect" + tag] = (h, "minimal", F)
        legs["compact:rect" + tag] = (h, "compact", F)
    h0 = handles[list(handles)[-1]]
    legs["corr:wobble"] = (h0, "corr", F | _lib.PAIR_WORD_STRIPES)
    legs["minimal:wobble"] = (h0, "minimal", F | _lib.PAIR_WORD_STRIPES)
    if a.legs:
        legs = {k: v for k, v in legs.items() if any(t in k for t in a.legs.split(","))}
    for k in list(legs):                     # an older library may not take a set at this shape (rc != 0): drop the leg
        try:
            launch(*legs[k])
        except RuntimeError as e:
            print(f"# {k}: not available in this library ({e})")
            del legs[k]
    torch.cuda.synchronize()
    res = {k: [] for k in legs}
    for _ in range(a.rounds):
        for k, (h, which, fl) in legs.items():
            if os.environ.get("MSPA_AB_TRACE"):
                print("leg", k, file=sys.stderr, flush=True)
            for _ in range(3):
                launch(h, which, fl)
            if os.environ.get("MSPA_AB_TRACE"):
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.steps):
                launch(h, which, fl)
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / a.steps * 1000.0 / a.pairs)
    c = outs["corr"]["counts"].cpu().numpy()
    print(f"# ScanNet shape, {a.pairs} pairs ({info['rule']}), visible fraction {c[:, 1].sum() / max(1, c[:, 0].sum()):.3f}; ms per 1 000 pairs (median of {a.rounds} rounds x {a.steps} launches)")
    for k, v in res.items():
        print(f"{k:28s} {np.median(v):.4f} (+-{(max(v) - min(v)) / 2:.4f})")


if __name__ == "__main__":
    main()
