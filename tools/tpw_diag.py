#!/usr/bin/env python3
"""Diagnostic: one output set of one library at one shape, compared with the exact kernel (run in its own process)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch
from mspa import engine, synth, _lib
which, H, W, DH, DW = sys.argv[1], *[int(x) for x in sys.argv[2:6]]
hw, dhw = (H, W), (DH, DW)
sc = synth.make_scene(3030, n_points=64, n_frames=4, color_hw=hw, depth_hw=dhw, invalid_pose_frac=0.0, with_color=False, trajectory="sweep", walk_step=0.08)
ids = sc.valid_image_ids
depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), "cuda")
mats = torch.from_numpy(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])).cuda()
pairs = torch.tensor([[0, 1], [1, 0], [2, 3]], dtype=torch.int32, device="cuda")
sets = {"corr": ("vis_bits", "pix_i16", "counts"), "minimal": ("vis_bits", "counts")}
if which == "compact":
    a = engine.pair_correspondences(depth, mats, pairs, hw, flags=_lib.PAIR_FAST)
    torch.cuda.synchronize()
    b = engine.pair_correspondences(depth, mats, pairs, hw, flags=0)
    torch.cuda.synchronize()
    print(which, hw, dhw, "kernel", _lib.load().mspa_pair_reproject_last_kernel(), "bits", torch.equal(a["vis_bits"], b["vis_bits"]), "counts", torch.equal(a["counts"], b["counts"]),
          "tile_counts", torch.equal(a["tile_counts"], b["tile_counts"]))
else:
    a = engine.alloc_pair_outputs(3, hw, sets[which], "cuda")
    engine.pair_reproject(depth, mats, pairs, hw, a, flags=_lib.PAIR_FAST)
    k = _lib.load().mspa_pair_reproject_last_kernel()
    torch.cuda.synchronize()
    b = engine.alloc_pair_outputs(3, hw, sets[which], "cuda")
    engine.pair_reproject(depth, mats, pairs, hw, b, flags=0)
    torch.cuda.synchronize()
    print(which, hw, dhw, "kernel", k, {n: bool(torch.equal(a[n], b[n])) for n in a}, a["counts"].tolist(), b["counts"].tolist())
