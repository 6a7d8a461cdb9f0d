import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]
from mspa import engine, synth, _lib
dev = "cuda"
for color_hw in [(480, 640), (968, 1296)]:
    sc = synth.make_scene(1000, n_points=64, n_frames=8, color_hw=color_hw, depth_hw=(480, 640), invalid_pose_frac=0, with_color=False)
    ids = sc.valid_image_ids
    reps = 32
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids] * reps), dev)
    mats = torch.from_numpy(np.tile(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids]), (reps, 1, 1))).to(dev)
    rng = np.random.default_rng(0)
    n = 200
    b1 = rng.integers(0, len(ids), n); b2 = (b1 + rng.integers(1, 4, n)) % len(ids); rep = np.arange(n) % reps
    pairs = torch.from_numpy(np.stack([rep * len(ids) + b1, rep * len(ids) + b2], 1).astype(np.int32)).to(dev)
    for flags, name in ((_lib.PAIR_FAST, "fast"), (0, "exact")):
        out = engine.alloc_pair_outputs(n, color_hw, ("vis_bits", "pix_i16", "counts"), dev)
        for _ in range(2):
            engine.pair_reproject(depth, mats, pairs, color_hw, out, flags=flags)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            engine.pair_reproject(depth, mats, pairs, color_hw, out, flags=flags)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        P = color_hw[0] * color_hw[1]
        print(color_hw, name, "ms per 1000 pairs %.3f" % (ms / n * 1000), "Mpx/s %.0f" % (n * P / ms / 1e3), "vis frac %.3f" % (out["counts"][:, 1].sum().item() / max(1, out["counts"][:, 0].sum().item())))
