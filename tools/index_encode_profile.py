#!/usr/bin/env python3
"""Where the visibility-index sweep's per-scene encoder time goes: K9 compaction + K10 text + download (visindex.from_bits),
arrow table (to_arrow), parquet row group (parquet_splice.encode_row_group) -- one 320-frame, 131 072-vertex scene, with the
lists' text written on the device (K10) and on the host.
    python tools/index_encode_profile.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def main():
    import torch
    from mspa import parquet_splice, synth, visindex
    from mspa.scene import SceneOnDevice
    H, W = 480, 640
    b = synth.make_scene(5000, n_points=131072, n_frames=8, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0, with_color=False)
    ids = b.image_ids
    E = {f"{5 * f:05d}": b.E[ids[f % 8]] for f in range(320)}
    depth = {f"{5 * f:05d}": b.depth[ids[f % 8]] for f in range(320)}
    scene = SceneOnDevice(b.K, b.A, E, depth, (H, W), b.points, torch.device("cuda", 0))
    bits = scene._visibility()["bits"]
    n = int(scene.xyz.shape[0])
    torch.cuda.synchronize()
    for text in (True, False):
        rows = []
        for rep in range(4):
            t0 = time.perf_counter()
            csr = visindex.from_bits(bits, scene.ids, n, text=text, indices=not text)
            t1 = time.perf_counter()
            table = csr.to_arrow("scene0000_00")
            t2 = time.perf_counter()
            blob = parquet_splice.encode_row_group(table, use_dictionary=False)
            t3 = time.perf_counter()
            rows.append((t1 - t0, t2 - t1, t3 - t2))
        r = np.median(np.array(rows[1:]), axis=0) * 1e3
        print(f"text on the {'device' if text else 'host  '}: from_bits {r[0]:6.1f} ms   to_arrow {r[1]:6.1f} ms   encode_row_group {r[2]:6.1f} ms   "
              f"(text {table.nbytes / 1e6:.1f} MB -> parquet {len(blob) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
