#!/usr/bin/env python3
"""How long a 100 MB page-locked host-to-device copy takes while N inflate waves are resident (tools/device_ingest_bench.py's streams),
per N -- is the copy an engine copy that does not care, or does it compete with the decode waves for the compute units?
    [HSA_ENABLE_SDMA=0|1] python tools/micro/h2d_under_decode.py"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]
import numpy as np, torch
from PIL import Image
from mspa import engine, ingest, synth

H, W = 480, 640
root = tempfile.mkdtemp(prefix="mspa_h2d_")
try:
    sc = synth.make_scene(5000, n_points=2048, n_frames=16, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0, with_color=False)
    paths = []
    for image_id in sc.valid_image_ids:
        p = os.path.join(root, f"{image_id}.png")
        Image.fromarray(sc.depth[image_id]).save(p, compress_level=6)
        paths.append(p)
    buf, offsets, nbytes, st, cap = ingest.pack_depth_pngs(paths, H, W, 4)
    dev = torch.device("cuda", 0)
    stride = (cap + 255) // 256 * 256
    one = torch.zeros(stride, dtype=torch.uint8)
    one[:cap] = torch.from_numpy(buf[:cap])
    block = H * (2 * W + 1)
    pitch = (block + 255) // 256 * 256
    host = torch.empty(100 << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(100 << 20, dtype=torch.uint8, device=dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
    for n in (0, 1280, 2560, 3584, 4096):
        reps = max(1, n // 16)
        if n:
            src = one.to(dev).repeat(reps)
            off = torch.from_numpy(np.concatenate([offsets + r * stride for r in range(reps)])).to(dev)
            nb = torch.from_numpy(np.tile(nbytes, reps)).to(dev)
            raw = torch.empty((n, pitch), dtype=torch.uint8, device=dev)
            status = torch.empty((n,), dtype=torch.int32, device=dev)
        times, infl = [], []
        for rep in range(4):
            torch.cuda.synchronize()
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            if n:
                with torch.cuda.stream(sa):
                    e[0].record(sa)
                    engine.inflate_blocks_device(src, off, nb, block, raw, status)
                    e[1].record(sa)
            with torch.cuda.stream(sb):
                torch.cuda._sleep(2_000_000)          # ~1 ms: the decode waves are resident when the copy starts
                e[2].record(sb)
                dst.copy_(host, non_blocking=True)
                e[3].record(sb)
            torch.cuda.synchronize()
            times.append(e[2].elapsed_time(e[3]))
            if n:
                infl.append(e[0].elapsed_time(e[1]))
        print(f"{n:5d} inflate waves resident: 100 MB H2D {np.median(times):6.2f} ms ({100 * 1.048576 / np.median(times):5.1f} GB/s)" + (f"   inflate {np.median(infl):.1f} ms" if n else ""))
finally:
    shutil.rmtree(root, ignore_errors=True)
