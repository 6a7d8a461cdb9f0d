#!/usr/bin/env python3
"""Do per-scene decode launches on separate HIP streams overlap?  S scenes of F frames each: one launch of S*F streams, S launches
on one stream, S launches on S streams (optionally GPU_MAX_HW_QUEUES set by the caller).
    python tools/device_ingest_streams.py [--scenes 10] [--frames 320]"""
import argparse, json, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=10)
    ap.add_argument("--frames", type=int, default=320)
    a = ap.parse_args()
    import torch
    from PIL import Image
    from mspa import engine, ingest, synth
    H, W = 480, 640
    root = tempfile.mkdtemp(prefix="mspa_streams_")
    try:
        sc = synth.make_scene(5000, n_points=2048, n_frames=8, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0, with_color=False)
        paths = []
        for image_id in sc.valid_image_ids:
            p = os.path.join(root, f"{image_id}.png")
            Image.fromarray(sc.depth[image_id]).save(p, compress_level=6)
            paths.append(p)
        buf, offsets, nbytes, st, cap = ingest.pack_depth_pngs(paths, H, W, 4)
        reps = a.frames // len(paths)
        F = reps * len(paths)
        stride = (cap + 255) // 256 * 256
        one = torch.zeros(stride, dtype=torch.uint8)
        one[:cap] = torch.from_numpy(buf[:cap])
        block = H * (2 * W + 1)
        pitch = (block + 255) // 256 * 256
        scenes = []
        for s in range(a.scenes):
            src = one.cuda().repeat(reps)
            off = torch.from_numpy(np.concatenate([offsets + r * stride for r in range(reps)])).cuda()
            nb = torch.from_numpy(np.tile(nbytes, reps)).cuda()
            scenes.append((src, off, nb, torch.empty((F, pitch), dtype=torch.uint8, device="cuda"), torch.empty((F,), dtype=torch.int32, device="cuda"),
                           torch.empty((F, H, W), dtype=torch.int16, device="cuda")))
        big_src = torch.cat([s[0] for s in scenes])
        big_off = torch.cat([s[1] + i * scenes[0][0].numel() for i, s in enumerate(scenes)])
        big_nb = torch.cat([s[2] for s in scenes])
        big_raw = torch.empty((F * a.scenes, pitch), dtype=torch.uint8, device="cuda")
        big_st = torch.empty((F * a.scenes,), dtype=torch.int32, device="cuda")
        big_out = torch.empty((F * a.scenes, H, W), dtype=torch.int16, device="cuda")
        streams = [torch.cuda.Stream() for _ in range(a.scenes)]

        def one_launch():
            engine.inflate_blocks_device(big_src, big_off, big_nb, block, big_raw, big_st)
            engine.png_unfilter_device(big_raw, H, W, big_st, big_out)

        def same_stream():
            for (src, off, nb, raw, stt, out) in scenes:
                engine.inflate_blocks_device(src, off, nb, block, raw, stt)
                engine.png_unfilter_device(raw, H, W, stt, out)

        def many_streams():
            for st_, (src, off, nb, raw, stt, out) in zip(streams, scenes):
                with torch.cuda.stream(st_):
                    engine.inflate_blocks_device(src, off, nb, block, raw, stt)
                    engine.png_unfilter_device(raw, H, W, stt, out)

        res = {"scenes": a.scenes, "frames_per_scene": F, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
        for name, fn in (("one_launch", one_launch), ("same_stream", same_stream), ("many_streams", many_streams)):
            fn()
            torch.cuda.synchronize()
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                t.append(time.perf_counter() - t0)
            ms = float(np.median(t)) * 1e3
            res[name] = {"ms": round(ms, 2), "frames_per_s": round(F * a.scenes / ms * 1e3)}
        print(json.dumps(res))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
