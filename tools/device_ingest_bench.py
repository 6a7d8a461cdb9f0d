#!/usr/bin/env python3
"""The gate of the on-device depth decode (Next 3): N real-sized depth streams in flight on the MI355X against
the host decode on the CPUs this container may use.

    python tools/device_ingest_bench.py [--streams 2560] [--base-frames 16] [--level 6] [--reps 5]

Depth frames: 640 x 480 of the synthetic room, written as 16-bit PNG by Pillow (adaptive row filters), every
stream a distinct copy in device memory.  Reported: inflate + Adler-32 (mspa_inflate_blocks_device) and un-filter
(mspa_png_unfilter_device) times from HIP events, frames/s and GB/s of decoded output; the host reader on the same files
(`mspa_read_depth_png_host`, table-driven inflate) with as many threads as the CPU quota allows; and the composed
`ingest.read_depth_frames_device` on one 320-frame scene including file reads and the H2D copy.
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def run(streams=2560, base_frames=16, level=6, reps=5, with_host=True, with_composed=True, smooth=False):
    """One measurement -> dict (bench.py's `variants.device_decode` leg calls this in-process).  ``smooth``: the frames without
    their sensor noise (a 9 x 9 box filter over the rendered depth) -- an adaptive PNG writer then picks the Paeth filter for
    practically every row, as it does for real sensor depth of smooth surfaces; the noisy frames get Sub / Up rows."""
    a = argparse.Namespace(streams=streams, base_frames=base_frames, level=level, reps=reps)
    import torch
    from PIL import Image
    from mspa import engine, hostinfo, ingest, synth
    H, W = 480, 640
    root = tempfile.mkdtemp(prefix="mspa_devingest_")
    try:
        sc = synth.make_scene(5000, n_points=2048, n_frames=a.base_frames, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0,
                              with_color=False)
        paths = []
        for image_id in sc.valid_image_ids:
            p = os.path.join(root, f"{image_id}.png")
            if smooth:
                from scipy.ndimage import uniform_filter
                sc.depth[image_id] = uniform_filter(sc.depth[image_id].astype(np.float64), 9).astype(np.uint16)
            Image.fromarray(sc.depth[image_id]).save(p, compress_level=a.level)
            paths.append(p)
        nb_files = len(paths)
        png_bytes = sum(os.path.getsize(p) for p in paths)
        buf, offsets, nbytes, st, cap = ingest.pack_depth_pngs(paths, H, W, 4)
        assert (st == 0).all()
        reps = -(-a.streams // nb_files)
        n = reps * nb_files
        dev = torch.device("cuda", 0)
        stride = (cap + 255) // 256 * 256
        one = torch.zeros(stride, dtype=torch.uint8)
        one[:cap] = torch.from_numpy(buf[:cap])
        src = one.to(dev).repeat(reps)                                        # every stream its own bytes in HBM
        off = torch.from_numpy(np.concatenate([offsets + r * stride for r in range(reps)])).to(dev)
        nb = torch.from_numpy(np.tile(nbytes, reps)).to(dev)
        block = H * (2 * W + 1)
        pitch = (block + 255) // 256 * 256
        raw = torch.empty((n, pitch), dtype=torch.uint8, device=dev)
        status = torch.empty((n,), dtype=torch.int32, device=dev)
        out = torch.empty((n, H, W), dtype=torch.int16, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        t_inf, t_unf = [], []
        for rep in range(a.reps + 1):
            ev[0].record()
            engine.inflate_blocks_device(src, off, nb, block, raw, status)
            ev[1].record()
            engine.png_unfilter_device(raw, H, W, status, out)
            ev[2].record()
            torch.cuda.synchronize()
            if rep:
                t_inf.append(ev[0].elapsed_time(ev[1]))
                t_unf.append(ev[1].elapsed_time(ev[2]))
        bad = int((status != 0).sum())
        want = np.stack([sc.depth[i] for i in sc.valid_image_ids])
        got = out[:nb_files].cpu().numpy().view(np.uint16)
        same = bool(np.array_equal(got, want)) and bool(torch.equal(out[-nb_files:], out[:nb_files]))
        ms_inf, ms_unf = float(np.median(t_inf)), float(np.median(t_unf))
        out_bytes = n * H * W * 2
        res = {"streams": n, "frames": "smooth (Paeth rows)" if smooth else "noisy (Sub / Up rows)", "distinct_frames": nb_files, "png_level": a.level, "png_bytes_per_frame": png_bytes // nb_files,
               "status_nonzero": bad, "bit_identical_to_the_rendered_frames": same,
               "device": {"inflate_adler_ms": round(ms_inf, 3), "unfilter_ms": round(ms_unf, 3),
                          "frames_per_s": round(n / ((ms_inf + ms_unf) * 1e-3)), "GBps_out": round(out_bytes / ((ms_inf + ms_unf) * 1e-3) / 1e9, 2),
                          "inflate_only_frames_per_s": round(n / (ms_inf * 1e-3)),
                          "compressed_GBps_in": round(float(nb.sum()) / ((ms_inf) * 1e-3) / 1e9, 2)}}
        if not with_host:
            return res
        # the host decode on the same files, with the threads the quota allows
        threads = hostinfo.effective_cpus()
        many = (paths * (-(-320 // nb_files)))[:320]
        dst = np.empty((len(many), H, W), dtype=np.uint16)
        ingest.read_depth_frames(many, threads, out=dst)
        t = []
        for _ in range(5):
            t0 = time.perf_counter()
            ingest.read_depth_frames(many, threads, out=dst)
            t.append(time.perf_counter() - t0)
        th = float(np.median(t))
        res["host"] = {"threads": threads, "host_cpus": hostinfo.describe(), "frames": len(many), "ms": round(th * 1e3, 2),
                       "frames_per_s": round(len(many) / th), "GBps_out": round(len(many) * H * W * 2 / th / 1e9, 2)}
        res["device_over_host"] = round(res["device"]["frames_per_s"] / res["host"]["frames_per_s"], 2)
        if not with_composed:
            return res
        # the composed reader on a 320-frame scene: file reads + pack + H2D + kernels + status read-back
        ingest.read_depth_frames_device(many, dev, threads)
        t = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d = ingest.read_depth_frames_device(many, dev, threads)
            torch.cuda.synchronize()
            t.append(time.perf_counter() - t0)
        tc = float(np.median(t))
        res["composed_320_frame_scene"] = {"ms": round(tc * 1e3, 2), "frames_per_s": round(len(many) / tc),
                                          "equals_host": bool(np.array_equal(d.cpu().numpy().view(np.uint16), dst))}
        return res
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=2560)
    ap.add_argument("--base-frames", type=int, default=16)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--smooth", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.streams, a.base_frames, a.level, a.reps, smooth=a.smooth)))


if __name__ == "__main__":
    main()
