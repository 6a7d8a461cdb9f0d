#!/usr/bin/env python3
"""Host-side ingest of one scene from disk, stage by stage (no GPU needed): native PNG read + inflate at several thread counts,
the vertex file, the handler's host_scene as a whole.  MSPA_INGEST_ZLIB=1 switches the table-driven inflate off (A/B).
    python tools/ingest_bench.py [--frames 64]"""
import argparse, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    a = ap.parse_args()
    import bench
    from mspa import ingest, synth
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    root = tempfile.mkdtemp(prefix="mspa_ingest_")
    try:
        paths = synth.write_scannet_layout(bench._disk_scenes(1, a.frames, 131072), root, compress_level=6)
        h = SceneInfoHandler(paths["info_path"], posed_images_root=paths["posed_images_root"], instance_data_root=paths["instance_data_root"])
        sid = h.get_all_scene_ids()[0]
        files = [h.get_depth_image_path(sid, i) for i in h.get_all_extrinsic_valid_image_ids(sid)]
        mb = sum(os.path.getsize(f) for f in files) / 1e6
        print(f"# {len(files)} frames, {mb:.1f} MB of PNG, {os.cpu_count()} cores, inflate = {'zlib' if os.environ.get('MSPA_INGEST_ZLIB') else 'table-driven'}")
        for n in (1, 4, 8, 16, 25, 64):
            ingest.read_depth_frames(files, n)
            t = time.perf_counter()
            for _ in range(5):
                ingest.read_depth_frames(files, n)
            dt = (time.perf_counter() - t) / 5
            print(f"read_depth_frames threads {n:3d}: {dt * 1e3:7.2f} ms per scene, {dt / len(files) * 1e3 * min(n, len(files)):.2f} ms per frame and thread, {len(files) * 614400 / dt / 1e9:.2f} GB/s out")
        buf = ingest.read_depth_frames(files, 8).copy()        # a destination that already has its pages: what a reused block costs
        for n in (8, 25, 64):
            t = time.perf_counter()
            for _ in range(5):
                ingest.read_depth_frames(files, n, out=buf)
            dt = (time.perf_counter() - t) / 5
            print(f"read_depth_frames threads {n:3d}, REUSED destination: {dt * 1e3:7.2f} ms per scene, {len(files) * 614400 / dt / 1e9:.2f} GB/s out")
        t = time.perf_counter()
        for _ in range(5):
            h.get_scene_points_align(sid)
        print(f"np.load of the vertex file: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
        t = time.perf_counter()
        for _ in range(5):
            h.get_image_shape(sid)
        print(f"image size from the JPEG header: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
        for n in (8, 25):
            t = time.perf_counter()
            for _ in range(5):
                h.host_scene(sid, n)
            print(f"host_scene(num_workers={n}): {(time.perf_counter() - t) / 5 * 1e3:.2f} ms")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
