#!/usr/bin/env python3
"""The two split-sweeping entry points on ScanNet-sized on-disk scenes (320 frames of 640x480 depth PNGs, 131 072 vertices),
one GPU: calculate_frames_relations.run_split (pair table) and make_visibility_info.run_split (visibility index, parquet).
    python tools/dropin_big.py [--scenes 6] [--frames 320] [--workers 25]"""
import argparse, contextlib, io, json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=6)
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--workers", type=int, default=25)
    a = ap.parse_args()
    import bench
    from mspa import synth, sweep
    import spatial_engine.camera_movement.calculate_frames_relations as CFR
    import spatial_engine.utils.scannet_utils.make_visibility_info as MVI
    from spatial_engine.utils.scannet_utils.handler import info_handler as IH
    root = tempfile.mkdtemp(prefix="mspa_dropin_big_")
    try:
        t0 = time.perf_counter()
        paths = synth.write_scannet_layout(bench._disk_scenes(a.scenes, a.frames, 131072), root, compress_level=6)
        print(f"# inputs written in {time.perf_counter() - t0:.1f} s: {a.scenes} scenes x {a.frames} frames, {os.cpu_count()} host cores, num_workers {a.workers}")
        orig = IH.SceneInfoHandler.__init__

        def init(self, info_path, *x, **k):
            orig(self, info_path, posed_images_root=paths["posed_images_root"], instance_data_root=paths["instance_data_root"])
        IH.SceneInfoHandler.__init__ = init
        for name, fn, out in (("calculate_frames_relations.run_split", CFR.run_split, "pairs.parquet"),
                              ("make_visibility_info.run_split", MVI.run_split, "vis.parquet")):
            for rep in range(2):
                tm = sweep.Timings()
                with contextlib.redirect_stdout(io.StringIO()):
                    t0 = time.perf_counter()
                    fn(paths["info_path"], os.path.join(root, f"o{rep}", out), os.path.join(root, f"w{rep}.txt"), num_workers=a.workers,
                       keep=False, timings=tm)
                    dt = time.perf_counter() - t0
                size = os.path.getsize(os.path.join(root, f"o{rep}", out)) / 1e6
                print(f"{name} pass {rep}: {dt:.3f} s = {a.scenes / dt:.2f} scenes/s = {a.scenes * a.frames / dt:.0f} frames/s; output {size:.1f} MB; busy {json.dumps(tm.as_dict())}")
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
