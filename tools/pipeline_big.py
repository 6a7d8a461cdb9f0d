#!/usr/bin/env python3
"""`mspa.pipeline` over >= 64 on-disk scenes in the reference's layout (ScanNet-sized: 640 x 480 depth PNGs, 131 072 vertices):
what a rank keeps resident, what rank 0 holds, where the time goes -- one rank, then two ranks sharing the GPU (gloo), same bytes.
    python tools/pipeline_big.py [--scenes 64] [--frames 160] [--ranks 1,2] [--spill-mb 4]"""
import argparse, hashlib, json, os, shutil, socket, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT, os.path.join(ROOT, "tools")]


def worker(a):
    import torch
    from mspa import pipeline, shard
    from spatial_engine.utils.scannet_utils.handler.info_handler import SceneInfoHandler
    paths = json.load(open(os.path.join(a.root, "paths.json")))
    ctx = shard.context_from_env()
    dev = ctx.device if ctx is not None else torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    h = SceneInfoHandler(paths["info_path"], posed_images_root=paths["posed_images_root"], instance_data_root=paths["instance_data_root"])
    scenes = [pipeline.DiskScene(h, sid, a.workers) for sid in h.get_all_scene_ids()]
    world = ctx.world if ctx is not None else 1
    t0 = time.perf_counter()
    counts = pipeline.run(scenes, os.path.join(a.root, f"out_w{world}"), ctx, dev, seed=11, n_camera=2000, n_correspondence=2000,
                          depth_images_per_scene=4, object_perception=False, spill_bytes=a.spill_mb << 20)
    dt = time.perf_counter() - t0
    rank = ctx.rank if ctx is not None else 0
    json.dump({"rank": rank, "seconds": round(dt, 3), "records": counts, "timings": {k: round(v, 4) for k, v in pipeline.LAST_TIMINGS.items()}},
              open(os.path.join(a.root, f"w{world}_rank{rank}.json"), "w"))
    if ctx is not None:
        ctx.barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--root")
    ap.add_argument("--scenes", type=int, default=64)
    ap.add_argument("--frames", type=int, default=160)
    ap.add_argument("--ranks", default="1,2")
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--spill-mb", type=int, default=4)
    a = ap.parse_args()
    if a.worker:
        worker(a)
        return
    import dropin_ranks
    root = tempfile.mkdtemp(prefix="mspa_pipeline_big_")
    try:
        t0 = time.perf_counter()
        paths = dropin_ranks.write_inputs(root, a.scenes, a.frames, 131072)
        json.dump(paths, open(os.path.join(root, "paths.json"), "w"))
        res = {"scenes": a.scenes, "frames_per_scene": a.frames, "inputs_written_in_s": round(time.perf_counter() - t0, 1),
               "spill_limit_mb": a.spill_mb, "worlds": {}}
        digests = {}
        for world in [int(x) for x in a.ranks.split(",")]:
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MSPA_DIST_BACKEND="gloo")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
                env.pop(k, None)
            args = [os.path.abspath(__file__), "--worker", "--root", root, "--workers", str(a.workers), "--spill-mb", str(a.spill_mb)]
            if world == 1:
                cmd = [sys.executable] + args
            else:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                       "127.0.0.1", "--master-port", str(port)] + args
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
            if r.returncode != 0:
                res["worlds"][str(world)] = {"failed": (r.stderr or r.stdout)[-2000:]}
                continue
            per = [json.load(open(os.path.join(root, f"w{world}_rank{k}.json"))) for k in range(world)]
            out = os.path.join(root, f"out_w{world}")
            digests[world] = {n: hashlib.sha256(open(os.path.join(out, n), "rb").read()).hexdigest() for n in sorted(os.listdir(out)) if n.endswith(".jsonl")}
            res["worlds"][str(world)] = {"seconds": max(p["seconds"] for p in per), "records": per[0]["records"],
                                         "rank_timings": [p["timings"] for p in per],
                                         "files_identical_to_first_world": digests[world] == digests[min(digests)]}
        print(json.dumps(res))
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
