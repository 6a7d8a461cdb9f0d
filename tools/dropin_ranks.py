#!/usr/bin/env python3
"""Host-side scaling of the sharded from-disk sweeps: the two split-sweeping drop-in entry points
(`calculate_frames_relations.run_split`, `make_visibility_info.run_split`) over ScanNet-sized on-disk scenes with 1, 2 and 4
ranks SHARING the one GPU of the box over gloo.

Why this says something about an 8-GPU node: a 320-frame scene is ~0.2 ms of kernels and ~48 ms of host work (PNG inflate on
`num_workers` threads, staging, the H2D copy), so what an N-GPU job scales with is the host side -- decode threads per rank,
the per-window exchange towards rank 0, rank 0's writer -- and all of that is the real thing here; only the GPU and its PCIe
link are shared (4 ranks x 20 scenes/s x 197 MB = 16 GB/s of a 57 GB/s link).

    python tools/dropin_ranks.py [--ranks 1,2,4] [--scenes 32] [--frames 320] [--workers N] [--passes 3] [--per-rank 2]

`--workers` (decode threads per scene in flight, per rank) is the SAME for every world size and defaults to what lets the largest
world fit the container's CPU quota (16 CPUs on the MI355X boxes -> 1): the question is whether the job scales when every rank
brings its own resources, as on an 8-GPU node; one rank with all the box's CPUs is already decode-bound at the quota.

Driver mode (no RANK in the environment) writes the inputs once (rendered frames hard-linked under 320 image ids: every
file is still opened and inflated on its own), starts each world size as `python -m torch.distributed.run` of this file, and
prints one JSON object: per world size and entry point the scenes/s of the best of the later passes, every rank's stage
timings (`wait_at_exchange`, `produce`, `decode`, rank 0's `consume` / `writer_drain`), and whether the output files'
SHA-256 digests equal the one-rank run's.
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]

ENTRY_POINTS = ("calculate_frames_relations.run_split", "make_visibility_info.run_split")


def write_inputs(root, n_scenes, n_frames, n_points, base_scenes=4, smooth=False):
    """`n_scenes` scene ids over `base_scenes` rendered scenes of 8 frames each; a scene's `n_frames` image ids cycle through
    its 8 frames (hard links).  ``smooth``: the frames without their sensor noise (a 9 x 9 box filter, as
    tools/device_ingest_bench.py --smooth): the PNG writer then picks the Paeth filter for practically every row, which is what
    real sensor depth of smooth surfaces gets.  Returns write_scannet_layout's paths."""
    import numpy as np
    from mspa import synth
    H, W = 480, 640
    bases = [synth.make_scene(5000 + k, n_points=n_points, n_frames=8, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0,
                              with_color=False) for k in range(min(base_scenes, n_scenes))]
    if smooth:
        from scipy.ndimage import uniform_filter
        for b in bases:
            for i in list(b.depth):
                b.depth[i] = uniform_filter(b.depth[i].astype(np.float64), 9).astype(np.uint16)
    scenes = []
    for s in range(n_scenes):
        b = bases[s % len(bases)]
        ids = b.image_ids
        E = {f"{5 * f:05d}": b.E[ids[f % 8]] for f in range(n_frames)}
        depth = {f"{5 * f:05d}": b.depth[ids[f % 8]] for f in range(n_frames)}
        scenes.append(synth.SynthScene(f"scene{s:04d}_00", b.K, b.A, E, b.points, depth, {}, b.color_hw, b.depth_hw, b.boxes))
    return synth.write_scannet_layout(scenes, root, compress_level=6, link_identical=True)


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def worker(a):
    """One rank (or the plain one-process run): `passes` passes of both entry points; rank-local JSON into --out."""
    import torch
    from mspa import shard, sweep
    import spatial_engine.camera_movement.calculate_frames_relations as CFR
    import spatial_engine.utils.scannet_utils.make_visibility_info as MVI
    from spatial_engine.utils.scannet_utils.handler import info_handler as IH
    paths = json.load(open(os.path.join(a.root, "paths.json")))
    orig = IH.SceneInfoHandler.__init__

    def init(self, info_path, *x, **k):
        orig(self, info_path, posed_images_root=paths["posed_images_root"], instance_data_root=paths["instance_data_root"])
    IH.SceneInfoHandler.__init__ = init
    ctx = shard.context_from_env()
    rank, world = (ctx.rank, ctx.world) if ctx is not None else (0, 1)
    if a.per_rank:
        os.environ["MSPA_WINDOW_PER_RANK"] = str(a.per_rank)
    out = {"rank": rank, "world": world, "legs": {}}
    for name, fn, fname in ((ENTRY_POINTS[0], CFR.run_split, "pairs.parquet"), (ENTRY_POINTS[1], MVI.run_split, "vis.parquet")):
        runs = []
        for rep in range(a.passes):
            d = os.path.join(a.out, f"w{world}", f"pass{rep}")
            tm = sweep.Timings()
            if ctx is not None:
                ctx.barrier()
            with contextlib.redirect_stdout(io.StringIO()):
                t0 = time.perf_counter()
                fn(paths["info_path"], os.path.join(d, fname), os.path.join(d, fname + ".warn.txt"), num_workers=a.workers,
                   keep=False, ctx=ctx, timings=tm)                       # ends with the ranks' barrier
                dt = time.perf_counter() - t0
            runs.append({"seconds": round(dt, 4), "busy_s": tm.as_dict()})
        leg = {"passes": runs}
        if rank == 0:
            d = os.path.join(a.out, f"w{world}", f"pass{a.passes - 1}")
            leg["digests"] = {f: _sha(os.path.join(d, f)) for f in sorted(os.listdir(d)) if not f.endswith(".warn.txt")}
        out["legs"][name] = leg
    with open(os.path.join(a.out, f"w{world}_rank{rank}.json"), "w") as f:
        json.dump(out, f)
    if ctx is not None:
        ctx.barrier()


def drive(ranks=(1, 2, 4), n_scenes=32, n_frames=320, n_points=131072, workers=None, passes=3, per_rank=2, timeout_s=900,
          keep_root=None, decode="host", smooth=False):
    """Returns the leg's dict (see the module docstring).  ``workers`` = decode threads per scene in flight PER RANK, the same
    for every world size (an N-GPU node gives every rank its own cores: what is measured is whether the job scales when the
    per-rank resources are fixed).  Default: the CPUs this container may use (cgroup quota, mspa/hostinfo.py) divided by the
    largest world, by the 2 scenes a rank keeps in flight and by 2 again (the other half of a rank's CPUs is for its main thread:
    staging, JSON formatting, parquet encoding) -- 16 CPUs / 4 ranks / 2 / 2 = 1 on the MI355X boxes -- so that the largest world
    still fits inside the quota instead of being throttled by it (with 2, the index sweep at 4 ranks spent 9.9 CPU-seconds frozen:
    profiles/r06_dropin_ranks.md).  ``decode`` = "host" by default HERE: the ranks of this leg share one GPU, and with the depth
    decode on the device (the sweeps' default at these sizes) that one GPU is what all of them wait for -- 4 ranks x 2 560 frames in
    flight on a chip that holds 3 584 decode waves -- which says nothing about a node where every rank has its own."""
    from mspa import hostinfo
    eff = hostinfo.effective_cpus(per_rank=False)
    if workers is None:
        workers = max(1, min(25, eff // (4 * max(ranks))))
    root = keep_root or tempfile.mkdtemp(prefix="mspa_dropin_ranks_")
    try:
        t0 = time.perf_counter()
        if not os.path.exists(os.path.join(root, "paths.json")):      # (a kept root: the inputs of an earlier call serve again)
            paths = write_inputs(root, n_scenes, n_frames, n_points, smooth=smooth)
            json.dump(paths, open(os.path.join(root, "paths.json"), "w"))
        t_inputs = time.perf_counter() - t0
        out_dir = os.path.join(root, "out_" + (decode or "default"))
        os.makedirs(out_dir, exist_ok=True)
        res = {"scenes": n_scenes, "frames_per_scene": n_frames, "vertices": n_points, "num_workers_per_rank": workers,
               "frames": "smooth (Paeth rows)" if smooth else "noisy (Sub / Up rows)", "depth_decode": decode or os.environ.get("MSPA_DEPTH_DECODE", "device"), "cpus_per_rank": max(1, eff // max(ranks)), "scenes_in_flight_per_rank": 2, "window_scenes_per_rank": per_rank, "passes": passes, "host_cpus": hostinfo.describe(),
               "inputs_written_in_s": round(t_inputs, 1),
               "what": "ranks share ONE GPU and its PCIe link (gloo); decode threads, exchange and rank 0's writer are the real "
                       "ones -- the host-side scaling an N-GPU node sees",
               "statistic": "best of the passes after the first (page cache, pools and pinned slots warm)", "worlds": {}}
        deadline = time.perf_counter() + timeout_s
        for world in ranks:
            # every rank of every world size gets the same resources: `eff // max(ranks)` CPUs for its native pool, 2 scenes in flight
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MSPA_DIST_BACKEND="gloo", OMP_NUM_THREADS="1",
                       MSPA_HOST_CPUS=str(max(1, eff // max(ranks))), MSPA_LOOKAHEAD="2")
            if decode:
                env["MSPA_DEPTH_DECODE"] = decode
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MSPA_BENCH_FORCE_DIST"):
                env.pop(k, None)
            args = [os.path.abspath(__file__), "--worker", "--root", root, "--out", out_dir, "--workers", str(workers),
                    "--passes", str(passes), "--per-rank", str(per_rank)]
            if world == 1:
                cmd = [sys.executable] + args
            else:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                       "127.0.0.1", "--master-port", str(port)] + args
            left = deadline - time.perf_counter()
            if left < 30:
                res["worlds"][str(world)] = {"skipped": "the leg's time budget was used up by the smaller worlds"}
                continue
            # own session: on a timeout the launcher AND its ranks are ended by process group, never by name
            th0 = hostinfo.throttle_stats()
            proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT,
                                    start_new_session=True)
            try:
                so, se = proc.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                import signal
                os.killpg(proc.pid, signal.SIGKILL)
                proc.communicate()
                res["worlds"][str(world)] = {"failed": f"timed out after {left:.0f} s"}
                continue
            if proc.returncode != 0:
                res["worlds"][str(world)] = {"failed": (se or so)[-1500:]}
                continue
            per_rank_json = [json.load(open(os.path.join(out_dir, f"w{world}_rank{k}.json"))) for k in range(world)]
            w = {}
            for name in ENTRY_POINTS:
                legs = [pr["legs"][name] for pr in per_rank_json]
                later = range(1, passes) if passes > 1 else range(1)
                best = min(later, key=lambda p: legs[0]["passes"][p]["seconds"])
                secs = max(l["passes"][best]["seconds"] for l in legs)
                busy = [l["passes"][best]["busy_s"] for l in legs]
                w[name] = {"seconds": secs, "scenes_per_s": round(n_scenes / secs, 2),
                           "frames_per_s": round(n_scenes * n_frames / secs, 1),
                           "passes_s": [max(l["passes"][p]["seconds"] for l in legs) for p in range(passes)],
                           "rank0_consume_busy_s": busy[0].get("consume"), "rank0_writer_drain_s": busy[0].get("writer_drain"),
                           "rank0_writer_backpressure_s": busy[0].get("writer_backpressure"),
                           "wait_at_exchange_s": [b.get("wait_at_exchange", 0.0) for b in busy],
                           "produce_s": [b.get("produce") for b in busy], "encode_s": [b.get("encode_deferred", b.get("encode")) for b in busy],
                           "encode_wait_s": [b.get("encode_wait") for b in busy], "stage_s": [b.get("stage") for b in busy], "decode_busy_s": [b.get("decode") for b in busy],
                           "exchange_s": [b.get("exchange") for b in busy], "digests": legs[0]["digests"]}
            th1 = hostinfo.throttle_stats()
            w["cfs_throttling_whole_run"] = {k: th1[k] - th0.get(k, 0) for k in th1}
            res["worlds"][str(world)] = w
        base = res["worlds"].get(str(ranks[0]), {})
        for world in ranks:
            w = res["worlds"].get(str(world), {})
            for name in ENTRY_POINTS:
                if name in w and name in base:
                    w[name]["speedup_vs_%d" % ranks[0]] = round(w[name]["scenes_per_s"] / base[name]["scenes_per_s"], 3)
                    w[name]["files_identical_to_%d_rank" % ranks[0]] = w[name]["digests"] == base[name]["digests"]
        for w in res["worlds"].values():
            for name in ENTRY_POINTS:
                if name in w:
                    w[name].pop("digests", None)
        return res
    finally:
        if keep_root is None:
            shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", action="store_true")
    ap.add_argument("--root")
    ap.add_argument("--out")
    ap.add_argument("--ranks", default="1,2,4")
    ap.add_argument("--scenes", type=int, default=32)
    ap.add_argument("--frames", type=int, default=320)
    ap.add_argument("--points", type=int, default=131072)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--per-rank", type=int, default=2)
    ap.add_argument("--timeout", type=int, default=900)
    ap.add_argument("--decode", default="host", help="host (default: the ranks share one GPU) | device")
    ap.add_argument("--smooth", action="store_true", help="frames without sensor noise (Paeth rows, what real sensor depth gets)")
    a = ap.parse_args()
    if a.worker:
        worker(a)
        return
    res = drive(tuple(int(x) for x in a.ranks.split(",")), a.scenes, a.frames, a.points, a.workers, a.passes, a.per_rank, a.timeout, decode=a.decode, smooth=a.smooth)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
