#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec of the MultiSPA geometry pipe on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (kernel K3, mspa_pair_reproject: back-projection ->
reprojection -> depth-buffer visibility) over one batch of synthetic 640x480 RGB-D frame pairs that
are already resident in HBM.  Workload = BASELINE.json configs[1]: visual_correspondence on 1k
640x480 pairs, one MI355X.  With --gpus N (launched by torch.distributed.run, one rank per GPU)
every rank processes its own batch of the same size (weak scaling, pairs shard embarrassingly) and
the per-pair records of the whole job are collated with one RCCL all_gather inside the timed region.

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
  roofline      algorithmic HBM bytes of the K3 launch / its HIP-event-measured duration vs 8 TB/s
  cpu_baseline  the NumPy restatement of the reference path (oracle/np_oracle.py) timed on this box

    python bench.py                      # N=1, 1000 pairs/step, corr variant
    MSPA_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2
                                         # testing aid for 1-GPU boxes: both ranks on GPU 0, collation over gloo
    python bench.py --variant dense      # + rgb in, byte mask, xyz f32, rgba out
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(ROOT, "multi-spatialmllm_amd"), ROOT]

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
H, W = 480, 640
P = H * W

# bytes per pixel that one pair MUST move (DESIGN.md "Algorithmic bytes"; SURVEY.md 8d formula):
#   depth1 u16 + depth2 u16 (+ rgb u8x3) in; visibility + pixel index (+ xyz f32x3 + rgba) out
VARIANTS = {
    "corr": {"outputs": ("vis_bits", "pix_i16", "counts"), "rgb": False,
             "bytes_per_px": 2 + 2 + 1 / 8 + 4},
    "dense": {"outputs": ("vis_u8", "pix_i16", "xyz_f32", "rgba", "counts"), "rgb": True,
              "bytes_per_px": 2 + 2 + 3 + 1 + 4 + 12},    # SURVEY.md 8d canonical dense = 7,372,800 B/pair; the rgba output
                                                             # (4 B/px) this variant also writes is NOT counted
    "minimal": {"outputs": ("vis_bits", "counts"), "rgb": False, "bytes_per_px": 2 + 2 + 1 / 8},
}


def _legs(spec):
    return [v for v in spec.replace("'", "").replace('"', "").split(",") if v and v != "none"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1000, help="frame pairs per step per GPU")
    ap.add_argument("--frames", type=int, default=2048, help="distinct device-resident frames per GPU")
    ap.add_argument("--base-frames", type=int, default=24, help="frames rendered on the host per GPU")
    ap.add_argument("--variant", choices=sorted(VARIANTS), default="corr")
    ap.add_argument("--mode", choices=("fast", "exact"), default="fast",
                    help="fast: MSPA_PAIR_FAST (bit-exact integers via guarded composed matrices); exact: the "
                         "reference's own operation order")
    ap.add_argument("--stream", choices=("auto", "on", "off"), default="auto",
                    help="MSPA_PAIR_STREAM hint (frame 1 read non-temporally); auto = on when the step's pairs touch each resident "
                         "frame at most ~1.5 times, as they do with the default 2040 frames / 1000 pairs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--pair-offsets", default="1,2,3", help="frame-index distances the pairs are drawn from "
                    "(neighbouring views of the camera walk; larger = less of frame 1 lands in frame 2)")
    ap.add_argument("--walk-step", type=float, default=0.15, help="camera random-walk step (m) of the synthetic scene")
    ap.add_argument("--target-jitter", type=float, default=0.8, help="look-at jitter (m) of the synthetic scene")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scene-legs", action="store_true", help="skip the K1/K2 informational legs")
    ap.add_argument("--also", default="corr:exact,dense:fast,dense:exact,minimal:fast",
                    help="comma list of extra variant:mode legs timed briefly on rank 0 ('none' = skip)")
    return ap.parse_args()


def make_base_scene(args, rank):
    """Host-only part of the inputs: the seeded synthetic scene and the step's pair list."""
    from mspa import synth
    sc = synth.make_scene(1000 + rank, n_points=64, n_frames=args.base_frames, color_hw=(H, W),
                          depth_hw=(H, W), invalid_pose_frac=0.0, with_color=False, walk_step=args.walk_step,
                          target_jitter=args.target_jitter)
    nb = len(sc.valid_image_ids)
    reps = max(1, args.frames // nb)
    rng = np.random.default_rng(77 + rank)
    rep = np.arange(args.pairs) % reps
    b1 = rng.integers(0, nb, args.pairs)
    offs = np.array([int(v) for v in args.pair_offsets.split(",")])
    b2 = (b1 + offs[rng.integers(0, len(offs), args.pairs)]) % nb
    # offset 0 = the same pose seen in the next replica (independent depth noise): every pixel lands in view
    rep2 = np.where(b2 == b1, (rep + 1) % reps, rep)
    pairs_np = np.stack([rep * nb + b1, rep2 * nb + b2], axis=1).astype(np.int32)
    return sc, pairs_np


def build_inputs(args, rank, device, sc, pairs_np):
    """Expand the host-rendered frames on the device into `frames` distinct frames."""
    import torch
    from mspa import engine

    ids = sc.valid_image_ids
    nb = len(ids)
    base_depth = np.stack([sc.depth[i] for i in ids])
    base_mats = engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids])
    reps = max(1, args.frames // nb)
    n_frames = reps * nb
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    d0 = torch.from_numpy(base_depth.astype(np.int32)).to(device)                     # [nb,H,W]
    depth = torch.empty((n_frames, H, W), dtype=torch.int16, device=device)
    for r in range(reps):   # every replica gets its own +-3 mm noise so no two frames share bytes
        noise = torch.randint(-3, 4, d0.shape, generator=g, device=device, dtype=torch.int32)
        d = torch.where(d0 > 0, (d0 + noise).clamp_(1, 65535), d0)
        depth[r * nb:(r + 1) * nb] = d.to(torch.int16)      # same 16 bits; the kernels read them as uint16
    mats = torch.from_numpy(np.tile(base_mats, (reps, 1, 1))).to(device)
    rgb = None
    if VARIANTS[args.variant]["rgb"] or any(VARIANTS[v.split(":")[0]]["rgb"] for v in _legs(args.also)):
        rgb = torch.randint(0, 256, (n_frames, H, W, 3), generator=g, device=device, dtype=torch.uint8)
    # pairs: two different views of the same replica, walking through all replicas (make_base_scene)
    pairs = torch.from_numpy(pairs_np).to(device)
    return ids, depth, mats, rgb, pairs, nb


def stream_hint(args, n_frames):
    if args.stream != "auto":
        return args.stream == "on"
    return 2 * args.pairs <= 1.5 * n_frames


def time_variant(variant, mode, depth, mats, rgb, pairs, steps, warmup, dist_ctx, stream=False):
    """Returns (wall seconds for `steps` steps, mean kernel ms from HIP events, outputs)."""
    import torch
    from mspa import engine, shard

    from mspa import _lib
    spec = VARIANTS[variant]
    flags = (_lib.PAIR_FAST | (_lib.PAIR_STREAM if stream else 0)) if mode == "fast" else 0
    # With N > 1 every step's per-pair records land in one job-level table [steps, pairs, 2] (the kernel writes its
    # slice directly) which is collated ONCE, inside the timed region, with a single RCCL all_gather -- the
    # pipeline's exchange step is per job (mspa/pipeline.py), not per launch.  (Collating after every launch was
    # measured at +13 % per step: the RCCL kernel competes with K3 for CUs and HBM.)
    n = pairs.shape[0]
    out = engine.alloc_pair_outputs(n, (H, W), spec["outputs"], depth.device)
    job_counts = torch.zeros((max(steps, warmup, 1), n, 2), dtype=torch.int32, device=depth.device) \
        if dist_ctx is not None else None
    rgb_in = rgb if spec["rgb"] else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def step(k, timed):
        if job_counts is not None:
            out["counts"] = job_counts[k]
        if timed:
            ev[k][0].record()
        engine.pair_reproject(depth, mats, pairs, (H, W), out, rgb=rgb_in, flags=flags)
        if timed:
            ev[k][1].record()

    def collate(n_steps):
        if dist_ctx is None:
            return None
        table, work = shard.collate_records_async(job_counts[:n_steps].reshape(n_steps * n, 2), dist_ctx)
        work.wait()                              # orders the stream after the collective; the host does not block
        return table

    for k in range(warmup):
        step(k, False)
    collate(max(warmup, 1))                      # also warms the communicator up
    if dist_ctx is not None:
        dist_ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k, True)
    table = collate(steps)
    torch.cuda.synchronize()
    if dist_ctx is not None:
        dist_ctx.barrier()
    wall = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if table is not None and dist_ctx.rank == 0:         # the collated table really holds every rank's records
        assert table.shape[0] == dist_ctx.world * steps * n and int(table[:, 0].min()) > 0
    return wall, kern_ms, out


def measured_hbm_ceilings(device):
    """What this GPU's memory system delivers to plain torch kernels (GB/s): write-only fill and read+write copy of 2 GiB.
    The copy figure is the practical ceiling for a kernel that, like K3, both reads and writes HBM."""
    import torch
    n = (1 << 31) // 4
    x = torch.empty(n, dtype=torch.int32, device=device)
    y = torch.empty_like(x)
    out = {}
    for name, fn, nbytes in (("fill_write_only", lambda: x.fill_(3), 4 * n), ("copy_read_write", lambda: y.copy_(x), 8 * n)):
        for _ in range(2):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name] = round(nbytes * 10 / (a.elapsed_time(b) * 1e-3) / 1e9, 1)
    del x, y
    return out


def time_scene_kernels(device, n_points=131072, n_frames=320, reps=5):
    """K1 (vertex visibility) + K2 (all-pairs overlap) + K4 (pair pose) on one synthetic scene:
    the per-scene work of CFR.process_scene.  Informational legs with their own byte formulas
    (DESIGN.md section 4): K1 24*N + 2*DW*DH + N/8 per image, K2 2*N/8 + 8 per pair."""
    import torch
    from mspa import engine, synth

    sc = synth.make_scene(4000, n_points=n_points, n_frames=8, color_hw=(H, W), depth_hw=(H, W),
                          invalid_pose_frac=0.0, with_color=False)
    ids = sc.valid_image_ids
    reps_f = n_frames // len(ids)
    Ea = [sc.A @ sc.E[i] for i in ids] * reps_f
    cam = torch.from_numpy(engine.camera_matrices(sc.K, Ea)).to(device)
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids] * reps_f), device)
    xyz = torch.from_numpy(np.ascontiguousarray(sc.points[:, :3])).to(device)
    F = len(Ea)
    pairs = engine.all_pairs(F, device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    yaw, pitch = engine.extract_yaw_pitch_host(Ea)
    E_t = torch.from_numpy(np.stack(Ea).reshape(F, 16)).to(device)
    yaw_t, pitch_t = torch.from_numpy(yaw).to(device), torch.from_numpy(pitch).to(device)
    t1 = t2 = t3 = 0.0
    for r in range(reps + 1):
        ev[0].record()
        vis = engine.vertex_visibility(xyz, cam, depth, (H, W), ("bits", "count"))
        ev[1].record()
        ov = engine.pair_overlap(vis["bits"], pairs)
        ev[2].record()
        pose = engine.pair_pose(E_t, cam[:, 0, :].contiguous(), yaw_t, pitch_t, pairs)
        ev[3].record()
        torch.cuda.synchronize()
        if r:
            t1 += ev[0].elapsed_time(ev[1])
            t2 += ev[1].elapsed_time(ev[2])
            t3 += ev[2].elapsed_time(ev[3])
    t1, t2, t3 = t1 / reps, t2 / reps, t3 / reps
    # K8 (object extents for the coverage search) on the same bitsets: the scene's furniture as objects, and
    # K7 (rigid-body distance-change accumulation) on a TAPVid-sized track block
    idx, _, _ = sc.objects()
    offsets = np.zeros(len(idx) + 1, dtype=np.int32)
    offsets[1:] = np.cumsum([len(v) for v in idx.values()])
    off_t = torch.from_numpy(offsets).to(device)
    verts_t = torch.from_numpy(np.concatenate(list(idx.values())).astype(np.int32)).to(device)
    tracks = torch.randn((300, 256, 3), dtype=torch.float64, device=device)
    t8 = t7 = 0.0
    for r in range(reps + 1):
        ev[0].record()
        engine.object_extents(vis["bits"], xyz, off_t, verts_t)
        ev[1].record()
        engine.track_rigidity_loss(tracks)
        ev[2].record()
        torch.cuda.synchronize()
        if r:
            t8 += ev[0].elapsed_time(ev[1])
            t7 += ev[1].elapsed_time(ev[2])
    t8, t7 = t8 / reps, t7 / reps
    b1 = F * (24 * n_points + 2 * H * W + n_points // 8)
    b2 = pairs.shape[0] * (2 * n_points // 8 + 8)
    return {"K1_vertex_visibility": {"images": F, "vertices": n_points, "kernel_ms": round(t1, 4),
                                     "images_per_s": round(F / (t1 * 1e-3), 1),
                                     "achieved_GBs": round(b1 / (t1 * 1e-3) / 1e9, 1),
                                     "frac": round(b1 / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            "K2_pair_overlap": {"pairs": int(pairs.shape[0]), "kernel_ms": round(t2, 4),
                                "pairs_per_s": round(pairs.shape[0] / (t2 * 1e-3), 1),
                                "streaming_GBs": round(b2 / (t2 * 1e-3) / 1e9, 1)},
            "K4_pair_pose": {"pairs": int(pairs.shape[0]), "kernel_ms": round(t3, 4)},
            "K8_object_extents": {"objects": len(idx), "object_vertices": int(offsets[-1]), "images": F,
                                  "kernel_ms": round(t8, 4),
                                  "bit_tests_per_s": round(float(offsets[-1]) * F / (t8 * 1e-3), 1)},
            "K7_track_rigidity": {"frames": 300, "points": 256, "kernel_ms": round(t7, 4)},
            "scene_total": {"frames": F, "vertices": n_points, "ms": round(t1 + t2 + t3, 4),
                            "note": "CFR.process_scene for one ScanNet-sized scene (every-5th-frame average)"}}


_CPU_SCENE = None


def _cpu_pair(job):
    """One frame pair through the NumPy restatement of the reference path (worker side)."""
    from oracle import np_oracle as O
    sc, ids, color = _CPU_SCENE
    a, b = job
    r = O.frame_pair(sc.depth[ids[a]], sc.depth[ids[b]], sc.K, sc.E[ids[a]], sc.E[ids[b]], sc.A, (H, W), color)
    return r["n_vis"]


def _cpu_worker_init():
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)                 # one BLAS thread per worker, as OMP_NUM_THREADS=1 would
    except Exception:
        pass


def cpu_baseline(sc, pairs_np, budget_s):
    """Time the NumPy restatement of the reference path (oracle/np_oracle.frame_pair) on a bounded sample
    of the step's pairs: (i) one process, (ii) multiprocessing.Pool(min(25, cores)) -- the reference's own
    worker count (CFR:280).  Runs BEFORE the GPU is initialised so that the pool can fork safely."""
    global _CPU_SCENE
    import multiprocessing as mp
    ids = sc.valid_image_ids
    nb = len(ids)
    _CPU_SCENE = (sc, ids, np.zeros((H, W, 3), dtype=np.uint8))
    jobs = [(int(p[0] % nb), int(p[1] % nb)) for p in pairs_np]
    n, t0 = 0, time.perf_counter()
    while True:
        _cpu_pair(jobs[n])
        n += 1
        el = time.perf_counter() - t0
        if (el > budget_s / 2 and n >= 8) or n >= len(jobs):
            break
    single = n / el
    workers = min(25, os.cpu_count() or 1)
    pool_jobs = jobs[:max(workers * 4, int(single * workers * budget_s / 2))][:len(jobs)]
    pool_rate = None
    try:
        with mp.get_context("fork").Pool(workers, initializer=_cpu_worker_init) as pool:
            pool.map(_cpu_pair, pool_jobs[:workers])                       # warm the workers
            t1 = time.perf_counter()
            pool.map(_cpu_pair, pool_jobs, chunksize=max(1, len(pool_jobs) // (workers * 4)))
            pool_rate = len(pool_jobs) / (time.perf_counter() - t1)
    except Exception as e:                                                  # no fork / no /dev/shm: report single only
        print(f"[bench] pool baseline skipped: {e}", file=sys.stderr)
    return {"value": round(single, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n} of the step's 640x480 pairs through oracle/np_oracle.frame_pair "
                      f"(NumPy {np.__version__}, 1 process, in-memory images) in {el:.1f} s",
            "pool": None if pool_rate is None else {
                "value": round(pool_rate, 2), "cores": workers,
                "sample": f"{len(pool_jobs)} pairs over multiprocessing.Pool({workers}), 1 BLAS thread each"},
            "host_cores_available": os.cpu_count()}


def main():
    args = parse_args()
    import torch
    from mspa import _lib, engine, shard

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    sc, pairs_np = make_base_scene(args, rank)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sc, pairs_np, args.cpu_seconds)     # before any GPU initialisation (fork-safe)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    share = bool(os.environ.get("MSPA_BENCH_SHARE_GPU"))     # testing aid: every rank on GPU 0, collation over gloo
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    _lib.load()
    ids, depth, mats, rgb, pairs, nb = build_inputs(args, rank, device, sc, pairs_np)
    # MSPA_BENCH_FORCE_DIST=1 exercises the RCCL collation path with a single rank (1-GPU boxes)
    dist_ctx = (shard.init_distributed(device, backend="gloo" if share else None)
                if (world > 1 or os.environ.get("MSPA_BENCH_FORCE_DIST")) else None)
    stream = stream_hint(args, int(depth.shape[0]))
    wall, kern_ms, out = time_variant(args.variant, args.mode, depth, mats, rgb, pairs, args.steps, args.warmup,
                                      dist_ctx, stream)
    if dist_ctx is not None:
        wall = dist_ctx.max_over_ranks(wall)
    pairs_per_step = args.pairs * world
    value = pairs_per_step * args.steps / wall
    spec = VARIANTS[args.variant]
    bytes_per_launch = spec["bytes_per_px"] * P * args.pairs
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9

    extra = {}
    if rank == 0 and world == 1:      # informational single-GPU legs; with N > 1 every rank leaves together after the timed job
        for leg in [v for v in _legs(args.also) if v != f"{args.variant}:{args.mode}"]:
            v, m = leg.split(":")
            w2, k2, _ = time_variant(v, m, depth, mats, rgb, pairs, max(3, args.steps // 4), 1, None, stream)
            b2 = VARIANTS[v]["bytes_per_px"] * P * args.pairs
            extra[leg] = {"pairs_per_s_1gpu": round(args.pairs / (k2 * 1e-3), 1), "kernel_ms": round(k2, 4),
                        "achieved_GBs": round(b2 / (k2 * 1e-3) / 1e9, 1),
                        "frac": round(b2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "bytes_per_pair": int(VARIANTS[v]["bytes_per_px"] * P)}
        if world == 1 and not args.no_scene_legs:
            extra["scene"] = time_scene_kernels(device)

    if rank == 0:
        traffic = None
        counts = out["counts"].cpu().numpy()
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            t = json.load(open(tfile))
            if t.get("variant") == args.variant and t.get("pairs") == args.pairs and t.get("mode") == args.mode:
                traffic = t.get("hbm_bytes_per_launch")
        info = _lib.device_info(local_rank)
        ceilings = measured_hbm_ceilings(device) if (world == 1 and not args.no_scene_legs) else None
        line = {
            "metric": "frame-pairs/sec MultiSPA geometry pipe (640x480 RGB-D)",
            "value": round(value, 1), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "visual_correspondence unproject+reproject+occlusion kernel (K3) on "
                                   f"{args.pairs} 640x480 frame pairs per GPU per step (BASELINE.json configs[1])",
                       "variant": args.variant, "mode": args.mode, "outputs": list(spec["outputs"]),
                       "stream_hint": bool(stream and args.mode == "fast"),
                       "pairs_per_step_per_gpu": args.pairs, "pair_offsets": args.pair_offsets, "distinct_frames_per_gpu": int(depth.shape[0]),
                       "image": "640x480 depth u16 (+rgb u8x3 for dense)", "parallelism": f"dp{world}",
                       "collation": "one RCCL all_gather of the job's per-pair records inside the timed region" if world > 1 else "none (1 GPU)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": "mspa::pair_fast_tight_kernel" if args.mode == "fast" else "mspa::pair_exact_kernel",
                         "kernel_ms": round(kern_ms, 4), "bytes_per_pair": int(spec["bytes_per_px"] * P),
                         "measured_ceilings_GBs": ceilings,
                         "traffic_GBs": round(traffic / (kern_ms * 1e-3) / 1e9, 1) if traffic else None},
            "cpu_baseline": cpu,
            "variants": extra,
            "device": info,
            "visible_fraction": round(float(counts[:, 1].sum() / max(1, counts[:, 0].sum())), 4),
        }
        print(json.dumps(line), flush=True)
    if dist_ctx is not None:
        dist_ctx.barrier()          # leave together: no rank tears the communicator down under another one's feet
        dist_ctx.close()


if __name__ == "__main__":
    main()
