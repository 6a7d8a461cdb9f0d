#!/usr/bin/env python3
"""bench.py -- frame-pairs/sec of the MultiSPA geometry pipe on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (kernel K3, mspa_pair_reproject: back-projection -> reprojection ->
depth-buffer visibility) over one batch of synthetic 640x480 RGB-D frame pairs that are already resident in HBM.
Workload = BASELINE.json configs[1]: visual_correspondence on 1k 640x480 pairs, one MI355X.

WHICH pairs is part of the contract (the fast kernels cull what cannot land in frame 2, so their cost depends on
the overlap of the two views): the pairs are drawn from the scene's all-pairs overlap table (K1 + K2 = the
reference's calculate_camera_overlap) with the reference's own overlap-binned sampler -- equal quotas over the
1 %-wide bins 6..35 % (VC_C:485-505), both frame orders (VC_C:280) -- see mspa/workload.py.  `value` is that
workload ("vc"); the same line carries a fixed three-point sweep (low / vc / high overlap) in `sweep`.

    python bench.py                      # N=1, 1000 pairs/step, corr variant, vc workload
    python bench.py --gpus 8             # re-executes itself under torch.distributed.run, one rank per GPU
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8     # what the driver does: same thing
    python bench.py --variant dense      # + rgb in, byte mask, xyz f32, rgba out

With --gpus N every rank processes its own batch of the same size (weak scaling, pairs shard embarrassingly) and
the per-pair records of the whole job are collated with one RCCL all_gather inside the timed region.  On a box
with fewer GPUs than ranks the ranks share the GPUs that exist and collate over gloo (reported in the line:
"gpus_shared": true -- a launch check, not a scaling measurement).

Prints ONE JSON line on rank 0 (see the contract in the task statement) with two extra objects:
  roofline      algorithmic HBM bytes of the K3 launch / its HIP-event-measured duration vs 8 TB/s
  cpu_baseline  the NumPy restatement of the reference path (oracle/np_oracle.py) timed on this box
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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # before the first HIP call: the from-disk legs overlap ten decode streams (mspa/__init__.py)

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
    "minimal": {"outputs": ("vis_bits", "counts"), "rgb": False, "bytes_per_px": 2 + 2 + 1 / 8,
                "note": "this build's smallest set (rgb = 0, bit mask only) -- NOT SURVEY 8d's 'minimal', which also reads rgb "
                        "(2,188,800 B/pair)"},
    # SURVEY.md 8d's formula with rgb = 0: byte mask, pixel index, float32 point -- the dense set without the colour words
    "dense_xyz": {"outputs": ("vis_u8", "pix_i16", "xyz_f32", "counts"), "rgb": False, "bytes_per_px": 2 + 2 + 1 + 4 + 12},
    # compacted correspondences (mspa_pair_correspondences): bit mask + 4 B per VISIBLE pixel + a count per tile;
    # `bytes_per_px` is the fixed part, the visible part is added from the launch's own counters (variant_bytes)
    "compact": {"outputs": ("vis_bits", "cpix", "tile_counts", "counts"), "rgb": False, "bytes_per_px": 2 + 2 + 1 / 8,
                "compact": True},
}


def variant_bytes(variant, n_pairs, out=None):
    """Algorithmic HBM bytes of one launch of `variant` over n_pairs pairs.  The compacted set's output size depends on the
    data: 4 bytes per visible pixel (read off the launch's own counters) + 4 bytes per tile count."""
    spec = VARIANTS[variant]
    b = spec["bytes_per_px"] * P * n_pairs
    if spec.get("compact") and out is not None:
        b += 4.0 * float(out["counts"][:, 1].sum().item()) + 4.0 * out["tile_counts"].numel()
    return b


def _legs(spec):
    return [v for v in spec.replace("'", "").replace('"', "").split(",") if v and v != "none"]


WORKLOADS = ("vc", "low", "high")
SCENE_WORKLOAD = "scenes"        # BASELINE.json configs[2] in the bench's shape: whole scenes (K1 + K2 + K4), sharded longest-first


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1000, help="frame pairs per step per GPU")
    ap.add_argument("--frames", type=int, default=2048, help="distinct device-resident frames per GPU")
    ap.add_argument("--base-frames", type=int, default=64, help="frames of the synthetic scene rendered on the host per "
                    "GPU (SURVEY.md 8d: F = 64 cameras per scene)")
    ap.add_argument("--scene-points", type=int, default=131072, help="scene vertices the overlap table is measured on")
    ap.add_argument("--workload", choices=WORKLOADS + (SCENE_WORKLOAD,), default="vc",
                    help="which pairs of the scene (mspa/workload.py): vc = the reference's overlap-binned sample 6..35 %% "
                         "(headline), low = overlap < 6 %%, high = near-identical views; 'scenes' = the scene-shaped job of "
                         "configs[2] (K1 + K2 + K4 over 8 ScanNet-sized scenes per GPU, pair-table rows collated over RCCL)")
    ap.add_argument("--scenes-per-gpu", type=int, default=8, help="--workload scenes: resident scenes per GPU")
    ap.add_argument("--no-dropin-sweep", action="store_true", help="skip the from-disk run_split leg")
    ap.add_argument("--no-dropin-ranks", action="store_true",
                    help="skip the 1 / 2 / 4-ranks-on-one-GPU from-disk leg (tools/dropin_ranks.py)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not re-run the headline under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two short child runs) to "
                         "measure `roofline.traffic` on THIS box; the committed profiles/traffic.json value is used instead")
    ap.add_argument("--variant", choices=sorted(VARIANTS), default="corr")
    ap.add_argument("--mode", choices=("fast", "exact"), default="fast",
                    help="fast: MSPA_PAIR_FAST (bit-exact integers via guarded composed matrices); exact: the "
                         "reference's own operation order")
    ap.add_argument("--stream", choices=("auto", "on", "off"), default="auto",
                    help="MSPA_PAIR_STREAM hint (frame 1 read non-temporally); auto = on when the step's pairs touch each resident "
                         "frame at most ~1.5 times, as they do with the default 2048 frames / 1000 pairs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--walk-step", type=float, default=0.08, help="camera random-walk step (m) of the synthetic scene")
    ap.add_argument("--target-step", type=float, default=0.25, help="look-at random-walk step (m) of the synthetic scene")
    ap.add_argument("--spinup-ms", type=float, default=100.0,
                    help="untimed clock spin-up before the W warm-up steps: the step itself, repeated for this many ms (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scene-legs", action="store_true", help="skip the K1/K2 informational legs")
    ap.add_argument("--no-sweep", action="store_true", help="skip the low / high overlap legs of the sweep")
    ap.add_argument("--also", default="compact:fast,corr:exact,dense:fast,dense_xyz:fast,dense:exact,minimal:fast",
                    help="comma list of extra variant:mode legs timed briefly on rank 0 ('none' = skip)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU, exactly
    the command line the driver uses for N > 1.  The rank-0 JSON line and the exit code pass through."""
    import socket
    import subprocess
    import torch
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print(f"[bench] {args.gpus} ranks on {n_dev} GPU(s): ranks share devices, collation over gloo "
              "(launch check only, not a scaling measurement)", file=sys.stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def _scene_cache_dir():
    """A directory only this user can write (0700, ownership checked): the cached scene is plain arrays (np.load with
    allow_pickle=False), and nobody else can plant a file under a name the bench will open."""
    import tempfile
    d = os.path.join(tempfile.gettempdir(), f"mspa_bench_{os.getuid()}")
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.stat(d)
        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
            return None
    except OSError:
        return None
    return d


def make_base_scene(args, rank):
    """Host-only part of the inputs: the seeded synthetic scene (SURVEY.md 8d recipe; the camera follows a hand-held
    sweep so that one scene populates every overlap bin the reference samples from).  Rendering 64 frames on the host
    takes ~20 s, so the finished scene's arrays are cached (profiling runs the same command six times)."""
    from mspa import synth
    d = _scene_cache_dir()
    key = f"scene_{1000 + rank}_{args.scene_points}_{args.base_frames}_{args.walk_step}_{args.target_step}.npz"
    path = os.path.join(d, key) if d else None
    if path and os.path.exists(path):
        try:
            z = np.load(path, allow_pickle=False)
            ids = [str(i) for i in z["ids"]]
            return synth.SynthScene(scene_id=str(z["scene_id"]), K=z["K"], A=z["A"], E={i: z["E"][k] for k, i in enumerate(ids)},
                                    points=z["points"], depth={i: z["depth"][k] for k, i in enumerate(ids)}, color={},
                                    color_hw=(H, W), depth_hw=(H, W), boxes=z["boxes"])
        except Exception:
            pass
    sc = synth.make_scene(1000 + rank, n_points=args.scene_points, n_frames=args.base_frames, color_hw=(H, W),
                          depth_hw=(H, W), invalid_pose_frac=0.0, with_color=False, trajectory="sweep",
                          walk_step=args.walk_step, target_step=args.target_step)
    if path:
        try:
            ids = sc.image_ids
            tmp = path + f".{os.getpid()}.npz"
            np.savez(tmp, scene_id=np.str_(sc.scene_id), ids=np.array(ids), K=sc.K, A=sc.A, E=np.stack([sc.E[i] for i in ids]),
                     points=sc.points, depth=np.stack([sc.depth[i] for i in ids]), boxes=sc.boxes)
            os.replace(tmp, path)
        except Exception:
            pass
    return sc


def build_inputs(args, rank, device, sc):
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
    return depth, mats, rgb, nb, reps


def scene_overlap_table(sc, device):
    """The scene's all-pairs overlap column exactly as the reference defines it (CFR:102-137: |a & b| / |a | b| * 100 over
    the per-image vertex visibility masks), from K1 + K2 -- bit-equal to the reference (tests/test_gpu_parity.py).
    Not timed: it only decides which pairs the step consists of."""
    import torch
    from mspa import engine
    ids = sc.valid_image_ids
    Ea = [sc.A @ sc.E[i] for i in ids]
    cam = torch.from_numpy(engine.camera_matrices(sc.K, Ea)).to(device)
    depth = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), device)
    xyz = torch.from_numpy(np.ascontiguousarray(sc.points[:, :3])).to(device)
    vis = engine.vertex_visibility(xyz, cam, depth, (H, W), ("bits",))
    return engine.scene_overlap(vis["bits"]).cpu().numpy()


def workload_pairs(overlap, nb, reps, n_pairs, kind, rank):
    """Frame-index pairs of one workload: base-scene pairs from mspa/workload.py, pair p placed in replica p % reps
    (1 000 pairs walk through 2 048 distinct frames).  Returns (pairs [n,2] int32, base pairs, description)."""
    from mspa import workload
    base, info = workload.select_pairs(overlap, nb, n_pairs, kind, seed=77 + rank)
    rep = (np.arange(n_pairs) % reps).astype(np.int32)
    return np.stack([rep * nb + base[:, 0], rep * nb + base[:, 1]], axis=1).astype(np.int32), base, info


def stream_hint(args, n_frames):
    if args.stream != "auto":
        return args.stream == "on"
    return 2 * args.pairs <= 1.5 * n_frames


SPINUP = {"steps": 0}


def time_variant(variant, mode, depth, mats, rgb, pairs, steps, warmup, dist_ctx, stream=False, spinup_ms=0.0,
                 per_step_events=False):
    """Returns (wall seconds for `steps` steps, mean kernel ms from HIP events, outputs).  ``per_step_events``: round 2's
    method -- an event pair around EVERY launch, mean of the per-step times (the `cold_ms_per_step` leg)."""
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
    compact = bool(spec.get("compact"))
    out = (engine.alloc_pair_correspondences(n, (H, W), depth.device) if compact
           else engine.alloc_pair_outputs(n, (H, W), spec["outputs"], depth.device))
    job_counts = torch.zeros((max(steps, warmup, 1), n, 2), dtype=torch.int32, device=depth.device) \
        if dist_ctx is not None else None
    rgb_in = rgb if spec["rgb"] else None
    # ONE pair of HIP events around the K timed launches (on the launch stream), not one pair per step: a timestamp marker
    # between two launches is a barrier packet of its own, and 2 K of them cost the step ~7 % (0.536 vs 0.501 ms measured on one
    # box against tools/ab_k3.py's back-to-back launches).  kernel_ms = elapsed / K therefore INCLUDES what else a launch
    # enqueues (the 8 KB counter memset) and the gaps between launches: an upper bound on the kernel's own duration, which the
    # committed rocprofv3 --kernel-trace averages (profiles/) sit just below.
    ev_start, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step(k):
        if job_counts is not None:
            out["counts"] = job_counts[k]
        if compact:
            engine.pair_correspondences(depth, mats, pairs, (H, W), out, flags=flags)
        else:
            engine.pair_reproject(depth, mats, pairs, (H, W), out, rgb=rgb_in, flags=flags)

    def collate(n_steps):
        if dist_ctx is None:
            return None
        table, work = shard.collate_records_async(job_counts[:n_steps].reshape(n_steps * n, 2), dist_ctx)
        work.wait()                              # orders the stream after the collective; the host does not block
        return table

    spun = 0
    if spinup_ms > 0:
        # Clock spin-up, untimed and reported (`config.clock_spinup`): coming from the host-side input preparation the part needs
        # ~30 ms under load to reach its sustained clock -- measured on one box: 0.553 ms per step with `--warmup 5 --steps 20`,
        # 0.506 with `--warmup 60`, 0.509 with `--steps 200`.  The same step is run until `spinup_ms` of device time have
        # passed; the W warm-up steps and the K timed steps follow unchanged.
        step(0)                                   # first launch: code-object load, allocator -- not device time
        torch.cuda.synchronize()
        spun = 1
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < spinup_ms:
            for _ in range(8):
                step(0)
            spun += 8
            torch.cuda.synchronize()
    for k in range(warmup):
        step(k)
    collate(max(warmup, 1))                      # also warms the communicator up
    if dist_ctx is not None:
        dist_ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if per_step_events:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for k in range(steps):
            evs[k][0].record()
            step(k)
            evs[k][1].record()
    else:
        ev_start.record()
        for k in range(steps):
            step(k)
        ev_end.record()
    table = collate(steps)
    torch.cuda.synchronize()
    if dist_ctx is not None:
        dist_ctx.barrier()
    wall = time.perf_counter() - t0
    kern_ms = (sum(float(a.elapsed_time(b)) for a, b in evs) / steps) if per_step_events else float(ev_start.elapsed_time(ev_end)) / steps
    if spinup_ms > 0:
        SPINUP["steps"] = spun
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
        ov = engine.scene_overlap(vis["bits"])           # tiled all-pairs K2
        ev[2].record()
        pose = engine.pair_pose(E_t, cam[:, 0, :].contiguous(), yaw_t, pitch_t, pairs)
        ev[3].record()
        torch.cuda.synchronize()
        if r:
            t1 += ev[0].elapsed_time(ev[1])
            t2 += ev[1].elapsed_time(ev[2])
            t3 += ev[2].elapsed_time(ev[3])
    t1, t2, t3 = t1 / reps, t2 / reps, t3 / reps
    # K1 once more on the same vertices in a spatially coherent order (Morton curve over a 1024^3 grid): mesh files list
    # vertices block by block of the reconstruction, the synthetic cloud above is shuffled -- the two ends of the gather's
    # cache-line locality.  Same bits up to the permutation.
    pts = np.ascontiguousarray(sc.points[:, :3])
    q = ((pts - pts.min(0)) / np.maximum(np.ptp(pts, axis=0), 1e-9) * 1023).astype(np.uint64)
    def part1by2(v):
        v = (v | (v << 32)) & np.uint64(0x1F00000000FFFF)
        v = (v | (v << 16)) & np.uint64(0x1F0000FF0000FF)
        v = (v | (v << 8)) & np.uint64(0x100F00F00F00F00F)
        v = (v | (v << 4)) & np.uint64(0x10C30C30C30C30C3)
        return (v | (v << 2)) & np.uint64(0x1249249249249249)
    order = np.argsort(part1by2(q[:, 0]) | (part1by2(q[:, 1]) << np.uint64(1)) | (part1by2(q[:, 2]) << np.uint64(2)), kind="stable")
    xyz_sorted = torch.from_numpy(np.ascontiguousarray(pts[order])).to(device)
    t1s = 0.0
    for r in range(reps + 1):
        ev[0].record()
        vis_s = engine.vertex_visibility(xyz_sorted, cam, depth, (H, W), ("bits", "count"))
        ev[1].record()
        torch.cuda.synchronize()
        if r:
            t1s += ev[0].elapsed_time(ev[1])
    t1s /= reps
    same_counts = bool(torch.equal(vis_s["count"], vis["count"]))
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
    # K1's compulsory HBM bytes for the scene: the vertex array once, every depth frame once, every bitset row once
    # (SURVEY.md 8d's streaming formula F * (24 N + ...) re-reads the vertices per image and credits no residency at all:
    # it gave "fractions" above 1 and is no longer printed)
    b1 = 24 * n_points + F * (2 * H * W + n_points // 8)
    b2 = pairs.shape[0] * (2 * n_points // 8 + 8)
    # The PRIMARY K1 reading is the spatially coherent vertex order (a Morton curve over the same vertices): mesh files list
    # vertices patch by patch, which is what ScanNet's aligned_points are; the shuffled synthetic cloud -- every wave's 64
    # vertices scattered over the whole room -- is kept beside it as the worst case.
    return {"K1_vertex_visibility": {"images": F, "vertices": n_points, "kernel_ms": round(t1s, 4),
                                     "images_per_s": round(F / (t1s * 1e-3), 1),
                                     "compulsory_bytes": int(b1),
                                     "compulsory_GBs": round(b1 / (t1s * 1e-3) / 1e9, 1),
                                     "compulsory_frac": round(b1 / (t1s * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                     "includes": "vertex_visibility_compact_kernel + bits_count_kernel",
                                     "vertex_order": "spatially coherent (Morton curve over the synthetic cloud: mesh-like)",
                                     "method_changed_in_round": 4,    # rounds 1-3 printed the shuffled cloud under these keys
                                                                      # (now `shuffled_worst_case`): not like for like across it
                                     "same_counts_as_shuffled": same_counts,
                                     "shuffled_worst_case": {"kernel_ms": round(t1, 4), "images_per_s": round(F / (t1 * 1e-3), 1),
                                                             "compulsory_frac": round(b1 / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                             "order": "the synthetic cloud as generated (shuffled)"}},
            "K2_pair_overlap": {"pairs": int(pairs.shape[0]), "kernel_ms": round(t2, 4), "form": "tiled (mspa_scene_overlap)",
                                "pairs_per_s": round(pairs.shape[0] / (t2 * 1e-3), 1),
                                "streaming_GBs": round(b2 / (t2 * 1e-3) / 1e9, 1)},
            "K4_pair_pose": {"pairs": int(pairs.shape[0]), "kernel_ms": round(t3, 4)},
            "K8_object_extents": {"objects": len(idx), "object_vertices": int(offsets[-1]), "images": F,
                                  "kernel_ms": round(t8, 4),
                                  "bit_tests_per_s": round(float(offsets[-1]) * F / (t8 * 1e-3), 1)},
            "K7_track_rigidity": {"frames": 300, "points": 256, "kernel_ms": round(t7, 4)},
            "scene_total": {"frames": F, "vertices": n_points, "ms": round(t1s + t2 + t3, 4),
                            "note": "CFR.process_scene for one ScanNet-sized scene (every-5th-frame average)"}}


def time_scene_pipeline(device, n_scenes=24, n_points=131072, n_frames=320):
    """End to end, host memory -> pair table: scenes of 320 frames x 131 072 vertices (the every-5th-frame ScanNet average)
    go through mspa.upload.ScenePrefetcher (pinned staging on a worker thread + H2D on a copy stream, overlapped with the
    previous scene's kernels), K1 + K2 + K4, and the pair-table columns come back to the host.  The scene's 8 rendered
    frames are referenced 40 times on the host (staging still copies 320 frames = 197 MB per scene)."""
    import torch
    from mspa import synth, upload

    sc = synth.make_scene(4000, n_points=n_points, n_frames=8, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0,
                          with_color=False)
    ids = sc.valid_image_ids
    reps = n_frames // len(ids)

    class HostScene:
        K, A, color_hw, points = sc.K, sc.A, sc.color_hw, sc.points
        E = {f"{r:03d}_{i}": sc.E[i] for r in range(reps) for i in ids}
        depth = {f"{r:03d}_{i}": sc.depth[i] for r in range(reps) for i in ids}

    def run(n):
        pairs = 0
        for scene in upload.ScenePrefetcher([HostScene] * n, device):
            pairs += len(scene.frames_relations_arrays()["overlap"])
        torch.cuda.synchronize()
        return pairs

    run(upload.UPLOAD_SLOTS + 1)                     # slots, pinned buffers, kernels warm
    runs = []
    for _ in range(3):                               # host-side leg (staging threads, PCIe): the spread between runs is reported
        t0 = time.perf_counter()
        pairs = run(n_scenes)
        runs.append(time.perf_counter() - t0)
    dt = sorted(runs)[1]                             # median of three
    nbytes = n_frames * H * W * 2 + n_points * 24
    return {"scenes": n_scenes, "frames_per_scene": n_frames, "vertices": n_points, "seconds": round(dt, 4),
            "scenes_per_s": round(n_scenes / dt, 2), "frames_per_s": round(n_scenes * n_frames / dt, 1),
            "pairs_per_s": round(pairs / dt, 1), "h2d_GBs": round(n_scenes * nbytes / dt / 1e9, 2),
            "scenes_per_s_runs": [round(n_scenes / t, 1) for t in runs], "statistic": "median of three runs",
            "includes": "pinned staging (worker thread) + H2D (copy stream, overlapped) + K1 + K2 + K4 + D2H of the "
                        "pair-table columns; bounded by host memcpy / PCIe, not by the kernels"}


def time_scannet_shape(device, n_pairs=200, base_frames=16, steps=20, warm=8):
    """K3 at ScanNet's OWN shape: a 1296 x 968 colour grid over 640 x 480 depth frames (extract_posed_images.py:93-97; the
    reference's project_mask_to_3d spans the colour grid, OPS:276-290) -- the tight kernel's rectangular-tile form, correspondence output
    set, pairs drawn by the reference's overlap-binned sampler from a 16-frame sweep of the SURVEY 8d room.
    Algorithmic bytes per pair: the two depth frames once (2 x 2 B x 307 200) + per colour pixel 1/8 B of bitset and 4 B of
    pixel index (4.125 B x 1 254 528) = 6 403 728 B."""
    import torch
    from mspa import _lib, engine, synth, workload
    CH, CW, DH, DW = 968, 1296, 480, 640
    sc = synth.make_scene(1000, n_points=32768, n_frames=base_frames, color_hw=(CH, CW), depth_hw=(DH, DW), invalid_pose_frac=0.0,
                          with_color=False, trajectory="sweep", walk_step=0.08, target_step=0.25)
    ids = sc.valid_image_ids
    nb = len(ids)
    cam = torch.from_numpy(engine.camera_matrices(sc.K, [sc.A @ sc.E[i] for i in ids])).to(device)
    d_base = engine.depth_to_device(np.stack([sc.depth[i] for i in ids]), device)
    xyz = torch.from_numpy(np.ascontiguousarray(sc.points[:, :3])).to(device)
    overlap = engine.scene_overlap(engine.vertex_visibility(xyz, cam, d_base, (CH, CW), ("bits",))["bits"]).cpu().numpy()
    base, info = workload.select_pairs(overlap, nb, n_pairs, "vc", seed=77)
    reps = 8                                                       # 128 resident frames: a pair's frames are rarely re-read
    depth = d_base.repeat(reps, 1, 1).contiguous()
    mats = torch.from_numpy(np.tile(engine.frame_matrices(sc.K, sc.A, [sc.E[i] for i in ids]), (reps, 1, 1))).to(device)
    rep = (np.arange(n_pairs) % reps).astype(np.int32)
    pairs = torch.from_numpy(np.stack([rep * nb + base[:, 0], rep * nb + base[:, 1]], 1).astype(np.int32)).to(device)
    out = engine.alloc_pair_outputs(n_pairs, (CH, CW), ("vis_bits", "pix_i16", "counts"), device)
    flags = _lib.PAIR_FAST | _lib.PAIR_STREAM
    def timed(fn):
        for _ in range(warm):
            fn()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(steps):
            fn()
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / steps

    corr = lambda: engine.pair_reproject(depth, mats, pairs, (CH, CW), out, flags=flags)
    timed(corr)                    # the leg follows seconds of host-side scene building: one untimed pass lets the clock come back up
    ms = timed(corr)
    kern = _lib.load().mspa_pair_reproject_last_kernel()
    bpp = 2 * 2 * DH * DW + (4 + 1 / 8) * CH * CW
    c = out["counts"].cpu().numpy()

    # the same set through round 2-3's wobbling-stripe kernel (MSPA_PAIR_WORD_STRIPES; the rectangular-tile kernel is the
    # default since round 4), and the fused compacted set: bitset + 4 B per VISIBLE pixel + a count per tile, no dense table
    ms_ws = timed(lambda: engine.pair_reproject(depth, mats, pairs, (CH, CW), out, flags=flags | _lib.PAIR_WORD_STRIPES))
    kern_ws = _lib.load().mspa_pair_reproject_last_kernel()
    comp = engine.alloc_pair_correspondences(n_pairs, (CH, CW), device)
    ms_comp = timed(lambda: engine.pair_correspondences(depth, mats, pairs, (CH, CW), comp, flags=flags))
    kern_comp = _lib.load().mspa_pair_reproject_last_kernel()
    n_vis = float(comp["counts"][:, 1].sum().item())
    bcomp = (2 * 2 * DH * DW + CH * CW / 8) * n_pairs + 4.0 * n_vis + 4.0 * comp["tile_counts"].numel()
    ws = int(_lib.load().mspa_pair_correspondences_workspace_bytes(n_pairs, DH, DW, CH, CW, flags))
    compact = {"shape": "colour 1296x968 over depth 640x480", "pairs": n_pairs, "kernel_ms": round(ms_comp, 4),
               "kernel": "mspa::pair_fast_tight_kernel<compact, SCALED> (rectangular tiles)" if kern_comp == _lib.KERNEL_PAIR_FAST_RECT
               else f"kernel id {kern_comp}", "workspace_bytes": ws,
               "ms_per_1000_pairs": round(ms_comp / n_pairs * 1000, 4), "pairs_per_s_1gpu": round(n_pairs / (ms_comp * 1e-3), 1),
               "bytes_per_pair": int(bcomp / n_pairs),
               "bytes_formula": "2 x 2 B x 307 200 + 1 254 528 / 8 (bitset) + 4 B x n_visible + 4 B x 441 tiles (n_visible from this launch's counters)",
               "achieved_GBs": round(bcomp / (ms_comp * 1e-3) / 1e9, 1), "frac": round(bcomp / (ms_comp * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "visible_fraction": round(float(c[:, 1].sum() / max(1, c[:, 0].sum())), 4)}
    return {"shape": "colour 1296x968 over depth 640x480", "pairs": n_pairs, "kernel_ms": round(ms, 4),
            "word_stripes_ms_per_1000_pairs": round(ms_ws / n_pairs * 1000, 4) if kern_ws == _lib.KERNEL_PAIR_FAST_SCALED else None,
            "compact": compact,
            "kernel": "mspa::pair_fast_tight_kernel<corr, SCALED> (rectangular tiles)" if kern == _lib.KERNEL_PAIR_FAST_RECT else f"kernel id {kern}",
            "ms_per_1000_pairs": round(ms / n_pairs * 1000, 4), "pairs_per_s_1gpu": round(n_pairs / (ms * 1e-3), 1),
            "colour_pixels_per_s": round(n_pairs * CH * CW / (ms * 1e-3), 1),
            "bytes_per_pair": int(bpp), "bytes_formula": "2 x 2 B x 307 200 (both depth frames, once) + 4.125 B x 1 254 528 (bitset + pixel index per colour pixel)",
            "achieved_GBs": round(bpp * n_pairs / (ms * 1e-3) / 1e9, 1),
            "frac": round(bpp * n_pairs / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "visible_fraction": round(float(c[:, 1].sum() / max(1, c[:, 0].sum())), 4), "pairs_rule": info["rule"],
            "timing": f"{warm} untimed + {steps} timed launches per leg, one HIP event pair (one untimed pass of the first leg before)"}


def time_track_geometry(device, T=300, P=256, n_scenes=256, reps=5):
    """K5a (mspa_track_to_world: camera->world + normalised projection + validity, OM_C:446-454, 293-315) on TAPVid-3D-shaped
    blocks -- BASELINE.json configs[3].  One block (T = 300 frames x P = 256 tracks) is 3.7 MB: a single launch
    is latency-bound, so the leg also times `n_scenes` blocks concatenated along the frame axis (every frame carries its own
    camera matrix, so scenes batch into one launch).  Bytes: SURVEY.md 8d's K5 figure T*P*(12 in + 12 out) + T*128 + T*P is
    written for float32 points; the path computes in the reference's float64, so the bytes actually moved are
    T*P*(24 in + 24 world + 16 uvn + 1 ok) + T*128 -- both fractions are printed."""
    import torch
    from mspa import engine, synth
    tr = synth.make_tracks(300, T=T, P=P, n_groups=8)
    c2w = np.linalg.inv(tr.extrinsics_w2c).reshape(T, 16)
    out = {}
    for name, n in (("one_block", 1), ("batched", n_scenes)):
        tracks = torch.from_numpy(np.ascontiguousarray(tr.tracks_XYZ)).to(device).repeat(n, 1, 1)
        cam = torch.from_numpy(c2w).to(device).repeat(n, 1)
        want = ("world", "uvn", "ok")
        for _ in range(2):
            engine.track_to_world(tracks, cam, tr.fx_fy_cx_cy, tr.image_hw, want)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            engine.track_to_world(tracks, cam, tr.fx_fy_cx_cy, tr.image_hw, want)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        Tn = T * n
        b_8d = Tn * P * 24 + Tn * 128 + Tn * P
        b_f64 = Tn * P * (24 + 24 + 16 + 1) + Tn * 128
        out[name] = {"scenes": n, "frames": Tn, "tracks": P, "ms_per_launch": round(ms, 4),
                     "scenes_per_s": round(n / (ms * 1e-3), 1), "track_points_per_s": round(Tn * P / (ms * 1e-3), 1),
                     "bytes_8d_formula": int(b_8d), "frac_8d_formula": round(b_8d / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "bytes_f64_moved": int(b_f64), "achieved_GBs": round(b_f64 / (ms * 1e-3) / 1e9, 1),
                     "frac": round(b_f64 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        del tracks, cam
    out["includes"] = "launch + allocation of the three outputs by the caching allocator (one HIP event pair around the launches)"
    return out


def _disk_scenes(n_scenes, n_frames, n_points):
    """Synthetic scenes for the from-disk leg: 8 rendered 640x480 frames each, referenced n_frames times under distinct
    image ids (every file is written and decoded separately)."""
    from mspa import synth
    out = []
    for k in range(n_scenes):
        base = synth.make_scene(5000 + k, n_points=n_points, n_frames=8, color_hw=(H, W), depth_hw=(H, W), invalid_pose_frac=0.0,
                                with_color=False, scene_id=f"scene{5000 + k:04d}_00")
        ids = base.image_ids
        E = {f"{5 * f:05d}": base.E[ids[f % 8]] for f in range(n_frames)}
        depth = {f"{5 * f:05d}": base.depth[ids[f % 8]] for f in range(n_frames)}
        out.append(synth.SynthScene(base.scene_id, base.K, base.A, E, base.points, depth, {}, base.color_hw, base.depth_hw,
                                    base.boxes))
    return out


def time_dropin_sweep(n_scenes=8, n_frames=64, n_points=131072, num_workers=None):
    """The drop-in entry point on on-disk inputs, one GPU: ``calculate_frames_relations.run_split`` (CFR:200-253) over scenes
    written in the reference's layout (scene-info pickle, posed_images/<scene>/<id>.png 16-bit depth, aligned_points.npy):
    native PNG ingest on `num_workers` host threads -> pinned staging -> H2D on the copy stream -> K1 + K2 + K4 -> parquet row
    groups.  Reports scenes/s and the seconds each stage was busy (they overlap, so they sum to more than the wall time)."""
    import shutil
    import tempfile
    from mspa import synth, sweep
    import spatial_engine.camera_movement.calculate_frames_relations as CFR
    from spatial_engine.utils.scannet_utils.handler import info_handler as IH
    from mspa import hostinfo
    # the reference's worker count (CFR:280), cut to the CPUs this process may really use (cgroup quota: mspa/hostinfo.py)
    num_workers = num_workers or min(25, hostinfo.effective_cpus())
    root = tempfile.mkdtemp(prefix="mspa_dropin_")
    try:
        t0 = time.perf_counter()
        paths = synth.write_scannet_layout(_disk_scenes(n_scenes, n_frames, n_points), root, compress_level=6)
        t_write_inputs = time.perf_counter() - t0
        png_bytes = sum(os.path.getsize(os.path.join(b, n)) for b, _, fs in os.walk(paths["posed_images_root"]) for n in fs
                        if n.endswith(".png"))
        orig_init = IH.SceneInfoHandler.__init__

        def init(self, info_path, *a, **k):                          # run_split builds its handler with the default roots
            orig_init(self, info_path, posed_images_root=paths["posed_images_root"], instance_data_root=paths["instance_data_root"])
        IH.SceneInfoHandler.__init__ = init
        try:
            import contextlib
            import io
            runs, throttled = [], []
            # Seven passes over the same 8 scenes; the median of the last three is reported, every pass's rate is kept.  The first
            # passes warm the page cache, the upload slots, the decode threads -- and the allocator: glibc stops mapping and
            # unmapping its large blocks (every page faulted in again by 25 threads) only after a few of them have been freed;
            # a fresh process ran its first 3 passes at 88 scenes/s and every later 8-scene sweep at 190-240
            # (tools/dropin_order_check.py).  A real split has hundreds of scenes: the steady state is what it runs at.
            for rep in range(7):
                tm = sweep.Timings()
                th0 = hostinfo.throttle_stats()
                with contextlib.redirect_stdout(io.StringIO()):
                    t0 = time.perf_counter()
                    CFR.run_split(paths["info_path"], os.path.join(root, f"out{rep}", "pairs.parquet"),
                                  os.path.join(root, f"warn{rep}.txt"), num_workers=num_workers, keep=False, timings=tm)
                    dt = time.perf_counter() - t0
                th1 = hostinfo.throttle_stats()
                throttled.append({k: th1[k] - th0.get(k, 0) for k in th1})
                runs.append((dt, tm.as_dict()))
        finally:
            IH.SceneInfoHandler.__init__ = orig_init
        dt, stages = sorted(runs[-3:], key=lambda r: r[0])[1]
        n_pairs = n_scenes * n_frames * (n_frames - 1) // 2
        return {"entry_point": "spatial_engine.camera_movement.calculate_frames_relations.run_split", "scenes": n_scenes,
                "frames_per_scene": n_frames, "vertices": n_points, "num_workers": num_workers,
                "seconds": round(dt, 4), "scenes_per_s": round(n_scenes / dt, 2), "frames_per_s": round(n_scenes * n_frames / dt, 1),
                "pair_rows_per_s": round(n_pairs / dt, 1), "png_MB_per_s": round(png_bytes / dt / 1e6, 1),
                "stage_busy_s": {"decode (PNG read + inflate + np.load, loader threads)": stages.get("decode"),
                                 "stage (pinned staging + H2D enqueue, copy stream)": stages.get("stage"),
                                 "produce (K1 + K2 + K4 + the scene's small D2H)": stages.get("produce"),
                                 "encode (owner rank: arrow table + parquet row group bytes, on the sweep's encoder threads)": stages.get("encode_deferred"),
                                 "consume (rank 0's writer thread: warnings + splice of the encoded row groups)": stages.get("consume"),
                                 "write (splice only)": stages.get("write")},
                "first_pass_seconds": round(runs[0][0], 4), "inputs_written_in_s": round(t_write_inputs, 2),
                "png_bytes": int(png_bytes), "statistic": "median of the last three of seven passes over the same 8 scenes (page cache warm: disk is not what is measured)",
                "passes_scenes_per_s": [round(n_scenes / r[0], 1) for r in runs],
                "host_cpus": hostinfo.describe(),
                "passes_cfs_throttling": throttled,
                "throttling_note": "per pass: CFS periods in which this container's threads were frozen by its cgroup CPU quota "
                                   "(`nr_throttled`) and for how long (`throttled_usec`, summed over CPUs) -- the cause of round "
                                   "5's 2 x pass-to-pass alternation (25 threads x 4 scenes in flight against a 16-CPU quota); "
                                   "thread counts now follow the quota (mspa/hostinfo.py, csrc/host_pool.h)"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def time_dropin_sweep_ranks(ranks=(1, 2, 4), n_scenes=32, n_frames=320, passes=2, per_rank=2, timeout_s=240, num_workers=None):
    """`variants.dropin_sweep_ranks`: both split-sweeping drop-in entry points (CFR:200-253, MVI:127-177) over ScanNet-sized
    on-disk scenes with 1, 2 and 4 ranks sharing this GPU over gloo, each world size in its own processes
    (tools/dropin_ranks.py): scenes/s including the writing, rank 0's writer busy time, every rank's wait at the window
    exchange, and whether the files equal the one-rank run's byte for byte.  The kernels are < 1 % of a scene, so this is the
    host-side scaling of an N-GPU job; the GPU and its PCIe link are the one thing the ranks share here."""
    import shutil
    import tempfile
    mod = _tool("dropin_ranks")
    root = tempfile.mkdtemp(prefix="mspa_dropin_ranks_")
    try:
        res = mod.drive(ranks=tuple(ranks), n_scenes=n_scenes, n_frames=n_frames, workers=num_workers, passes=passes,
                        per_rank=per_rank, timeout_s=timeout_s, keep_root=root)
        # once more, ONE rank with the sweeps' own defaults at this size: depth frames inflated on the MI355X
        # (csrc/device_ingest.hip), windows of 8 scenes, 8 loader threads -- what one GPU of a node does with its share of a split
        try:
            # (its own inputs, 96 scenes: a pass over the 32 above is a quarter of a second, a third of it the pipeline filling and draining)
            dev = mod.drive(ranks=(1,), n_scenes=96, n_frames=n_frames, workers=8, passes=4, per_rank=8, timeout_s=150, decode="device")
            res["one_rank_device_decode"] = dict(dev["worlds"].get("1", {}), scenes=96, num_workers=8, window_scenes_per_rank=8,
                                                 depth_decode="device", passes=4)
        except Exception as e:                               # informational
            res["one_rank_device_decode"] = {"skipped": f"{type(e).__name__}: {e}"}
        return res
    finally:
        shutil.rmtree(root, ignore_errors=True)


def _tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mspa_tool_" + name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def time_device_decode(streams=4096):
    """`variants.device_decode`: the depth decode on the MI355X by itself (tools/device_ingest_bench.py) -- `streams` 640 x 480
    16-bit depth PNGs (level 6, Pillow's adaptive row filters) resident in HBM as compressed bytes, inflated by one wave each
    (mspa_inflate_blocks_device incl. the Adler-32 pass) and un-filtered (mspa_png_unfilter_device), HIP events around the two
    calls; beside it the host reader on the CPUs this container may use.  Instruction-bound, not HBM-bound: the roofline of
    this kernel is its instruction count (profiles/r06_inflate_v5_pmc.md), so what is reported is frames/s and the ratio.
    4 096 streams = 16 waves per compute unit, what a wave's 104 registers and 9 KB of LDS (2 KB output ring) allow; the leg ran 3 584
    (14 per unit, 11 KB of LDS each) until the ring shrank."""
    r = _tool("device_ingest_bench").run(streams=streams, reps=3, with_composed=False)
    # the roofline that bounds it: instruction issue.  Instruction counts per frame from the committed PMC passes of the same
    # build (not measured in this run); cycles from this run's inflate time at the device's clock.
    pmc = {"salu": 3.83e6, "branch": 1.35e6, "valu": 3.07e6, "lds": 0.34e6, "source": "profiles/r06_inflate_v6_pmc.md (committed PMC pass, NOT measured in this run)"}
    try:
        import torch
        from mspa import _lib
        info = _lib.device_info(torch.cuda.current_device())
        n_cu, clock = int(info["n_cu"]), float(info["clock_khz"]) * 1e3
        cyc = r["device"]["inflate_adler_ms"] * 1e-3 * clock                      # (the Adler pass is < 1 % of it)
        per_cu = streams / n_cu
        tot = pmc["salu"] + pmc["branch"] + pmc["valu"] + pmc["lds"]
        r["roofline_issue"] = {"bound": "instruction issue (one scalar + one vector instruction per cycle and CU at most; DEFLATE's chain is scalar)",
                               "instructions_per_frame": round(tot), "waves_per_cu": round(per_cu, 1),
                               "achieved_instr_per_cycle_per_cu": round(per_cu * tot / cyc, 3),
                               "salu_issue_frac": round(per_cu * pmc["salu"] / cyc, 3), "valu_issue_frac": round(per_cu * pmc["valu"] / cyc, 3),
                               "counts": pmc}
    except Exception as e:                                   # informational
        r["roofline_issue"] = {"skipped": f"{type(e).__name__}: {e}"}
    # the same frames without their sensor noise: an adaptive PNG writer then gives practically every row the Paeth filter (as it
    # does for real sensor depth of smooth surfaces) -- fewer literals for the inflate, the skewed pipeline for the un-filter
    try:
        sm = _tool("device_ingest_bench").run(streams=streams, reps=3, with_composed=False, smooth=True)
        r["smooth_frames"] = {"frames": sm["frames"], "png_bytes_per_frame": sm["png_bytes_per_frame"], "device": sm["device"], "host": {k: sm["host"][k] for k in ("threads", "frames_per_s")},
                              "device_over_host": sm["device_over_host"], "status_nonzero": sm["status_nonzero"],
                              "bit_identical_to_the_rendered_frames": sm["bit_identical_to_the_rendered_frames"]}
    except Exception as e:                                   # informational
        r["smooth_frames"] = {"skipped": f"{type(e).__name__}: {e}"}
    r["what"] = ("replaces the per-frame cv2.imread / zlib.decompress of info_handler.py:149-155 and extract_posed_images.py:49-57 for "
                 "streaming sweeps of >= 1 024 frames")
    return r


def run_scene_workload(args, rank, world, device, dist_ctx, share):
    """--workload scenes: BASELINE.json configs[2] in the bench's shape.  `scenes_per_gpu` x world ScanNet-sized scenes
    (131 072 vertices; 160..400 frames, so costs differ) are dealt longest-first (shard.lpt_assign, cost F^2 N / 64 + F N); each
    rank keeps its scenes resident and a step = K1 + K2 + K4 over all of them, the [n_pairs, 7] float64 pair-table rows
    written into one job tensor which is collated with ONE all_gather (shard.collate_records: RCCL over xGMI) inside the timed
    region -- SURVEY.md 8e's exchange, 56 B per frame pair.  `value` = pair-table rows of all ranks / slowest rank's time."""
    import torch
    import torch.distributed as dist
    from mspa import engine, shard, synth
    from mspa.scene import SceneOnDevice
    n_total = args.scenes_per_gpu * world
    frames = [160 + 40 * ((3 * k) % 7) for k in range(n_total)]                   # 160 .. 400, the every-5th-frame range
    costs = [shard.scene_cost(f, args.scene_points) for f in frames]
    bins = shard.lpt_assign(costs, world)
    base = synth.make_scene(4000 + rank, n_points=args.scene_points, n_frames=8, color_hw=(H, W), depth_hw=(H, W),
                            invalid_pose_frac=0.0, with_color=False)
    ids = base.image_ids
    scenes = []
    for k in bins[rank]:
        F = frames[k]
        E = {f"{f:05d}": base.E[ids[f % 8]] for f in range(F)}
        depth = {f"{f:05d}": base.depth[ids[f % 8]] for f in range(F)}
        scenes.append((k, SceneOnDevice(base.K, base.A, E, depth, (H, W), base.points, device)))
    n_rows = sum(frames[k] * (frames[k] - 1) // 2 for k in bins[rank])
    job = torch.empty((n_rows, 7), dtype=torch.float64, device=device)
    pair_idx = {k: engine.all_pairs(frames[k], device) for k in bins[rank]}

    def step():
        lo = 0
        for k, sc in scenes:
            sc._vis = None                                                        # K1 runs again every step
            vis = sc._visibility()
            pairs = pair_idx[k]
            n = pairs.shape[0]
            rows = job[lo:lo + n]
            rows[:, 0] = k
            rows[:, 1:3] = pairs.to(torch.float64)
            rows[:, 3] = engine.scene_overlap(vis["bits"])
            rows[:, 4:7] = engine.pair_pose(*sc.pose_tables(), pairs)[:, 0:3]
            lo += n
        return shard.collate_records(job, dist_ctx) if dist_ctx is not None else job

    for _ in range(max(1, args.warmup)):
        table = step()
    if dist_ctx is not None:
        dist_ctx.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        table = step()
    ev1.record()
    torch.cuda.synchronize()
    if dist_ctx is not None:
        dist_ctx.barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1) / args.steps
    rank_walls, rows_all = [wall], [n_rows]
    if dist_ctx is not None:
        t = torch.tensor([wall, float(n_rows)], dtype=torch.float64, device=dist_ctx.collective_device)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t, group=dist_ctx.group)
        rank_walls, rows_all = [float(p[0]) for p in parts], [int(p[1]) for p in parts]
        wall = max(rank_walls)
    total_rows = sum(rows_all)
    if rank != 0:
        return None
    assert table.shape[0] == total_rows, "the collated pair table does not hold every rank's rows"
    got = np.sort(np.unique(table[:, 0].cpu().numpy().astype(np.int64)))
    assert got.tolist() == list(range(n_total)), "a scene's rows are missing from the collated table"
    # algorithmic bytes of this rank's step: K1 compulsory (vertices once per scene, every depth frame, every bitset row) +
    # K2 (every bitset row once per scene + 8 B per pair out) + the row table written once
    mine = bins[0]
    b_k1 = sum(24 * args.scene_points + frames[k] * (2 * H * W + args.scene_points // 8) for k in mine)
    b_k2 = sum(frames[k] * args.scene_points // 8 for k in mine) + 56 * rows_all[0]
    rates = [r * args.steps / w for r, w in zip(rows_all, rank_walls)]
    line = {"metric": "frame-pairs/sec MultiSPA geometry pipe (640x480 RGB-D)", "value": round(total_rows * args.steps / wall, 1),
            "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "camera_movement pair table (BASELINE.json configs[2] shape): K1 vertex visibility + K2 all-pairs "
                                   "overlap + K4 relative pose over whole scenes, every frame pair i < j of a scene is one row",
                       "scenes_per_gpu": args.scenes_per_gpu, "scenes": n_total, "vertices_per_scene": args.scene_points,
                       "frames_per_scene": sorted(set(frames)), "assignment": "shard.lpt_assign (longest first, cost F^2 N / 64 + F N)",
                       "rows_per_rank": {"min": min(rows_all), "max": max(rows_all)},
                       "collation": ("none (1 GPU)" if dist_ctx is None else
                                     "ONE all_gather of every rank's [rows, 7] float64 pair-table rows per step, inside the timed "
                                     "region (" + ("gloo: ranks share GPUs" if share else "RCCL") + ")"),
                       "collation_backend": dist_ctx.backend if dist_ctx is not None else None,
                       "bytes_collated_per_step": 56 * total_rows if dist_ctx is not None else 0,
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "achieved": round((b_k1 + b_k2) / (dev_ms * 1e-3) / 1e9, 1),
                         "frac": round((b_k1 + b_k2) / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": "mspa::vertex_visibility_compact_kernel + overlap tiles + pair_pose (whole step on rank 0)",
                         "kernel_ms": round(dev_ms, 4), "bytes_per_launch": int(b_k1 + b_k2),
                         "note": "compulsory bytes of rank 0's step / its HIP-event time; K1 is latency-bound (DESIGN 7.4), so this "
                                 "fraction is low by construction -- the headline roofline is the K3 line of the default workload"},
            "cpu_baseline": None,
            "per_rank_pairs_per_s": {"min": round(min(rates), 1), "max": round(max(rates), 1)},
            "device": None}
    if dist_ctx is not None:
        line["rccl_world"] = dist.get_world_size()
        line["gpus_shared"] = bool(share)
    return line


_CPU_SCENE = None


def _cpu_pair(job):
    """One frame pair through the NumPy restatement of the reference path (worker side)."""
    from oracle import np_oracle as O
    depth, K, E, A, color = _CPU_SCENE
    a, b = job
    r = O.frame_pair(depth[a], depth[b], K, E[a], E[b], A, (H, W), color)
    return r["n_valid"], r["n_vis"]


def _cpu_worker_init(payload_path):
    """Pool workers are SPAWNED (fresh interpreters: the parent holds a live HIP runtime by now and must not be
    forked); the scene reaches them as one .npz file."""
    global _CPU_SCENE
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)                 # one BLAS thread per worker, as OMP_NUM_THREADS=1 would
    except Exception:
        pass
    z = np.load(payload_path)
    _CPU_SCENE = (z["depth"], z["K"], z["E"], z["A"], np.zeros((H, W, 3), dtype=np.uint8))


def cpu_baseline(sc, base_pairs, budget_s, with_pool=True, timed=None):
    """Time the NumPy restatement of the reference path (oracle/np_oracle.frame_pair) on a bounded sample
    of the step's pairs: (i) one process, (ii) multiprocessing.Pool(min(25, cores)) -- the reference's own
    worker count (CFR:280).  The oracle is used here as the thing timed BESIDE the product, never inside it.

    ``timed`` = (pairs [n, 2] frame indices of the timed launch, the resident depth tensor, the launch's `counts` [n, 2]):
    the one-process leg then runs the oracle on the VERY frames the timed launch read (pair p's two replica frames, downloaded
    outside the clock) and its (n_valid, n_vis) -- the sizes of the reference's validity and visibility masks, OPS:272-296 and
    IH:375-386 -- are compared with what the timed launch wrote for the same pair: `parity_in_run`."""
    global _CPU_SCENE
    import multiprocessing as mp
    import tempfile
    ids = sc.valid_image_ids
    depth = np.stack([sc.depth[i] for i in ids])
    E = np.stack([sc.E[i] for i in ids])
    color = np.zeros((H, W, 3), dtype=np.uint8)
    _CPU_SCENE = (depth, sc.K, E, sc.A, color)
    jobs = [(int(p[0]), int(p[1])) for p in base_pairs]
    parity = None
    if timed is not None:
        t_pairs, t_depth, t_counts = timed
        gpu_counts = t_counts.detach().cpu().numpy().reshape(-1, 2)
        parity = {"pairs": 0, "mismatches": 0, "first_mismatch": None,
                  "what": "(n_valid, n_vis) of oracle/np_oracle.frame_pair on the frames the timed launch read (replica frames "
                          "downloaded from HBM) == the `counts` the timed launch wrote for the same pairs"}
    n, el = 0, 0.0
    while True:
        if timed is not None:                 # this pair's own two frames, as resident on the device (outside the clock)
            fa, fb = int(t_pairs[n][0]), int(t_pairs[n][1])
            pair_depth = t_depth[[fa, fb]].cpu().numpy().view(np.uint16)
            _CPU_SCENE = (pair_depth, sc.K, np.stack([E[jobs[n][0]], E[jobs[n][1]]]), sc.A, color)
            job = (0, 1)
        else:
            job = jobs[n]
        t0 = time.perf_counter()
        got = _cpu_pair(job)
        el += time.perf_counter() - t0
        if parity is not None:
            parity["pairs"] += 1
            want = (int(gpu_counts[n][0]), int(gpu_counts[n][1]))
            if tuple(int(x) for x in got) != want:
                parity["mismatches"] += 1
                if parity["first_mismatch"] is None:
                    parity["first_mismatch"] = {"pair": n, "frames": [fa, fb], "oracle": [int(got[0]), int(got[1])], "gpu": list(want)}
        n += 1
        if (el > budget_s / 2 and n >= 8) or n >= len(jobs):
            break
    _CPU_SCENE = (depth, sc.K, E, sc.A, color)
    single = n / el
    from mspa import hostinfo
    # the reference's Pool(25) (CFR:280); `cores` is what the pool can really burn: a container's cgroup quota may be far
    # below os.cpu_count() (16 CPUs of 256 on the MI355X boxes), and 25 workers then share 16 CPUs' worth of time
    workers = min(25, os.cpu_count() or 1)
    usable = min(workers, hostinfo.effective_cpus())
    pool_rate, pool_jobs = None, []
    if with_pool:
        pool_jobs = jobs[:max(workers * 4, int(single * workers * budget_s / 2))][:len(jobs)]
        try:
            with tempfile.TemporaryDirectory() as td:
                payload = os.path.join(td, "scene.npz")
                np.savez(payload, depth=depth, K=sc.K, E=E, A=sc.A)
                with mp.get_context("spawn").Pool(workers, initializer=_cpu_worker_init, initargs=(payload,)) as pool:
                    pool.map(_cpu_pair, pool_jobs[:workers])                       # warm the workers
                    t1 = time.perf_counter()
                    pool.map(_cpu_pair, pool_jobs, chunksize=max(1, len(pool_jobs) // (workers * 4)))
                    pool_rate = len(pool_jobs) / (time.perf_counter() - t1)
        except Exception as e:                                                  # no /dev/shm etc.: report single only
            print(f"[bench] pool baseline skipped: {e}", file=sys.stderr)
    return {"value": round(single, 3), "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n} of the step's 640x480 pairs through oracle/np_oracle.frame_pair "
                      f"(NumPy {np.__version__}, 1 process, in-memory images) in {el:.1f} s",
            "pool": None if pool_rate is None else {
                "value": round(pool_rate, 2), "cores": usable, "pool_processes": workers,
                "sample": f"{len(pool_jobs)} pairs over multiprocessing.Pool({workers}), 1 BLAS thread each"
                          + (f"; the container's CPU quota lets them use {usable} CPUs" if usable < workers else "")},
            "host_cores_available": hostinfo.effective_cpus(), "host_cpus": hostinfo.describe(), "parity_in_run": parity}


def live_traffic(args):
    """`roofline.traffic` measured in THIS run on THIS box: the headline step re-run twice as a child process under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only -- never combined with
    sys / hip / hsa traces), mean per dispatch of the K3 kernel, FETCH_SIZE doubled (gfx950 reports half of the streamed bytes:
    profiles/r01_counter_calibration.md, MI355X_MICROARCH.md), both in KiB.  Returns (bytes per launch, kernel name, dispatches)
    or None when rocprofv3 is missing or a pass fails -- the committed value is used then, and the line says which it was."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "4", "--warmup", "1", "--pairs", str(args.pairs), "--frames",
             str(args.frames), "--base-frames", str(args.base_frames), "--scene-points", str(args.scene_points), "--workload",
             args.workload, "--variant", args.variant, "--mode", args.mode, "--stream", args.stream, "--walk-step", str(args.walk_step),
             "--target-step", str(args.target_step), "--spinup-ms", "0", "--no-cpu-baseline", "--no-scene-legs", "--no-sweep",
             "--also", "none", "--no-live-traffic"]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MSPA_BENCH_FORCE_DIST"):
        env.pop(k, None)
    got = {}
    kernel = None
    out_dir = tempfile.mkdtemp(prefix="mspa_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out_dir, counter)
            r = subprocess.run([tool, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "pair_" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                        vals.append(float(row["Counter_Value"]))
                        kernel = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if r.returncode != 0 or not vals:
                return None
            got[counter] = (sum(vals) / len(vals), len(vals))
        return int(got["FETCH_SIZE"][0] * 2048 + got["WRITE_SIZE"][0] * 1024), kernel, min(got["FETCH_SIZE"][1], got["WRITE_SIZE"][1])
    except Exception as e:                                     # a profiler that cannot run here must not cost the line
        print(f"[bench] live traffic measurement skipped: {type(e).__name__}: {e}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def committed_traffic(key):
    """PMC-measured HBM bytes per launch for a named leg, from the committed profiles/traffic.json (collected with
    tools/profile.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very command).  NOT measured in
    this run -- counters cannot be read from inside the process -- and reported as such (`traffic_source`)."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None
    return json.load(open(tfile)).get(key)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0"))
    if world == 0:
        if args.gpus > 1:
            self_launch(args)                       # does not return
        world = 1
    import torch
    from mspa import _lib, engine, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the product path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    # fewer GPUs than ranks (a 1-GPU box asked for --gpus 2): ranks share devices and collate over gloo -- RCCL refuses
    # two ranks on one device.  A launch check; the line says so.
    share = bool(os.environ.get("MSPA_BENCH_SHARE_GPU")) or n_dev < world
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    _lib.load()
    if args.workload == SCENE_WORKLOAD:
        dist_ctx = (shard.init_distributed(device, backend="gloo" if share else None)
                    if (world > 1 or os.environ.get("MSPA_BENCH_FORCE_DIST")) else None)
        line = run_scene_workload(args, rank, world, device, dist_ctx, share)
        if rank == 0:
            line["device"] = _lib.device_info(dev_index)
        if dist_ctx is not None:
            dist_ctx.barrier()
            dist_ctx.close()
        if rank == 0:
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            print(json.dumps(line), flush=True)
        return
    sc = make_base_scene(args, rank)
    depth, mats, rgb, nb, reps = build_inputs(args, rank, device, sc)
    overlap = scene_overlap_table(sc, device)
    pairs_np, base_pairs, winfo = workload_pairs(overlap, nb, reps, args.pairs, args.workload, rank)
    pairs = torch.from_numpy(pairs_np).to(device)
    # MSPA_BENCH_FORCE_DIST=1 exercises the RCCL collation path with a single rank (1-GPU boxes)
    dist_ctx = (shard.init_distributed(device, backend="gloo" if share else None)
                if (world > 1 or os.environ.get("MSPA_BENCH_FORCE_DIST")) else None)
    stream = stream_hint(args, int(depth.shape[0]))
    # `cold_ms_per_step`: rounds 1-2's method, kept beside the steady-state number so that rounds stay comparable -- no clock
    # spin-up (the part comes straight from the host-side input preparation), an event pair around every step, W warm-up steps.
    # Rank 0 of a single-GPU run only; it runs BEFORE the headline (afterwards the clock would be warm).
    cold_ms = None
    if world == 1 and not os.environ.get("MSPA_BENCH_FORCE_DIST"):
        _, cold_ms, _cold_out = time_variant(args.variant, args.mode, depth, mats, rgb, pairs, args.steps, args.warmup, None, stream,
                                             spinup_ms=0.0, per_step_events=True)
        del _cold_out
    wall, kern_ms, out = time_variant(args.variant, args.mode, depth, mats, rgb, pairs, args.steps, args.warmup,
                                      dist_ctx, stream, spinup_ms=args.spinup_ms)
    rank_walls = None
    if dist_ctx is not None:
        # every rank's own wall time of the timed region (they all include the same final barrier; what differs is how long a
        # rank waited there): per-rank rates in the line make a scaling record self-validating
        import torch.distributed as dist
        w_t = torch.tensor([wall], dtype=torch.float64, device=dist_ctx.collective_device)
        w_all = [torch.zeros_like(w_t) for _ in range(dist_ctx.world)]
        dist.all_gather(w_all, w_t, group=dist_ctx.group)
        rank_walls = [float(x.item()) for x in w_all]
        rccl_world = dist.get_world_size()
        wall = dist_ctx.max_over_ranks(wall)
    pairs_per_step = args.pairs * world
    value = pairs_per_step * args.steps / wall
    spec = VARIANTS[args.variant]
    bytes_per_launch = variant_bytes(args.variant, args.pairs, out)
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9

    def vis_fraction(o):
        c = o["counts"].cpu().numpy().reshape(-1, 2)
        return round(float(c[:, 1].sum() / max(1, c[:, 0].sum())), 4)

    def with_traffic(d, v, m, wl, k_ms):
        """`traffic_frac` = PMC-measured HBM bytes (committed profile of this leg) / this run's kernel time / peak, printed
        beside `frac` (the SURVEY 8d formula) for every point; a point whose formula exceeds the measured traffic by more than
        10 % is flagged: there the formula counts bytes the kernel never moves (depth-2 of culled tiles) and `traffic_frac`
        is the reading to use."""
        t = committed_traffic(f"{v}:{m}:{wl}") if args.pairs == 1000 else None
        if t:
            d["traffic"] = t["hbm_bytes_per_launch"]
            d["traffic_frac"] = round(t["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            d["traffic_source"] = t["source"]
            over = d["bytes_per_pair"] * args.pairs / t["hbm_bytes_per_launch"]
            d["formula_over_traffic"] = round(over, 3)
            if over > 1.10:
                d["frac_flag"] = ("formula bytes exceed the measured HBM traffic by more than 10 % (depth-2 is counted at 2 B per "
                                  "pixel although culled tiles never read it): read traffic_frac, not frac")
        else:
            d["traffic_frac"] = None
        return d

    dev_info = _lib.device_info(dev_index)

    def valu_roofline(v, m, wl, k_ms):
        """The instruction-issue roofline of a leg that is VALU-bound rather than HBM-bound: VALU instructions per launch
        (SQ_INSTS_VALU of the committed PMC pass, as `traffic`) x 64 lanes / this run's kernel time, against the chip's
        lane-instruction issue peak n_cu x 4 SIMDs x 16 lanes x clock.  Every instruction is counted at the 4-cycle full rate
        although float64 multiply / fma / divide-step issue at half of it or less (profiles/r04_valu_rates.md), so `frac` is a
        LOWER bound of how busy the pipe is."""
        t = committed_traffic(f"{v}:{m}:{wl}") if args.pairs == 1000 else None
        if not t or not t.get("valu_insts_per_launch"):
            return None
        peak = dev_info["n_cu"] * 4 * 16 * dev_info["clock_khz"] * 1e3
        ach = t["valu_insts_per_launch"] * 64.0 / (k_ms * 1e-3)
        return {"bound": "fp64_valu_issue", "unit": "lane-instr/s", "peak_lane_instr_per_s": peak, "achieved": round(ach, 1),
                "frac": round(ach / peak, 4), "valu_insts_per_launch": t["valu_insts_per_launch"],
                "valu_insts_per_wave": round(t["valu_insts_per_launch"] / max(1, t.get("waves_per_launch", 0) or 1), 1),
                "source": t.get("valu_source", t.get("source")) + " -- committed PMC pass, NOT measured in this run"}

    def leg(v, m, prs, steps, wl):
        w2, k2, o2 = time_variant(v, m, depth, mats, rgb, prs, steps, 1, None, stream)
        b2 = variant_bytes(v, args.pairs, o2)
        d = {"pairs_per_s_1gpu": round(args.pairs / (k2 * 1e-3), 1), "kernel_ms": round(k2, 4),
             "achieved_GBs": round(b2 / (k2 * 1e-3) / 1e9, 1),
             "frac": round(b2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "bytes_per_pair": int(b2 / args.pairs), "visible_fraction": vis_fraction(o2)}
        if VARIANTS[v].get("compact"):
            d["bytes_formula"] = "P * (2 + 2 + 1/8) + 4 * n_visible + 4 * n_tiles per pair (n_visible from this launch's counters)"
        if v == "dense":     # the set also writes 4 B/px of rgba, which SURVEY 8d's formula leaves out: both readings
            bw = (VARIANTS[v]["bytes_per_px"] + 4) * P * args.pairs
            d["frac_counting_rgba_out"] = round(bw / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            d["bytes_per_pair_counting_rgba_out"] = int(bw / args.pairs)
        if "note" in VARIANTS[v]:
            d["note"] = VARIANTS[v]["note"]
        if m == "exact":
            d["note"] = ("pair_exact_kernel: the reference's own operation order (five 3x4 products + IEEE division per pixel), "
                         "FP64-issue bound, not HBM bound -- the bit-exact path every float64-output call takes; the fast kernels "
                         "reproduce its integers")
        d = with_traffic(d, v, m, wl, k2)
        rv = valu_roofline(v, m, wl, k2)
        if rv:
            d["roofline_valu"] = rv
        return d

    extra, sweep = {}, {}
    head_vis = vis_fraction(out) if rank == 0 else None
    if rank == 0 and world == 1:      # informational single-GPU legs; with N > 1 every rank leaves together after the timed job
        short = max(3, args.steps // 4)
        if not args.no_sweep:         # the same variant on the other two points of the overlap sweep
            for kind in WORKLOADS:
                if kind == args.workload:
                    sweep[kind] = with_traffic({"pairs_per_s_1gpu": round(args.pairs / (kern_ms * 1e-3), 1),
                                                "kernel_ms": round(kern_ms, 4), "achieved_GBs": round(achieved, 1),
                                                "frac": round(achieved / HBM_PEAK_GBS, 4),
                                                "bytes_per_pair": int(bytes_per_launch / args.pairs), "visible_fraction": head_vis,
                                                "pairs": winfo, "headline": True}, args.variant, args.mode, kind, kern_ms)
                    continue
                p2, _, i2 = workload_pairs(overlap, nb, reps, args.pairs, kind, rank)
                sweep[kind] = dict(leg(args.variant, args.mode, torch.from_numpy(p2).to(device), max(short, 5), kind), pairs=i2)
        for lg in [v for v in _legs(args.also) if v != f"{args.variant}:{args.mode}"]:
            v, m = lg.split(":")
            extra[lg] = leg(v, m, pairs, short, args.workload)
        if not args.no_sweep and "compact:fast" in _legs(args.also) and args.variant != "compact":
            # the compacted set is what the correspondence head consumes: its own three-point sweep
            sweep_compact = {}
            for kind in WORKLOADS:
                if kind == args.workload:
                    sweep_compact[kind] = extra["compact:fast"]
                    continue
                p2, _, i2 = workload_pairs(overlap, nb, reps, args.pairs, kind, rank)
                sweep_compact[kind] = dict(leg("compact", "fast", torch.from_numpy(p2).to(device), max(short, 5), kind), pairs=i2)
            extra["sweep_compact"] = sweep_compact
        if not args.no_sweep:
            # the same pairs four times over in ONE launch (same mix of overlaps): does a launch of configs[1]'s size pay for the
            # ramp and the tail of its grid?  Measured: no -- within +-4 % of the 1 000-pair launch, either way.
            _, k4, _ = time_variant(args.variant, args.mode, depth, mats, rgb, pairs.repeat(4, 1), max(short, 5), 3, None, stream)
            extra["launch_size"] = {"variant": f"{args.variant}:{args.mode}", "pairs_per_launch": 4 * args.pairs,
                                    "kernel_ms_per_1000_pairs": round(k4 / 4 * 1000.0 / args.pairs, 4),
                                    "headline_kernel_ms_per_1000_pairs": round(kern_ms * 1000.0 / args.pairs, 4),
                                    "note": "informational: the timed step stays one launch of configs[1]'s batch"}
        if not args.no_scene_legs:
            extra["scannet_shape:fast"] = time_scannet_shape(device)
            extra["scannet_shape:compact"] = extra["scannet_shape:fast"].pop("compact")
            extra["scene"] = time_scene_kernels(device)
            extra["pipeline"] = time_scene_pipeline(device)
            extra["K5_track_geometry"] = time_track_geometry(device)
            if not args.no_dropin_sweep:
                try:
                    extra["dropin_sweep"] = time_dropin_sweep()
                except Exception as e:                       # e.g. no Pillow / no pyarrow on the box: the leg is informational
                    extra["dropin_sweep"] = {"skipped": f"{type(e).__name__}: {e}"}
            try:
                extra["device_decode"] = time_device_decode()
            except Exception as e:                           # informational: never costs the line
                extra["device_decode"] = {"skipped": f"{type(e).__name__}: {e}"}
            if not args.no_dropin_ranks:
                try:
                    extra["dropin_sweep_ranks"] = time_dropin_sweep_ranks()
                except Exception as e:                       # informational: never costs the line
                    extra["dropin_sweep_ranks"] = {"skipped": f"{type(e).__name__}: {e}"}
            t1 = committed_traffic("K1_vertex_visibility")
            if t1:
                k1 = extra["scene"]["K1_vertex_visibility"]["shuffled_worst_case"]   # the PMC passes ran on the shuffled cloud
                k1["frac_note"] = ("`compulsory_frac` = (24 N once + per image 2 DW DH + N/8) / time / peak: the bytes any K1 must "
                                   "move; `traffic_frac` = PMC-measured L2-miss traffic / time / peak (the vertex array is re-read "
                                   "by each of the 40 image groups, and 8 depth frames + 3 MB of vertices overflow a 4 MB L2)")
                k1["traffic"] = t1["hbm_bytes_per_launch"]
                k1["traffic_frac"] = round(t1["hbm_bytes_per_launch"] / (k1["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                k1["traffic_source"] = t1["source"]

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # after the GPU legs (the pair list comes from K1 + K2); at N > 1 only the 1-process leg, with half the budget,
        # while the other ranks wait at the final barrier
        cpu = cpu_baseline(sc, base_pairs, args.cpu_seconds if world == 1 else args.cpu_seconds / 2, with_pool=(world == 1),
                           timed=(pairs_np, depth, out["counts"]))

    if rank == 0:
        tkey = f"{args.variant}:{args.mode}:{args.workload}"
        t = committed_traffic(tkey) if args.pairs == 1000 else None
        traffic = t["hbm_bytes_per_launch"] if t else None
        traffic_committed = traffic
        traffic_source = (t["source"] + " -- committed PMC passes of this command, NOT measured in this run") if t else None
        live = None
        under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) or \
            "rocprof" in os.environ.get("LD_PRELOAD", "")            # never a profiler inside a profiler
        if world == 1 and not args.no_live_traffic and not under_profiler and not os.environ.get("MSPA_BENCH_FORCE_DIST"):
            live = live_traffic(args)
        if live:
            traffic = live[0]
            traffic_source = (f"measured in this run on this box: two child runs of this command under rocprofv3 --kernel-trace --pmc "
                              f"FETCH_SIZE / WRITE_SIZE (separate passes; mean of {live[2]} dispatches of {live[1]}; FETCH_SIZE x 2 KiB + "
                              "WRITE_SIZE x 1 KiB)")
        info = dev_info
        ceilings = measured_hbm_ceilings(device) if (world == 1 and not args.no_scene_legs) else None
        tight = args.mode != "exact"
        line = {
            "metric": "frame-pairs/sec MultiSPA geometry pipe (640x480 RGB-D)",
            "value": round(value, 1), "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "cold_ms_per_step": None if cold_ms is None else round(cold_ms, 4),
            "cold_method": "no clock spin-up, HIP event pair around every step, mean of the K per-step times after W warm-up "
                           "steps (rounds 1-2's method); `ms_per_step` / `roofline.kernel_ms` are the steady-state reading",
            "config": {"workload": "visual_correspondence unproject+reproject+occlusion kernel (K3) on "
                                   f"{args.pairs} 640x480 frame pairs per GPU per step (BASELINE.json configs[1]); pairs = "
                                   f"'{args.workload}' sample of the scene's overlap table: {winfo['rule']}",
                       "pairs": winfo, "visible_fraction": head_vis,
                       "scene": {"frames": nb, "vertices": args.scene_points, "trajectory": "sweep",
                                 "walk_step_m": args.walk_step, "target_step_m": args.target_step,
                                 "recipe": "SURVEY.md 8d (6x6x3 m room, 8 boxes, 5 mm depth noise, 7 % invalid pixels)"},
                       "variant": args.variant, "mode": args.mode, "outputs": list(spec["outputs"]),
                       "stream_hint": bool(stream and args.mode != "exact"),
                       "clock_spinup": {"ms": args.spinup_ms, "untimed_steps": SPINUP["steps"],
                                        "why": "the part needs ~30 ms under load to reach its sustained clock; the W warm-up "
                                               "steps and the K timed steps follow unchanged (--spinup-ms 0 disables it)"},
                       "pairs_per_step_per_gpu": args.pairs, "distinct_frames_per_gpu": int(depth.shape[0]),
                       "image": "640x480 depth u16 (+rgb u8x3 for dense)", "parallelism": f"dp{world}",
                       "collation": ("one all_gather of the job's per-pair records inside the timed region ("
                                     + ("gloo: ranks share GPUs" if share else "RCCL") + ")") if dist_ctx is not None
                       else "none (1 GPU)",
                       "collation_backend": dist_ctx.backend if dist_ctx is not None else None},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "traffic_committed": traffic_committed,
                         "traffic_GBs": round(traffic / (kern_ms * 1e-3) / 1e9, 1) if traffic else None,
                         "traffic_frac": round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                         "kernel": "mspa::pair_fast_tight_kernel" if tight else "mspa::pair_exact_kernel",
                         "kernel_ms": round(kern_ms, 4), "bytes_per_pair": int(bytes_per_launch / args.pairs),
                         "bytes_per_launch": int(bytes_per_launch),
                         "measured_ceilings_GBs": ceilings},
            "roofline_valu": valu_roofline(args.variant, args.mode, args.workload, kern_ms),
            "cpu_baseline": cpu,
            "parity_in_run": cpu.pop("parity_in_run") if cpu else None,
            "cpu_baseline_pool": None if not cpu or not cpu.get("pool") else dict(
                cpu["pool"], unit="frame-pairs/s", kind="port",
                what="the same NumPy restatement over multiprocessing.Pool(min(25, cores)) -- the reference's own fan-out (CFR:280)"),
            "sweep": sweep,
            "variants": extra,
            "device": info,
            "visible_fraction": head_vis,
        }
        if world > 1:
            line["gpus_shared"] = bool(share)
            line["physical_gpus"] = n_dev
        if rank_walls is not None:
            rates = [args.pairs * args.steps / w for w in rank_walls]
            line["rccl_world"] = rccl_world               # torch.distributed.get_world_size() of the communicator that collated
            line["per_rank_pairs_per_s"] = {"min": round(min(rates), 1), "max": round(max(rates), 1),
                                            "note": "each rank's own pairs / its own wall time of the timed region; `value` = all "
                                                    "ranks' pairs / the slowest rank's wall time"}
    if dist_ctx is not None:
        dist_ctx.barrier()          # leave together: no rank tears the communicator down under another one's feet
        dist_ctx.close()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which sits in libc's buffer
        # until it is flushed -- flush it first, then print
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(line), flush=True)
        if (line.get("parity_in_run") or {}).get("mismatches"):
            raise SystemExit(f"bench.py: the timed launch disagrees with the oracle: {line['parity_in_run']}")


if __name__ == "__main__":
    main()
