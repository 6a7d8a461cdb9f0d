"""bench.py through the driver's command shapes: `python bench.py --gpus N` with no launcher (self-launch) and
`python -m torch.distributed.run ... bench.py --gpus N` (the driver's N > 1 form).  On a 1-GPU box N = 2 shares the device
and collates over gloo -- the 2-rank control flow, sharding and collation are the real ones."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--pairs", "96", "--frames", "64", "--base-frames", "16", "--scene-points", "8192",
         "--no-scene-legs", "--also", "none", "--cpu-seconds", "1", "--no-live-traffic"]


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run(cmd, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def check_line(j, n):
    assert j["n_gpus"] == n and j["steps"] == 2 and j["warmup"] == 1 and j["unit"] == "frame-pairs/s"
    assert j["value"] > 0 and j["scaling"] == "weak" and j["dtype"] == "f64" and j["vs_baseline"] is None
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "traffic_source" in r
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    assert "cpu_baseline_pool" in j and "roofline_valu" in j
    # the timed launch's own counts against the oracle on the frames it read, inside the run
    assert j["parity_in_run"]["pairs"] >= 8 and j["parity_in_run"]["mismatches"] == 0
    assert j["config"]["pairs"]["rule"].startswith("equal quotas over the overlap bins 6..35")
    assert 0 < j["config"]["visible_fraction"] < 1


def test_bench_n1_plain_command():
    j = run([sys.executable, "bench.py", "--gpus", "1"] + SMALL)
    check_line(j, 1)
    assert set(j["sweep"]) == {"low", "vc", "high"} and j["sweep"]["vc"]["headline"]
    assert j["sweep"]["low"]["visible_fraction"] < j["sweep"]["vc"]["visible_fraction"] < j["sweep"]["high"]["visible_fraction"]


def test_bench_n2_self_launch_and_torchrun():
    j = run([sys.executable, "bench.py", "--gpus", "2"] + SMALL)
    check_line(j, 2)
    import torch
    assert j["gpus_shared"] == (torch.cuda.device_count() < 2)
    j = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), "bench.py", "--gpus", "2"] + SMALL)
    check_line(j, 2)


def test_bench_n1_with_the_rccl_collation_forced():
    """One rank, MSPA_BENCH_FORCE_DIST=1: the communicator is RCCL (backend "nccl"), the job's record table is collated by
    all_gather_into_tensor on DEVICE tensors inside the timed region and the ranks meet at barrier(device_ids=...) -- the
    branch every N > 1 run on real multi-GPU hardware takes, here with a world of one."""
    j = run([sys.executable, "bench.py", "--gpus", "1"] + SMALL, MSPA_BENCH_FORCE_DIST="1")
    check_line(j, 1)
    assert j["config"]["collation_backend"] == "nccl" and "RCCL" in j["config"]["collation"]


def test_bench_measures_the_headline_traffic_in_the_run():
    """Without --no-live-traffic the N = 1 line's `roofline.traffic` comes from two rocprofv3 --pmc child passes of the same
    command on this box (FETCH_SIZE x 2 KiB + WRITE_SIZE x 1 KiB per launch of the K3 kernel), not from the committed file."""
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this box")
    small = [a for a in SMALL if a != "--no-live-traffic"] + ["--no-sweep", "--no-cpu-baseline"]
    j = run([sys.executable, "bench.py", "--gpus", "1"] + small)
    r = j["roofline"]
    assert r["traffic"] is not None and r["traffic_source"].startswith("measured in this run")
    # 96 pairs of 640x480: the two depth frames in, bitset + index table out -- the measured bytes sit near the formula's
    assert 0.5 < r["traffic"] / r["bytes_per_launch"] < 1.5 and r["traffic_frac"] > 0


RCCL_WORLD1 = r"""
import os, sys, torch
sys.path[:0] = [os.path.join(os.getcwd(), "multi-spatialmllm_amd"), os.getcwd()]
from mspa import shard
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = shard.init_distributed(dev)                       # backend None on a GPU -> "nccl" (= RCCL)
assert ctx.backend == "nccl" and ctx.world == 1 and ctx.collective_device == dev
local = torch.arange(21, dtype=torch.float64, device=dev).reshape(7, 3)
full = shard.collate_records(local, ctx)                # counts all_gather + padded all_gather_into_tensor, on the device
assert full.is_cuda and torch.equal(full, local)
empty = shard.collate_records(local[:0], ctx)           # a rank without records
assert tuple(empty.shape) == (0, 3)
g, work = shard.collate_records_async(local.to(torch.int32).contiguous(), ctx)
work.wait()
torch.cuda.synchronize()
assert torch.equal(g, local.to(torch.int32))
ctx.barrier()                                           # dist.barrier(device_ids=[0])
assert ctx.max_over_ranks(2.5) == 2.5
ctx.close()
print("rccl world-1 ok")
"""


def test_collate_records_on_device_tensors_over_rccl():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", RCCL_WORLD1], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0 and "rccl world-1 ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


SCENES_SMALL = ["--workload", "scenes", "--steps", "2", "--warmup", "1", "--scenes-per-gpu", "3", "--scene-points", "8192"]


def check_scene_line(j, n):
    assert j["n_gpus"] == n and j["unit"] == "frame-pairs/s" and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["scenes"] == 3 * n and j["config"]["parallelism"] == f"dp{n}"
    assert j["config"]["rows_per_rank"]["min"] > 10000 and j["roofline"]["kernel_ms"] > 0
    assert j["per_rank_pairs_per_s"]["min"] > 0


def test_bench_scene_workload_one_rank_rccl_and_two_ranks():
    """`--workload scenes` (BASELINE.json configs[2] shape): whole scenes dealt longest-first, K1 + K2 + K4 per rank, the
    pair-table rows collated inside the timed region -- as one rank over RCCL (a world of one: the device-tensor branch) and as
    two ranks sharing the GPU over gloo."""
    j = run([sys.executable, "bench.py", "--gpus", "1"] + SCENES_SMALL)
    check_scene_line(j, 1)
    assert j["config"]["collation"].startswith("none")
    j = run([sys.executable, "bench.py", "--gpus", "1"] + SCENES_SMALL, MSPA_BENCH_FORCE_DIST="1")
    check_scene_line(j, 1)
    assert j["config"]["collation_backend"] == "nccl" and j["rccl_world"] == 1 and j["config"]["bytes_collated_per_step"] > 0
    j = run([sys.executable, "bench.py", "--gpus", "2"] + SCENES_SMALL)
    check_scene_line(j, 2)
    assert j["rccl_world"] == 2 and j["config"]["bytes_collated_per_step"] == 56 * sum(
        f * (f - 1) // 2 for f in [160 + 40 * ((3 * k) % 7) for k in range(6)])


def test_bench_informational_legs_run_small():
    """The from-disk drop-in leg and the K5 leg of the default line, at sizes that take seconds."""
    sys.path[:0] = [ROOT]
    import bench
    import torch
    d = bench.time_dropin_sweep(n_scenes=3, n_frames=6, n_points=4096, num_workers=4)
    assert d["scenes"] == 3 and d["scenes_per_s"] > 0 and d["pair_rows_per_s"] > 0
    busy = d["stage_busy_s"]
    assert all(v is not None and v >= 0 for v in busy.values()) and len(busy) == 6
    k5 = bench.time_track_geometry(torch.device("cuda", 0), T=60, P=32, n_scenes=4, reps=2)
    assert k5["batched"]["frames"] == 240 and k5["one_block"]["frac"] > 0
