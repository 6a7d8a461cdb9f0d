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
         "--no-scene-legs", "--also", "none", "--cpu-seconds", "1"]


def run(cmd):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
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
             "--master-port", "29611", "bench.py", "--gpus", "2"] + SMALL)
    check_line(j, 2)
