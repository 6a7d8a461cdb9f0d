"""Benchmark workloads (mspa/workload.py): the overlap-binned pair sampler is the reference's sample_dataframe rule."""
import os

import numpy as np
import pytest

from mspa import workload

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(n_frames, seed):
    rng = np.random.default_rng(seed)
    n = n_frames * (n_frames - 1) // 2
    ov = rng.uniform(0, 40, n)
    ov[rng.random(n) < 0.1] = 0.0
    ov[rng.random(n) < 0.02] = np.nan
    return ov


def test_vc_sample_matches_sample_dataframe_quota_rule():
    """Same bins, same quotas and carry-over as mspa.sampling.sample_dataframe (itself pinned row for row to the
    reference, tests/test_sampling_vs_reference.py): per-bin counts must agree for any table."""
    import pandas as pd
    from mspa import sampling
    for seed, n in ((1, 600), (2, 3000)):
        ov = np.concatenate([_table(64, seed)] * 2)
        idx = workload.binned_quota_sample(ov, n, np.random.default_rng(0))
        assert len(np.unique(idx)) == len(idx)
        df = pd.DataFrame({"overlap": ov[np.isfinite(ov)]})
        ref = sampling.sample_dataframe(df, n, 0, overlap_min=6, overlap_max=35, interval=1)
        edges = np.arange(6, 36)
        got_hist = np.histogram(ov[idx], bins=edges)[0]
        ref_hist = np.histogram(ref["overlap"].to_numpy(), bins=edges)[0]
        # np.histogram bins are [a, b) and pd.cut's (a, b]: identical for continuous draws
        assert np.array_equal(got_hist, ref_hist), (got_hist, ref_hist)
        assert ov[idx].min() >= 6 and ov[idx].max() <= 35


def test_select_pairs_kinds_and_orders():
    ov = _table(48, 3)
    for kind, check in (("vc", lambda o: (o >= 6) & (o <= 35)), ("low", lambda o: ~(o >= 6)), ("high", lambda o: o >= 25)):
        pairs, info = workload.select_pairs(ov, 48, 1000, kind, seed=5)
        assert pairs.shape == (1000, 2) and pairs.dtype == np.int32 and (pairs[:, 0] != pairs[:, 1]).all()
        i, j = np.minimum(pairs[:, 0], pairs[:, 1]), np.maximum(pairs[:, 0], pairs[:, 1])
        lut = np.full((48, 48), np.nan)
        lut[np.triu_indices(48, 1)] = ov
        assert check(lut[i, j]).all(), kind
        swapped = (pairs[:, 0] > pairs[:, 1]).mean()
        assert 0.3 < swapped < 0.7, "both frame orders occur (VC_C:280)"
        assert info["candidates"] == 48 * 47
    again, _ = workload.select_pairs(ov, 48, 1000, "vc", seed=5)
    assert np.array_equal(again, workload.select_pairs(ov, 48, 1000, "vc", seed=5)[0])
    with pytest.raises(ValueError):
        workload.select_pairs(np.zeros(6), 4, 10, "vc", seed=0)
    # a scene without pairs of the named kind falls back to its extreme tenth (tiny test scenes)
    pairs, info = workload.select_pairs(np.full(28, 12.0), 8, 10, "low", seed=0)
    assert pairs.shape == (10, 2) and "lowest tenth" in info["rule"]
    pairs, info = workload.select_pairs(np.linspace(1, 20, 28), 8, 10, "high", seed=0)
    assert "highest tenth" in info["rule"] and info["overlap_pct_min"] > 18


def test_bench_refuses_to_run_without_gpu_and_self_launches():
    """`python bench.py --gpus 2` re-executes itself under torch.distributed.run (the driver's command shape works with
    and without a launcher); on a box without a GPU every rank stops with the no-CPU-fallback message."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test of the same entry point")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("no CPU fallback") >= 1 and "torch.distributed" in out.stderr
