"""mspa.sampling.sample_dataframe against the reference's (same global NumPy seed, same rows out)."""
import numpy as np
import pandas as pd
import pytest

from mspa.sampling import sample_dataframe
from oracle import ref_harness as RH

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not RH.reference_available(), reason="/root/reference not mounted")]


@pytest.mark.parametrize("seed,n,total,zeros,lo,hi,step", [(0, 5000, 600, 40, 6, 35, 1), (1, 800, 900, 0, 0, 100, 5),
                                                           (2, 300, 50, 10, 6, 35, 1), (3, 50, 10, 5, 40, 60, 2)])
def test_sample_dataframe(seed, n, total, zeros, lo, hi, step, capsys):
    ref = RH.import_reference()
    rng = np.random.default_rng(seed)
    ov = rng.gamma(2.0, 8.0, n)
    ov[rng.random(n) < 0.2] = 0.0
    ov[rng.random(n) < 0.01] = np.nan
    df = pd.DataFrame({"scene_id": "s", "image_id1": np.arange(n).astype(str), "image_id2": "x", "overlap": ov,
                       "distance": rng.random(n), "yaw": rng.random(n), "pitch": rng.random(n)})
    np.random.seed(123)
    want = ref.CME.sample_dataframe(df, total, zeros, lo, hi, step)
    np.random.seed(123)
    got = sample_dataframe(df, total, zeros, lo, hi, step)
    pd.testing.assert_frame_equal(got.reset_index(drop=True), want.reset_index(drop=True))
