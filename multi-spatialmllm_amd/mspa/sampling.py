"""Overlap-binned sampling of the pair table (host glue of the camera-movement and
visual-correspondence heads; reference: camera_movement_engine_train_val.py:29-151, repeated in the
visual-correspondence scripts).

Behaviour reproduced, including its use of the *global* NumPy generator through ``DataFrame.sample``
(so that ``np.random.seed`` makes runs repeatable exactly as upstream):
  1. rows with overlap == 0 are sampled on their own (at most ``non_overlap_samples``);
  2. the remaining rows are cut into ``interval``-wide overlap bins over [overlap_min, overlap_max]
     (lowest edge included, rows outside dropped);
  3. ``all_overlap_samples`` is split evenly over ALL bins in bin order, empty ones included (first
     ``remainder`` bins get one more), then bins are visited from the smallest to the largest and a bin
     that cannot fill its quota passes the shortfall on to the next one;
  4. result = binned samples followed by the zero-overlap samples, index reset.
"""
from __future__ import annotations

import numpy as np
import pandas as pd


def sample_dataframe(df: pd.DataFrame, all_overlap_samples: int, non_overlap_samples: int, overlap_min=0,
                     overlap_max=100, interval=1) -> pd.DataFrame:
    zero = df[df["overlap"] == 0].copy()
    zero_part = zero if len(zero) <= non_overlap_samples else zero.sample(n=non_overlap_samples)

    rest = df[df["overlap"] != 0].copy()
    edges = np.arange(overlap_min, overlap_max + interval, interval)
    rest["overlap_group"] = pd.cut(rest["overlap"], bins=edges, include_lowest=True)
    rest = rest.dropna(subset=["overlap_group"])
    groups = [g for _, g in rest.groupby("overlap_group", observed=False)]   # one entry per bin, empty ones too
    if not groups:
        return zero_part.drop(columns=["overlap_group"], errors="ignore")

    base, extra = divmod(all_overlap_samples, len(groups))
    quota = [base + (1 if k < extra else 0) for k in range(len(groups))]
    order = sorted(range(len(groups)), key=lambda k: len(groups[k]))   # stable: ties keep bin order
    picked, carry = [], 0
    for k in order:
        want = quota[k] + carry
        print(f"[sample_dataframe] current_quota v.s. group_df: {want} v.s. {len(groups[k])}")
        if len(groups[k]) <= want:
            picked.append(groups[k])
            carry = want - len(groups[k])
        else:
            picked.append(groups[k].sample(n=want))
            carry = 0
    if carry > 0:
        print(f"[sample_dataframe] Warning: bins not enough to reach {all_overlap_samples}; leftover {carry}")
    binned = pd.concat(picked, ignore_index=True) if picked else pd.DataFrame()
    out = pd.concat([binned, zero_part], ignore_index=True)
    return out.drop(columns=["overlap_group"], errors="ignore")
