"""Frame-pair workloads of the benchmark: which pairs of a scene's frames go through K3.

The cost of the fast K3 kernels depends on how much of frame 1 lands inside frame 2 (tile culling, group
early-out), so "1 000 pairs" is not a workload until the pairs are named.  The pairs are therefore drawn the way
the reference draws them: from the scene's all-pairs overlap table (``calculate_camera_overlap``, CFR:102-137 --
here K1 + K2, bit-equal to it) with the overlap-binned equal-quota sampler the visual-correspondence engine
uses (``sample_dataframe``, VC_C:29-151 = CME:29-151, called with overlap_min = 6, overlap_max = 35, interval = 1
at VC_C:485-505), and both frame orders occur (the 50 % swap of VC_C:280).

  vc    equal quotas over the 1 %-wide overlap bins [6, 7], (7, 8] ... (34, 35]; a bin that cannot fill its quota
        passes the shortfall to the next larger bin, smallest bin first -- exactly the reference's rule.  HEADLINE.
  low   pairs the reference's sampler never takes for correspondence: overlap < 6 % (zero and NaN included).
  high  near-identical views: the upper overlap bins, >= ``high_min`` % (default 25; with independent depth noise
        on the two frames two identical views reach ~30 %).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

VC_OVERLAP_MIN, VC_OVERLAP_MAX, VC_INTERVAL = 6, 35, 1           # VC_C:485-487


def ordered_candidates(n_frames: int) -> np.ndarray:
    """All ordered frame pairs (a, b), a != b: [n*(n-1), 2] int32, (i, j) for i < j first, then the swapped copies."""
    i, j = np.triu_indices(n_frames, k=1)
    fwd = np.stack([i, j], 1)
    return np.concatenate([fwd, fwd[:, ::-1]], 0).astype(np.int32)


def binned_quota_sample(overlap: np.ndarray, n: int, rng: np.random.Generator, lo=VC_OVERLAP_MIN, hi=VC_OVERLAP_MAX,
                        interval=VC_INTERVAL) -> np.ndarray:
    """Indices into ``overlap`` chosen like ``sample_dataframe``: pd.cut bins (lowest edge included), ``n`` split
    evenly over ALL bins (the first ``n % bins`` get one more), bins visited from the smallest population to the
    largest, a short bin is taken whole and its shortfall carried to the next one."""
    edges = np.arange(lo, hi + interval, interval, dtype=np.float64)
    nb = len(edges) - 1
    ov = np.asarray(overlap, dtype=np.float64)
    # pd.cut(right=True, include_lowest=True): (e_k, e_k+1], the first bin also holds e_0 itself
    k = np.searchsorted(edges, ov, side="left") - 1
    k = np.where(ov == edges[0], 0, k)
    ok = np.isfinite(ov) & (ov >= edges[0]) & (ov <= edges[-1]) & (ov != 0)
    members = [np.nonzero(ok & (k == b))[0] for b in range(nb)]
    base, extra = divmod(n, nb)
    quota = [base + (1 if b < extra else 0) for b in range(nb)]
    order = sorted(range(nb), key=lambda b: len(members[b]))          # stable: ties keep bin order
    picked, carry = [], 0
    for b in order:
        want = quota[b] + carry
        if len(members[b]) <= want:
            picked.append(members[b])
            carry = want - len(members[b])
        else:
            picked.append(rng.choice(members[b], size=want, replace=False))
            carry = 0
    return np.concatenate(picked) if picked else np.zeros(0, dtype=np.int64)


def select_pairs(overlap_ij: np.ndarray, n_frames: int, n_pairs: int, kind: str, seed: int,
                 high_min: float = 25.0) -> Tuple[np.ndarray, Dict]:
    """``overlap_ij``: the scene's overlap column for (i, j), i < j, in ``engine.all_pairs`` order.  Returns
    ([n_pairs, 2] int32 frame indices, a description of what was drawn)."""
    cand = ordered_candidates(n_frames)
    ov = np.concatenate([overlap_ij, overlap_ij]).astype(np.float64)
    rng = np.random.default_rng(seed)
    if kind == "vc":
        idx = binned_quota_sample(ov, n_pairs, rng)
        rule = f"equal quotas over the overlap bins {VC_OVERLAP_MIN}..{VC_OVERLAP_MAX} % step {VC_INTERVAL} (VC_C:485-505)"
    elif kind == "low":
        pool = np.nonzero(~(ov >= VC_OVERLAP_MIN))[0]                # NaN (empty union) lands here too
        idx = rng.choice(pool, size=n_pairs, replace=len(pool) < n_pairs) if len(pool) else pool
        rule = f"uniform over pairs with overlap < {VC_OVERLAP_MIN} % (zero / NaN included)"
    elif kind == "high":
        pool = np.nonzero(ov >= high_min)[0]
        idx = rng.choice(pool, size=n_pairs, replace=len(pool) < n_pairs) if len(pool) else pool
        rule = f"uniform over pairs with overlap >= {high_min:g} % (near-identical views)"
    else:
        raise ValueError(f"unknown workload {kind!r}")
    if len(idx) == 0:
        # a tiny scene (tests) may have no pair in the named range: fall back to the nearest tenth of the candidates by overlap
        finite = np.where(np.isfinite(ov), ov, 0.0)
        order = np.argsort(finite, kind="stable")
        n10 = max(1, len(order) // 10)
        if kind == "low":
            pool = order[:n10]
        elif kind == "high":
            pool = order[-n10:]
        else:
            raise ValueError(f"workload {kind!r}: the scene has no pair with overlap in {VC_OVERLAP_MIN}..{VC_OVERLAP_MAX} %")
        idx = rng.choice(pool, size=n_pairs, replace=len(pool) < n_pairs)
        rule += f" -- none in this scene: the {'lowest' if kind == 'low' else 'highest'} tenth of its pairs instead"
    if len(idx) < n_pairs:      # every bin exhausted (tiny scenes): top up with repeats, frames differ per replica anyway
        idx = np.concatenate([idx, rng.choice(idx, size=n_pairs - len(idx), replace=True)])
    idx = rng.permutation(idx)
    sel = ov[idx]
    fin = sel[np.isfinite(sel)]
    info = {"rule": rule, "candidates": int(len(cand)), "distinct_pairs_drawn": int(len(np.unique(idx))),
            "overlap_pct_min": round(float(fin.min()), 3) if len(fin) else None,
            "overlap_pct_mean": round(float(fin.mean()), 3) if len(fin) else None,
            "overlap_pct_max": round(float(fin.max()), 3) if len(fin) else None}
    return cand[idx], info
