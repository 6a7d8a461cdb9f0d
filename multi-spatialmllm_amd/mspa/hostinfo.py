"""How many host CPUs this process may actually use, and whether the kernel has been throttling it.

``os.cpu_count()`` is the machine (256 on the MI355X boxes); a container's CFS bandwidth quota (cgroup ``cpu.max``) can be far
below it (16 CPUs on those boxes).  Running more busy threads than the quota does not just fail to help: once the group has
burnt its quota for the 100 ms period EVERY thread of the group is frozen until the next period -- decode threads, the
staging thread, the thread that feeds the GPU -- which is the 2 x pass-to-pass alternation round 5's from-disk leg showed
(`nr_throttled` rises in the slow passes).  The ingest thread counts are sized from ``effective_cpus()``.
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, Optional


def _cgroup_quota() -> Optional[float]:
    """CPUs' worth of CFS quota of this process's cgroup (v2 ``cpu.max``, v1 ``cpu.cfs_quota_us``), None if unlimited."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return None


def effective_cpus(per_rank: bool = True) -> int:
    """min(affinity mask, cgroup quota), at least 1.  ``per_rank``: divided by ``LOCAL_WORLD_SIZE`` (torch.distributed.run sets
    it) -- the ranks of a node share its CPUs, and every rank sizes its own threads.  ``MSPA_HOST_CPUS`` overrides (taken as
    this process's own share)."""
    env = os.environ.get("MSPA_HOST_CPUS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = _cgroup_quota()
    if q is not None:
        n = min(n, max(1, int(q)))
    if per_rank:
        try:
            n //= max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        except ValueError:
            pass
    return max(1, n)


def throttle_stats() -> Dict[str, int]:
    """``nr_periods / nr_throttled / throttled_usec`` of this cgroup's ``cpu.stat`` ({} where there is none): the difference
    over an interval says how often, and for how long, the group's threads were frozen by the quota."""
    out: Dict[str, int] = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time"):
                    out["throttled_usec" if k == "throttled_time" else k] = int(v) // (1000 if k == "throttled_time" else 1)
            if out:
                return out
        except (OSError, ValueError):
            continue
    return out


def describe() -> Dict[str, object]:
    return {"os_cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
            "cgroup_cpu_quota": _cgroup_quota(), "effective_cpus": effective_cpus(per_rank=False),
            "effective_cpus_per_rank": effective_cpus()}


@contextlib.contextmanager
def quiet_collector():
    """For the duration of a sweep: what the process holds when it begins is taken out of the cyclic collector's sight
    (``gc.freeze()``) and a young-generation pass needs 50 000 net allocations instead of 700.

    A sweep runs a dozen Python threads (the sweep thread, the staging thread, loaders, encoders, the writer, the exchange thread) that
    all stop for every pass of the collector.  With the split's scene infos in memory -- a dict of dicts of arrays per image: about a
    million container objects for 192 scenes of 320 images, several millions for ScanNet -- one full collection in the middle of a
    pass took 70-150 ms and the ~250 young-generation passes another 15-20: 7-13 % of a 192-scene pass (tools/sweep_timeline.py
    --series, profiles/r06_sweep_timeline.md: 123-138 -> 142-146 scenes/s).  Nothing the sweep allocates is cyclic garbage worth
    looking for; at the end the thresholds are restored and the frozen objects handed back (unless the application had frozen
    objects of its own before: then they stay frozen).  ``MSPA_GC_FREEZE=0`` leaves the collector alone."""
    import gc
    if os.environ.get("MSPA_GC_FREEZE", "1") == "0" or not gc.isenabled():
        yield
        return
    # collector settings are the process's: the outermost user (of any thread) changes them, the last one out puts them back
    with _QUIET_LOCK:
        _QUIET["depth"] += 1
        if _QUIET["depth"] == 1:
            _QUIET["was"], _QUIET["frozen_before"] = gc.get_threshold(), gc.get_freeze_count()
            gc.freeze()
            was = _QUIET["was"]
            gc.set_threshold(max(was[0], 50000), was[1], was[2])
    try:
        yield
    finally:
        with _QUIET_LOCK:
            _QUIET["depth"] -= 1
            if _QUIET["depth"] == 0:
                gc.set_threshold(*_QUIET["was"])
                if _QUIET["frozen_before"] == 0:
                    gc.unfreeze()


_QUIET = {"depth": 0, "was": None, "frozen_before": 0}
_QUIET_LOCK = __import__("threading").Lock()


def quietly(fn):
    """Decorator: ``fn`` runs inside ``quiet_collector`` (the dataset builders' record loops: a list of a few hundred thousand dict
    records that all stay alive is the collector's worst case -- tools/heads_bench.py: camera movement 63 k -> 116 k records/s,
    coordinate correspondences 27 k -> 77 k, object movement 17.6 k -> 32.9 k with the young generation's threshold raised)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with quiet_collector():
            return fn(*a, **k)
    return wrapped
