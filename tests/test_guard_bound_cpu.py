"""The fast kernels' guard band as a bound, checked on the CPU (no GPU): tools/guard_bound_emulation.py restates the kernels'
per-tile bound (mspa_common.h guard_from_bounds, fed by the frame records' MSPA_MAT_BOUNDS slot) in NumPy and compares, lane by
lane, the composed-matrix evaluation with oracle/np_oracle on adversarial pairs, on cameras within 1e-9 .. 1e-3 m of a
back-projected frame-1 point and on scenes 1e4 / 1e6 m from the origin: |q_fast - q_ref| <= B everywhere, no decision of the
guarded fast path differs from the reference's -- and round 3's constant band does flip decisions there (VERDICT round 3,
item 1).  The quick mode runs in seconds."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_guard_bound_holds_and_constant_band_fails():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "guard_bound_emulation.py"), "--quick"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    rows = [[c.strip() for c in l.strip().strip("|").split("|")] for l in r.stdout.splitlines() if l.startswith("| ") and "case" not in l]
    assert len(rows) >= 10
    over_bound = sum(int(x[4]) for x in rows)
    mism_bound = sum(int(x[8]) for x in rows)
    mism_const = sum(int(x[9]) for x in rows)
    worst = max(float(x[3]) for x in rows)
    assert over_bound == 0 and mism_bound == 0
    assert worst < 0.25                      # C = 256 against a counted 91 roundings: the measured worst case is far inside
    assert mism_const > 0                    # the regime is exercised: round 3's constants do flip decisions in it
