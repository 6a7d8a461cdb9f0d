#!/usr/bin/env python3
"""Calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on this GPU against kernels whose HBM traffic is known.

    tools/calibrate_hbm_counters.py workload          # the kernels: 1 GiB fill, 1 GiB copy, 1 GiB sum (run under rocprofv3)
    tools/calibrate_hbm_counters.py report <dir>      # reported KiB per kernel vs the bytes it must move

Driven by tools/calibrate.sh (one --pmc pass per counter, --kernel-trace only)."""
import collections
import csv
import glob
import os
import sys

GIB = 1 << 30


def workload():
    import torch
    x = torch.empty(GIB // 4, dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    for _ in range(3):
        x.fill_(7)                 # writes 1 GiB
        y.copy_(x)                 # reads 1 GiB, writes 1 GiB
        s = x.sum()                # reads 1 GiB
    torch.cuda.synchronize()
    print(int(s))


def report(d):
    want = {"fill": (0, GIB), "copy": (GIB, GIB), "sum": (GIB, 0)}
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("fetch", "write"):
        for f in glob.glob(os.path.join(d, sub, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                name, grid = r["Kernel_Name"], int(r["Grid_Size"])
                kind = ("fill" if "FillFunctor" in name else "copy" if "copy" in name.lower() else
                        "sum" if "reduce" in name.lower() and grid > 10000 else None)
                if kind:
                    rows[kind][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("| kernel | counter | reported KiB (mean) | expected KiB | reported / expected |")
    print("|---|---|---|---|---|")
    for kind, (rd, wr) in want.items():
        for counter, expect in (("FETCH_SIZE", rd), ("WRITE_SIZE", wr)):
            v = rows[kind].get(counter)
            if not v:
                continue
            mean = sum(v) / len(v)
            ratio = f"{mean * 1024 / expect:.3f}" if expect else "-"
            print(f"| {kind} (1 GiB) | {counter} | {mean:.0f} | {expect // 1024} | {ratio} |")


if __name__ == "__main__":
    if sys.argv[1] == "workload":
        workload()
    else:
        report(sys.argv[2])
