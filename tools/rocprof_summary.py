#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile.sh into a short markdown summary
(the file committed under profiles/).  Usage: rocprof_summary.py <prof_dir> [kernel substring]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "mspa::"
    print(f"# rocprofv3 summary of `{os.path.basename(d)}` (kernels matching `{pat}`)\n")
    for f in glob.glob(os.path.join(d, "stats", "*kernel_stats.csv")):
        print("## --kernel-trace --stats\n")
        print("| kernel | calls | avg us | min us | max us | % of GPU time |")
        print("|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            name = r["Name"]
            short = name.split("(")[0][-70:]
            if pat in name or float(r["Percentage"]) > 5:
                print(f"| `{short}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                      f"{float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
        print()
    print("## PMC passes (mean per dispatch of the matching kernels, warm-up dispatches included)\n")
    print("| pass | counter | mean per dispatch | dispatches |")
    print("|---|---|---|---|")
    allc = {}
    for sub in ("sq", "sq2", "fetch", "write", "tcc"):
        for f in glob.glob(os.path.join(d, sub, "*counter_collection.csv")):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if pat in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                allc[k] = sum(v) / len(v)
                print(f"| {sub} | {k} | {sum(v)/len(v):.4g} | {len(v)} |")
    print()
    g = allc.get
    if g("SQ_WAVE_CYCLES") and g("SQ_WAVES"):
        print("## derived\n")
        print(f"- waves per dispatch: {g('SQ_WAVES'):.0f}; wave-cycles (quad-cycle units) per wave: "
              f"{g('SQ_WAVE_CYCLES')/g('SQ_WAVES'):.0f}")
        if g("SQ_INSTS_VALU"):
            print(f"- VALU instructions per wave: {g('SQ_INSTS_VALU')/g('SQ_WAVES'):.0f}")
        for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if g(k):
                print(f"- {k} / SQ_WAVE_CYCLES = {g(k)/g('SQ_WAVE_CYCLES'):.3f}")
    if g("FETCH_SIZE") is not None:
        print(f"- FETCH_SIZE (KiB as reported) = {g('FETCH_SIZE'):.0f}  -> x1024 x2 (gfx950 half-count correction, "
              f"MI355X_MICROARCH.md HBM section) = {g('FETCH_SIZE')*2048/1e6:.1f} MB read per dispatch")
    if g("WRITE_SIZE") is not None:
        print(f"- WRITE_SIZE (KiB as reported) = {g('WRITE_SIZE'):.0f}  -> x1024 = {g('WRITE_SIZE')*1024/1e6:.1f} MB written "
              f"per dispatch (x1.000 on fill / copy kernels of known size: profiles/r01_counter_calibration.md)")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        print(f"- L2 hit rate = {g('TCC_HIT_sum')/(g('TCC_HIT_sum')+g('TCC_MISS_sum')):.3f}")


if __name__ == "__main__":
    main()
