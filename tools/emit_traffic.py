#!/usr/bin/env python3
"""Write / update profiles/traffic.json from the rocprofv3 PMC passes that tools/profile.sh left under gpurun_out/prof_<tag>/
(fetch/ and write/ counter CSVs), so that the HBM bytes bench.py divides by its own kernel time are never hand-copied.

    python tools/emit_traffic.py KEY=TAG[:committed-summary] ...     e.g.  corr:fast:vc=r03_corr_vc:profiles/r03_k3_corr_vc_pmc.md
    python tools/emit_traffic.py --entry <prof_dir>                  (on the GPU box, by tools/profile.sh: one profile's two counters)
    python tools/emit_traffic.py --valu KEY=profiles/<summary>.md ...  add `valu_insts_per_launch` / `waves_per_launch` to entry KEY
                                                                     from a committed PMC summary's SQ_INSTS_VALU / SQ_WAVES rows
FETCH_SIZE is doubled (it reports half of the streamed bytes on gfx950: profiles/r01_counter_calibration.md,
MI355X_MICROARCH.md HBM section); WRITE_SIZE is taken as reported.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATTERN = os.environ.get("MSPA_PROF_PATTERN", "pair_")


def counter(prof_dir, sub, name):
    vals, kernels = [], collections.Counter()
    for f in glob.glob(os.path.join(prof_dir, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if PATTERN in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
                kernels[r["Kernel_Name"].split("(")[0].replace("void ", "")] += 1
    if not vals:
        raise SystemExit(f"{prof_dir}/{sub}: no {name} rows for kernels matching {PATTERN!r}")
    return sum(vals) / len(vals), kernels.most_common(1)[0][0]


def entry(prof_dir):
    """The two counters of one profile directory as a small JSON object (written on the GPU box by tools/profile.sh, where
    the raw CSVs are; they are too big to bring back)."""
    fetch, kern = counter(prof_dir, "fetch", "FETCH_SIZE")
    write, _ = counter(prof_dir, "write", "WRITE_SIZE")
    out = {"kernel": kern, "pairs": 1000, "fetch_kib_reported": round(fetch), "write_kib_reported": round(write),
           "hbm_bytes_per_launch": int(fetch * 2048 + write * 1024)}
    try:                                                     # the instruction-issue roofline's numerator (bench.py roofline_valu)
        out["valu_insts_per_launch"] = round(counter(prof_dir, "sq", "SQ_INSTS_VALU")[0])
        out["waves_per_launch"] = round(counter(prof_dir, "sq", "SQ_WAVES")[0])
    except SystemExit:
        pass
    return out


def valu_from_summary(path):
    """(SQ_INSTS_VALU, SQ_WAVES) means from a committed summary table (either of the two layouts the tools write)."""
    got = {}
    for line in open(path):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        for name in ("SQ_INSTS_VALU", "SQ_WAVES"):
            if name in cells:
                got[name] = float(cells[cells.index(name) + 1])
    return got.get("SQ_INSTS_VALU"), got.get("SQ_WAVES")


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--entry":
        print(json.dumps(entry(sys.argv[2])))
        return
    path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    if len(sys.argv) > 2 and sys.argv[1] == "--valu":
        for spec in sys.argv[2:]:
            key, src = spec.split("=", 1)
            valu, waves = valu_from_summary(os.path.join(ROOT, src))
            if valu is None:
                raise SystemExit(f"{src}: no SQ_INSTS_VALU row")
            table.setdefault(key, {}).update(valu_insts_per_launch=round(valu), waves_per_launch=round(waves or 0), valu_source=src)
            print(key, table[key])
        with open(path, "w") as f:
            json.dump(table, f, indent=1)
            f.write("\n")
        return
    for spec in sys.argv[1:]:
        key, rest = spec.split("=", 1)
        tag, _, src = rest.partition(":")
        d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
        small = os.path.join(d, "traffic_entry.json")
        table[key] = dict(json.load(open(small)) if os.path.exists(small) else entry(d),
                          source=src or f"gpurun_out/prof_{tag}/summary.md")
        print(key, table[key])
    table["_note"] = ("HBM bytes per launch of the benchmarked kernels, from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                      "passes of bench.py's own command (tools/profile.sh), written by tools/emit_traffic.py. FETCH_SIZE is doubled (it "
                      "reports half of the streamed bytes on gfx950: profiles/r01_counter_calibration.md, MI355X_MICROARCH.md HBM "
                      "section); WRITE_SIZE is exact. bench.py reads this file: the counters cannot be read from inside the run.")
    with open(path, "w") as f:
        json.dump(table, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
