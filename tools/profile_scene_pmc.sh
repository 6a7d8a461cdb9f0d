#!/bin/bash
# PMC passes (separate, --kernel-trace only) for the scene-level kernels K1 / K2 / K8 through bench.py's informational legs.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_scene_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --also none --no-dropin-sweep"
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/$name -o p -- $B > $OUT/$name.log 2>&1; }
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
python - <<PY | tee $OUT/summary.md
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for sub in ("sq", "fetch", "write"):
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % sub):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "mspa::" in name and "pair_fast" not in name:
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if sub == "sq" and r["Counter_Name"] == "SQ_WAVES":
                    dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# PMC of the scene-level kernels (320 images x 131072 vertices; 51040 pairs; 8 objects / 34104 object vertices)\n")
print("| kernel | us (under PMC) | waves | VALU instr / wave | VALU busy (ACTIVE_INST_VALU / WAVE_CYCLES) | HBM read MB (FETCH x2) | HBM written MB |")
print("|---|---|---|---|---|---|---|")
for name, c in acc.items():
    m = lambda k: sum(c[k]) / len(c[k]) if c.get(k) else float("nan")
    print(f"| \`{name[-40:]}\` | {sum(dur[name]) / max(1, len(dur[name])):.1f} | {m('SQ_WAVES'):.0f} | {m('SQ_INSTS_VALU') / m('SQ_WAVES'):.0f} | "
          f"{m('SQ_ACTIVE_INST_VALU') / m('SQ_WAVE_CYCLES'):.3f} | {m('FETCH_SIZE') * 2048 / 1e6:.1f} | {m('WRITE_SIZE') * 1024 / 1e6:.1f} |")
PY
for d in stats sq fetch write; do rm -rf $OUT/$d; done     # raw CSVs: too big to bring back
