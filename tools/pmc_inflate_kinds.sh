#!/bin/bash
# Instruction counts of the inflate kernel PER KIND of stream (tools/device_inflate_kinds.py under rocprofv3 --pmc): one row per
# dispatch, counters divided by the dispatch's waves.  usage: tools/pmc_inflate_kinds.sh <tag> [streams]
TAG=$1; N=${2:-1024}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmck_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -o p -- python $ROOT/tools/device_inflate_kinds.py --streams $N > $OUT/kinds.log 2>&1
python - <<PY | tee $OUT/summary.md
import csv, glob, collections
rows = collections.OrderedDict()
for f in glob.glob("$OUT/a/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "inflate_kernel" not in r["Kernel_Name"]:
            continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("| dispatch | waves | SALU/wave | BRANCH/wave | VALU/wave | LDS/wave | wave cycles (x4?) | wait_any | wait_inst |")
print("|---|---|---|---|---|---|---|---|---|")
for d, c in sorted(rows.items()):
    w = c.get("SQ_WAVES", 1) or 1
    print(f"| {d} | {w:.0f} | {c.get('SQ_INSTS_SALU',0)/w:.4g} | {c.get('SQ_INSTS_BRANCH',0)/w:.4g} | {c.get('SQ_INSTS_VALU',0)/w:.4g} | {c.get('SQ_INSTS_LDS',0)/w:.4g} | {c.get('SQ_WAVE_CYCLES',0)/w:.4g} | {c.get('SQ_WAIT_ANY',0)/w:.4g} | {c.get('SQ_WAIT_INST_ANY',0)/w:.4g} |")
PY
grep -v "^$" $OUT/kinds.log | tail -8
rm -rf $OUT/a
