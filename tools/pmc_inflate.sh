#!/bin/bash
# Instruction-mix / stall PMC passes for the on-device inflate kernel (run on the GPU box through gpurun).  Separate rocprofv3
# runs, --kernel-trace only.  usage: tools/pmc_inflate.sh <tag> <streams>
TAG=$1; N=${2:-640}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/device_ingest_bench.py --streams $N --reps 2"
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $CMD > $OUT/$name.log 2>&1; }
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU
run sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
python - <<PY | tee $OUT/summary.md
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "dinf" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# PMC of the on-device decode kernels ($TAG, $N streams): mean per dispatch\n")
for k in sorted(acc):
    print(f"## {k}\n\n| counter | mean |\n|---|---|")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"| {c} | {sum(v)/len(v):.5g} |")
    print()
PY
for d in sq sq2 sq3; do rm -rf $OUT/$d; done
