#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the GPU box: separate --pmc passes, --kernel-trace only.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/calib
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc ${c#*:} --output-format csv -d $OUT/${c%%:*} -o p -- python $ROOT/tools/calibrate_hbm_counters.py workload > $OUT/${c%%:*}.log 2>&1
done
python $ROOT/tools/calibrate_hbm_counters.py report $OUT | tee $OUT/calibration.md
# load shapes of the K3 kernel (tools/ubench/hbm_patterns.hip; built here if the binary did not travel)
[ -x $ROOT/tools/ubench/hbm_patterns ] || hipcc --offload-arch=gfx950 -O2 -o $ROOT/tools/ubench/hbm_patterns $ROOT/tools/ubench/hbm_patterns.hip
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/patterns -o p -- $ROOT/tools/ubench/hbm_patterns > $OUT/patterns.log 2>&1
python - <<PY | tee -a $OUT/calibration.md
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/patterns/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
print()
print("| load shape | FETCH_SIZE reported KiB (mean) | bytes the kernel reads (KiB) | reported / actual |")
print("|---|---|---|---|")
for k, v in acc.items():
    m = sum(v) / len(v)
    print(f"| {k} | {m:.0f} | {1<<20} | {m / (1<<20):.3f} |")
PY
