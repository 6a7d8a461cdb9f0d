#!/bin/bash
# rocprofv3 kernel-trace stats of the scene-level kernels (K1, popcount, K2, K4, K7, K8) through bench.py's informational legs.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_scene
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --also none --no-dropin-sweep > $OUT/stats.log 2>&1
python - <<PY | tee $OUT/summary.md
import csv, glob
print("# rocprofv3 --kernel-trace --stats: scene-level kernels (bench.py informational legs, 320 images x 131072 vertices; 300 x 256 tracks)\n")
print("| kernel | calls | avg us | min us | max us |")
print("|---|---|---|---|---|")
for f in glob.glob("$OUT/stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "mspa::" in r["Name"]:
            print(f"| \`{r['Name'].split('(')[0][-60:]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
PY
for d in stats sq fetch write; do rm -rf $OUT/$d; done     # raw CSVs: too big to bring back
