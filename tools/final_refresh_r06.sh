#!/bin/bash
# Round 6's evidence in one gpurun call: the default bench line, rocprofv3 --kernel-trace --stats of the headline command and of the
# on-device decode gate, the decode kernel's instruction counters (separate --pmc passes, kernel trace only).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/final_r06
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( time timeout 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err ) 2> $OUT/bench_n1.time
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k3_stats -o p -- python $ROOT/bench.py --steps 20 --warmup 5 --no-scene-legs --no-sweep --also none --no-cpu-baseline --no-live-traffic > $OUT/k3_stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/decode_stats -o p -- python $ROOT/tools/device_ingest_bench.py --streams 4096 --reps 3 > $OUT/decode_stats.log 2>&1
cd $ROOT
python - <<PY > $OUT/kernel_stats.md
import csv, glob
for tag, title in (("k3_stats", "bench.py --steps 20 --warmup 5 (headline K3 launch: 1 000 pairs of 640x480)"), ("decode_stats", "tools/device_ingest_bench.py --streams 4096 (depth decode on the device)")):
    print(f"## rocprofv3 --kernel-trace --stats -- {title}\n")
    print("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|")
    for f in glob.glob("$OUT/" + tag + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mspa::" in r["Name"]:
                print(f"| \`{r['Name'].split('(')[0][-80:]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
    print()
PY
rm -rf $OUT/k3_stats $OUT/decode_stats
bash tools/pmc_inflate.sh r06_inflate_v6 4096 > $OUT/pmc_inflate.log 2>&1
cp gpurun_out/pmc_r06_inflate_v6/summary.md $OUT/pmc_inflate_v6.md
tail -3 $OUT/bench_n1.time
