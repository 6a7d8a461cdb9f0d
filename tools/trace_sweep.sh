#!/bin/bash
# Kernel + memory-copy traces (timestamps per dispatch / copy, no counters) of one sweep process in two modes, for offline analysis.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $ROOT/gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-0 1}; do
  MSPA_PREPARE_ON_LOADER=${mode%%_*} timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/tr_$mode -o p -- python $ROOT/tools/sweep_timeline.py --scenes 96 --brief > $ROOT/gpurun_out/trace/run_$mode.txt 2>&1
  for f in $(find /tmp/tr_$mode -name "*kernel_trace.csv"); do python - "$f" $ROOT/gpurun_out/trace/kernels_$mode.csv <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["name", "queue", "stream", "start", "end", "grid"])
    for r in rows:
        n = r["Kernel_Name"]
        short = "inflate" if "inflate_kernel" in n else "hard" if "unfilter_hard" in n else "unfilter" if "png_unfilter" in n else "adler" if "adler32" in n else "K1" if "vertex_visibility" in n else "copyBuffer" if "copyBuffer" in n else "other"
        w.writerow([short, r.get("Queue_Id"), r.get("Stream_Id", ""), r["Start_Timestamp"], r["End_Timestamp"], r.get("Grid_Size_X", r.get("Grid_Size", ""))])
P
  done
  for f in $(find /tmp/tr_$mode -name "*memory_copy_trace.csv"); do cp $f $ROOT/gpurun_out/trace/copies_$mode.csv; done
  tail -1 $ROOT/gpurun_out/trace/run_$mode.txt | cut -c1-300
done
ls -la $ROOT/gpurun_out/trace
