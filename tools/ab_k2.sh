#!/bin/bash
# rocprofv3 average duration of the K2 kernels for every tools/ab/libmspa_*.so and the in-tree library.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
cd /tmp
for lib in $ROOT/tools/ab/libmspa_*.so $ROOT/multi-spatialmllm_amd/libmspa.so; do
  rm -rf /tmp/k2prof
  MSPA_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2prof -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-live-traffic --no-sweep --also none > /dev/null 2>&1
  python - <<PY
import csv, glob
out = []
for f in glob.glob("/tmp/k2prof/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "overlap" in r["Name"]:
            out.append("%s %.1f" % (r["Name"].split("(")[0].split("::")[-1][:24], float(r["AverageNs"]) / 1e3))
print("%-26s" % "$lib".split("/")[-1], " | ".join(sorted(out)))
PY
done
