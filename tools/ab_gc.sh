#!/bin/bash
# The sweeps with and without sweep.quiet_collector (gc.freeze + a young-generation threshold of 50 000) inside one box: 96 scenes x 7 passes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/gc
run() {
  echo -n "MSPA_GC_FREEZE=$1: "
  MSPA_GC_FREEZE=$1 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes ${2:-96} --workers 8 --decode device --passes 7 --per-rank 8 > gpurun_out/gc/g_$1.json 2> gpurun_out/gc/g.err
  python - <<P
import json, statistics
d = json.load(open("gpurun_out/gc/g_$1.json")); n = d["scenes"]
for k, v in d["worlds"]["1"].items():
    if isinstance(v, dict) and "passes_s" in v:
        p = v["passes_s"][1:]
        print(k.split(".")[0][:12], "median %.1f best %.1f scenes/s" % (n / statistics.median(p), n / min(p)), [round(x, 3) for x in p], end="; ")
print()
P
}
run 0; run 1; run 0; run 1; run 0 192; run 1 192
