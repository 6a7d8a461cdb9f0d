#!/bin/bash
# Admission of the on-device decode: frames leave the in-flight count when the consumer TAKES the scene (default) or when their decode
# has COMPLETED on the device (MSPA_DECODE_GATE=complete, more slots than the cap's worth of scenes).  One box, 96 scenes, 7 passes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT; mkdir -p gpurun_out/gate
run() {
  echo -n "gate $1 slots $2 cap $3: "
  MSPA_DECODE_GATE=$1 MSPA_DECODE_SLOTS=$2 MSPA_DECODE_MAX_FRAMES=$3 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes 96 --workers 8 --decode device --passes 7 --per-rank 8 > gpurun_out/gate/g_$1_$2_$3.json 2> gpurun_out/gate/g.err
  python - <<P
import json, statistics
d = json.load(open("gpurun_out/gate/g_$1_$2_$3.json"))["worlds"]["1"]
for k, v in d.items():
    if isinstance(v, dict) and "passes_s" in v:
        p = v["passes_s"][1:]
        print(k.split(".")[0][:12], "median %.1f best %.1f scenes/s" % (96 / statistics.median(p), 96 / min(p)), [round(x, 3) for x in p], end="; ")
print()
P
}
run taken 8 2560
run complete 10 2560
run complete 12 2560
run complete 16 2560
run complete 12 3200
run taken 8 2560
run complete 12 2560
