#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
run() { # tag prepare pace slots cap
  echo -n "$1 prepare=$2 slots $4 cap $5: "
  MSPA_PREPARE_ON_LOADER=$2 MSPA_DECODE_SLOTS=$4 MSPA_DECODE_MAX_FRAMES=$5 MSPA_LOOKAHEAD=2 timeout 300 python tools/sweep_timeline.py --scenes 192 --brief 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('scenes_per_s', 'host', 'cpus_busy', 'mid_region_scenes_per_s', 'slot_held_ms', 'stage_ms', 'h2d_ms', 'inflate_ms', 'unfilter_ms', 'consumer_waits_in_finish_decode_ms', 'consumer_holds_ms')})"
}
for i in 1 2 3; do
run a 0 0 8 2560
run b 1 0 8 2560
done
