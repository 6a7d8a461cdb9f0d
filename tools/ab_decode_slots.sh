#!/bin/bash
# Frames in flight on the device decode (upload.DECODE_SLOTS / DECODE_MAX_FRAMES) inside ONE box: the two from-disk sweeps, 48 scenes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for round in 1 2; do
for cfg in "8 2560" "9 2560" "10 3200" "12 3584" "12 9999"; do
  set -- $cfg
  MSPA_DECODE_SLOTS=$1 MSPA_DECODE_MAX_FRAMES=$2 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes 48 --workers 8 --decode device --passes 4 --per-rank 8 > gpurun_out/ab_slots_$1_$2.json 2> gpurun_out/ab_slots.err
  echo "== slots $1 cap $2 round $round"; python tools/show_ranks.py gpurun_out/ab_slots_$1_$2.json 2>&1 | grep "scenes/s\|produce" | cut -c1-200
done; done
