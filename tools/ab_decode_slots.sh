#!/bin/bash
# Slots / frames in flight of the on-device decode inside ONE box (after the streams became the process's own: the earlier rounds of
# this A/B ran 3-4 passes per process and from the third on torch's stream pool had wrapped onto the decode slots' streams).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/slots
run() {
  echo "== slots $1 cap $2"
  MSPA_DECODE_SLOTS=$1 MSPA_DECODE_MAX_FRAMES=$2 timeout 500 python tools/dropin_ranks.py --ranks 1 --scenes 96 --workers 8 --decode device --passes 4 --per-rank 8 > gpurun_out/slots/s_$1_$2.json 2> gpurun_out/slots/s.err
  python tools/show_ranks.py gpurun_out/slots/s_$1_$2.json | grep -v "^#" | grep -v "rank0\|cfs" | cut -c1-200
}
run 8 2560
run 10 3200
run 12 3840
run 9 2560
run 10 2560
run 8 2560
run 6 1920
