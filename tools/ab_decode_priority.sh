#!/bin/bash
# A/B inside one box: decode streams at the device's lowest stream priority against plain streams, over slots / frames in flight.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/prio
run() {  # tag slots cap prio
  echo "== $1: slots $2 cap $3 priority ${4:-plain}"
  MSPA_DECODE_SLOTS=$2 MSPA_DECODE_MAX_FRAMES=$3 MSPA_DECODE_PRIORITY_EXPERIMENT=$4 timeout 300 python tools/dropin_ranks.py --ranks 1 --scenes 96 --workers 8 --decode device --passes 3 --per-rank 8 > gpurun_out/prio/$1.json 2> gpurun_out/prio/$1.err
  python tools/show_ranks.py gpurun_out/prio/$1.json | grep -v "^#" | cut -c1-260
}
run a_8_2560_plain 8 2560 ""
run b_8_2560_low 8 2560 low
run c_10_3200_low 10 3200 low
run d_12_3584_low 12 3584 low
run e_12_3584_plain 12 3584 ""
run f_16_5120_low 16 5120 low
run g_8_2560_plain 8 2560 ""
