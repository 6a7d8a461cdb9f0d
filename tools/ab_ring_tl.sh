#!/bin/bash
# (Run when the 4 KB ring was the default and tools/ab/libmspa_ring2k.so the variant; today the default is 2 KB and the variant is
# tools/build_variant.sh ring4k -DMSPA_INFLATE_RING=4096 -- tools/ring_validate.sh is the current form of this A/B.)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
V=$ROOT/tools/ab/libmspa_ring2k.so
run() {
  L=$ROOT/multi-spatialmllm_amd/libmspa.so; [ $1 = ring2k ] && L=$V
  echo -n "$1 slots $2 cap $3: "
  MSPA_LIB=$L MSPA_DECODE_SLOTS=$2 MSPA_DECODE_MAX_FRAMES=$3 timeout 300 python tools/sweep_timeline.py --scenes 192 --brief 2>/dev/null | tail -1
}
run base 8 2560
run ring2k 8 2560
run ring2k 10 3200
run ring2k 11 3520
run ring2k 12 3840
run base 10 3200
run base 8 2560
