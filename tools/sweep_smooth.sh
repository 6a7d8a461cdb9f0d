#!/bin/bash
# From-disk sweeps on SMOOTH frames (Paeth rows: what real sensor depth gets) beside the noisy ones, decode on the device and on the
# host, inside one box: tools/dropin_ranks.py --ranks 1 --scenes 48 --workers 8 --passes 4 --per-rank 8 [--smooth] --decode ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/smooth
for kind in noisy smooth; do
  for dec in device host; do
    flag=""; [ $kind = smooth ] && flag="--smooth"
    timeout 400 python tools/dropin_ranks.py --ranks 1 --scenes 48 --workers 8 --decode $dec --passes 4 --per-rank 8 $flag > gpurun_out/smooth/${kind}_${dec}.json 2> gpurun_out/smooth/${kind}_${dec}.err
  done
done
python tools/show_ranks.py gpurun_out/smooth/*.json | tee gpurun_out/smooth/summary.txt
