#!/bin/bash
# A/B of the ScanNet-shape leg (bench.time_scannet_shape) on one GPU box: every tools/ab/libmspa_*.so and the in-tree library, twice.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in 1 2; do
  for lib in tools/ab/libmspa_*.so multi-spatialmllm_amd/libmspa.so; do
    [ -e $lib ] || continue
    MSPA_LIB=$ROOT/$lib python - <<PY 2>/dev/null
import sys, os
sys.path[:0] = ["$ROOT/multi-spatialmllm_amd", "$ROOT"]
import torch, bench
from mspa import _lib
_lib.load()
r = bench.time_scannet_shape(torch.device("cuda", 0), steps=10)
print("%-28s scannet_shape kernel_ms %.4f (%d pairs)  ms per 1000 pairs %.4f  visible %.3f" % ("$lib".split("/")[-1], r["kernel_ms"], r["pairs"], r["ms_per_1000_pairs"], r["visible_fraction"]))
PY
  done
done
