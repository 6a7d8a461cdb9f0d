#!/bin/bash
# Round 5, GPU call C: tiles per wave (SCALED form) -- parity of the in-tree build and of the TPW = 2 build, then the A/B.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05c
mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_rect.py tests/test_scannet_shape.py -m gpu -x -q 2>&1 | tail -6 > $O/intree_parity.txt
cat $O/intree_parity.txt
MSPA_LIB=$ROOT/tools/ab/libmspa_tpw2.so timeout 400 python -m pytest tests/test_gpu_rect.py tests/test_scannet_shape.py -m gpu -x -q 2>&1 | tail -12 > $O/tpw2_parity.txt
cat $O/tpw2_parity.txt
if grep -q "passed" $O/tpw2_parity.txt && ! grep -q "failed\|fault" $O/tpw2_parity.txt; then
  timeout 400 python tools/ab_scannet.py --steps 30 > $O/ab_scannet.txt 2>&1
  cat $O/ab_scannet.txt
fi
