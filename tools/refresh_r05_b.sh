#!/bin/bash
# Round 5, GPU call B: per-workgroup composition and batched culled-tile loads, A/B at ScanNet's shape and at 640x480.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05b
mkdir -p $O
python tools/ab_scannet.py --steps 30 > $O/ab_scannet.txt 2>&1
cat $O/ab_scannet.txt
python tools/ab_k3.py --sets corr,compact,minimal --steps 30 --rounds 3 > $O/ab_k3.txt 2>&1
cat $O/ab_k3.txt
MSPA_LIB=$ROOT/tools/ab/libmspa_wgc_cb16.so python -m pytest tests/test_gpu_rect.py tests/test_scannet_shape.py tests/test_gpu_guard.py tests/test_gpu_tight.py tests/test_gpu_compact.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -4 > $O/wgc_cb16_parity.txt
cat $O/wgc_cb16_parity.txt
