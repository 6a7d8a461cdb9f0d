#!/bin/bash
# Round 5, GPU call A: the from-disk sweep test, ScanNet-shape tile-height A/B, K1 traffic of the variants.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05a
mkdir -p $O
python -m pytest tests/test_gpu_sweep.py -m gpu -x -q --tb=long 2>&1 | grep -v "sample_dataframe\|process_scene\|Start processing\|Finished scene\|^Data from\|run_split\]" | tail -150 > $O/sweep.txt
tail -3 $O/sweep.txt
python tools/ab_scannet.py --steps 30 > $O/ab_scannet.txt 2>&1
cat $O/ab_scannet.txt
MSPA_LIB=$ROOT/tools/ab/libmspa_srows64.so python -m pytest tests/test_gpu_rect.py tests/test_scannet_shape.py tests/test_gpu_guard.py tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -4 > $O/srows64_parity.txt
cat $O/srows64_parity.txt
python tools/ab_k1.py > $O/ab_k1.txt 2>&1
cat $O/ab_k1.txt
bash tools/pmc_k1_traffic.sh > $O/k1_traffic.md 2> $O/k1_traffic.err
cat $O/k1_traffic.md
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python - <<PY
import json
j = json.loads(open("$O/bench_n1.json").read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"], json.dumps(j["variants"]["dropin_sweep"]))
PY
