#!/bin/bash
# Short round-end refresh: tests, smoke, the bench line and the scene-kernel profiles (the K3 PMC passes are in tools/final_refresh.sh).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/final
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --gpus 2 --steps 10 > $O/bench_n2.json 2> $O/bench_n2.err
bash tools/profile_scene.sh > /dev/null 2>&1
bash tools/profile_scene_pmc.sh > /dev/null 2>&1
cat $O/pytest_gpu.txt $O/smoke.txt
