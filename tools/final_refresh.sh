#!/bin/bash
# Round-end refresh on the GPU box: tests, smoke, bench lines, rocprofv3 summaries of the final build (copied to profiles/ afterwards).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
O=$ROOT/gpurun_out/final
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --gpus 2 --steps 10 > $O/bench_n2.json 2> $O/bench_n2.err
bash tools/profile.sh final_vc > /dev/null 2>&1
bash tools/profile.sh final_low --workload low > /dev/null 2>&1
bash tools/profile.sh final_high --workload high > /dev/null 2>&1
bash tools/profile.sh final_dense --variant dense > /dev/null 2>&1
bash tools/profile_scene.sh > /dev/null 2>&1
bash tools/profile_scene_pmc.sh > /dev/null 2>&1
python tools/heads_bench.py > $O/heads.md 2> $O/heads.err
cat $O/pytest_gpu.txt $O/smoke.txt
tail -c 300 $O/bench_n1.json
