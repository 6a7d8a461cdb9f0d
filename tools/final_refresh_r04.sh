#!/bin/bash
# Round-4 refresh on the GPU box: tests, smoke, bench lines, rocprofv3 summaries of the current build (copied to profiles/
# afterwards by tools/collect_profiles_r04.sh).  usage: tools/final_refresh_r04.sh [tag]   (default tag: r04)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-r04}
cd $ROOT
O=$ROOT/gpurun_out/final_$T
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --gpus 2 --steps 10 --no-scene-legs > $O/bench_n2.json 2> $O/bench_n2.err
MSPA_BENCH_FORCE_DIST=1 python bench.py --no-scene-legs --also none --no-sweep > $O/bench_n1_rccl.json 2> $O/bench_n1_rccl.err
bash tools/profile.sh ${T}_corr_vc > /dev/null 2>&1
bash tools/profile.sh ${T}_corr_low --workload low > /dev/null 2>&1
bash tools/profile.sh ${T}_corr_high --workload high > /dev/null 2>&1
bash tools/profile.sh ${T}_compact_vc --variant compact > /dev/null 2>&1
bash tools/profile.sh ${T}_compact_low --variant compact --workload low > /dev/null 2>&1
bash tools/profile.sh ${T}_compact_high --variant compact --workload high > /dev/null 2>&1
bash tools/profile.sh ${T}_dense_xyz_vc --variant dense_xyz > /dev/null 2>&1
bash tools/profile.sh ${T}_minimal_vc --variant minimal > /dev/null 2>&1
bash tools/pmc.sh ${T}_k3corr pair_fast_tight python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-scene-legs --also none --no-sweep > /dev/null 2>&1
bash tools/profile_scene.sh > /dev/null 2>&1
bash tools/profile_scene_pmc.sh > /dev/null 2>&1
python tools/ab_k3.py --sets corr,compact,minimal,dense_xyz,dense --steps 30 --rounds 3 > $O/ab_k3.txt 2>&1
python tools/ab_k1.py > $O/ab_k1.txt 2>&1
python tools/ab_scannet.py --steps 30 > $O/ab_scannet.txt 2>&1
python tools/heads_bench.py > $O/heads.md 2> $O/heads.err
cat $O/pytest_gpu.txt $O/smoke.txt $O/ab_k3.txt $O/ab_k1.txt $O/ab_scannet.txt
tail -c 300 $O/bench_n1.json
