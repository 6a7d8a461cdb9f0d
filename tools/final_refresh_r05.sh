#!/bin/bash
# Round-5 refresh on the GPU box: tests, smoke, bench lines (pair workload + scene workload), rocprofv3 summaries of the current
# build (copied to profiles/ afterwards by tools/collect_profiles_r05.sh).  usage: tools/final_refresh_r05.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-r05}
cd $ROOT
O=$ROOT/gpurun_out/final_$T
mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | tail -1 > $O/smoke.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --gpus 2 --steps 10 --no-scene-legs > $O/bench_n2.json 2> $O/bench_n2.err
MSPA_BENCH_FORCE_DIST=1 python bench.py --no-scene-legs --also none --no-sweep > $O/bench_n1_rccl.json 2> $O/bench_n1_rccl.err
python bench.py --workload scenes > $O/bench_scenes_n1.json 2> $O/bench_scenes_n1.err
MSPA_BENCH_FORCE_DIST=1 python bench.py --workload scenes > $O/bench_scenes_n1_rccl.json 2> $O/bench_scenes_n1_rccl.err
python bench.py --workload scenes --gpus 2 --steps 5 > $O/bench_scenes_n2.json 2> $O/bench_scenes_n2.err
bash tools/profile.sh ${T}_corr_vc > /dev/null 2>&1
bash tools/profile.sh ${T}_compact_vc --variant compact > /dev/null 2>&1
bash tools/profile.sh ${T}_minimal_vc --variant minimal > /dev/null 2>&1
bash tools/pmc.sh ${T}_k3corr pair_fast_tight python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-scene-legs --also none --no-sweep > /dev/null 2>&1
bash tools/pmc.sh ${T}_rect_minimal pair_fast_tight python $ROOT/tools/ab_scannet.py --legs minimal:rect --rounds 1 --steps 6 > /dev/null 2>&1
bash tools/pmc.sh ${T}_rect_corr pair_fast_tight python $ROOT/tools/ab_scannet.py --legs corr:rect --rounds 1 --steps 6 > /dev/null 2>&1
bash tools/profile_scene.sh > /dev/null 2>&1
bash tools/profile_scene_pmc.sh > /dev/null 2>&1
# the scene-shaped workload under --kernel-trace --stats (which kernels a step consists of, and their share)
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/scenes_stats -o p -- python $ROOT/bench.py --workload scenes --steps 5 > $O/scenes_stats.log 2>&1 )
python - <<PY > $O/scenes_stats.md
import csv, glob
print("# rocprofv3 --kernel-trace --stats of \`python bench.py --workload scenes --steps 5\` (8 scenes of 160..400 frames x 131 072 vertices per step)\n")
print("| kernel | calls | avg us | min us | max us | % of GPU time |")
print("|---|---|---|---|---|---|")
for f in glob.glob("$O/scenes_stats/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5 or "mspa::" in r["Name"]:
            print(f"| \`{r['Name'].split('(')[0][-70:]}\` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.1f} |")
PY
rm -rf $O/scenes_stats
python tools/ab_k3.py --sets corr,compact,minimal,dense_xyz,dense --steps 30 --rounds 3 > $O/ab_k3.txt 2>&1
python tools/ab_k1.py > $O/ab_k1.txt 2>&1
python tools/ab_scannet.py --steps 30 > $O/ab_scannet.txt 2>&1
python tools/heads_bench.py > $O/heads.md 2> $O/heads.err
{ python tools/ingest_bench.py; MSPA_INGEST_ZLIB=1 python tools/ingest_bench.py; } 2>&1 | grep -v "^Data from\|amdgpu" > $O/ingest_bench.txt
cat $O/pytest_gpu.txt $O/smoke.txt $O/ab_k3.txt $O/ab_k1.txt $O/ab_scannet.txt
tail -c 300 $O/bench_n1.json
