#!/bin/bash
# After a tools/final_refresh_r03.sh run (or single tools/profile.sh / tools/pmc.sh passes) came back under gpurun_out/: copy the
# summaries that are to be judged into profiles/ (tracked) and refresh profiles/traffic.json from the PMC numbers.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
T=${1:-r03}
cpif() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
for leg in corr_vc corr_low corr_high compact_vc dense_vc dense_xyz_vc minimal_vc; do
  cpif gpurun_out/prof_${T}_${leg}/summary.md profiles/${T}_k3_${leg}_pmc.md
done
cpif gpurun_out/pmc_${T}_k3corr/summary.md profiles/${T}_k3_memory_path_pmc.md
cpif gpurun_out/pmc_${T}_k3compact/summary.md profiles/${T}_k3_compact_memory_path_pmc.md
cpif gpurun_out/prof_scene/summary.md profiles/${T}_scene_kernels_stats.md
cpif gpurun_out/prof_scene_pmc/summary.md profiles/${T}_scene_kernels_pmc.md
for f in bench_n1 bench_n2 bench_n1_rccl; do
  src=gpurun_out/final_${T}/$f.json
  [ -s $src ] && tail -1 $src > profiles/${T}_${f/bench_n2/bench_n2_shared_gpu}.json && echo "  profiles/${T}_${f}.json"
done
cpif gpurun_out/final_${T}/ab_k3.txt profiles/${T}_ab_k3.txt
cpif gpurun_out/final_${T}/ab_k1.txt profiles/${T}_ab_k1.txt
cpif gpurun_out/final_${T}/heads.md profiles/${T}_heads_throughput.md
cpif gpurun_out/final_${T}/scaled.txt profiles/${T}_scannet_shape.txt
args=""
for kv in corr:fast:vc=corr_vc corr:fast:low=corr_low corr:fast:high=corr_high compact:fast:vc=compact_vc dense:fast:vc=dense_vc dense_xyz:fast:vc=dense_xyz_vc minimal:fast:vc=minimal_vc; do
  key=${kv%%=*}; leg=${kv##*=}
  [ -s gpurun_out/prof_${T}_${leg}/traffic_entry.json ] && args="$args $key=${T}_${leg}:profiles/${T}_k3_${leg}_pmc.md"
done
[ -n "$args" ] && python tools/emit_traffic.py $args
