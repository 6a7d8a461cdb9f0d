#!/bin/bash
# After a tools/final_refresh_r05.sh run (or single tools/profile.sh / tools/pmc.sh passes) came back under gpurun_out/: copy the
# summaries that are to be judged into profiles/ (tracked) and refresh profiles/traffic.json from the PMC numbers.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
T=${1:-r05}
cpif() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; }
for leg in corr_vc corr_low corr_high compact_vc compact_low compact_high dense_xyz_vc minimal_vc; do
  cpif gpurun_out/prof_${T}_${leg}/summary.md profiles/${T}_k3_${leg}_pmc.md
done
cpif gpurun_out/pmc_${T}_k3corr/summary.md profiles/${T}_k3_memory_path_pmc.md
cpif gpurun_out/pmc_${T}_k3compact/summary.md profiles/${T}_k3_compact_memory_path_pmc.md
cpif gpurun_out/pmc_${T}_rect_minimal/summary.md profiles/${T}_rect_minimal_pmc.md
cpif gpurun_out/pmc_${T}_rect_corr/summary.md profiles/${T}_rect_corr_pmc.md
cpif gpurun_out/final_${T}/scenes_stats.md profiles/${T}_scenes_workload_stats.md
for f in bench_scenes_n1 bench_scenes_n1_rccl bench_scenes_n2; do
  src=gpurun_out/final_${T}/$f.json
  [ -s $src ] && tail -1 $src > profiles/${T}_${f/bench_scenes_n2/bench_scenes_n2_shared_gpu}.json && echo "  profiles/${T}_${f}.json"
done
cpif gpurun_out/prof_scene/summary.md profiles/${T}_scene_kernels_stats.md
cpif gpurun_out/prof_scene_pmc/summary.md profiles/${T}_scene_kernels_pmc.md
for f in bench_n1 bench_n2 bench_n1_rccl; do
  src=gpurun_out/final_${T}/$f.json
  [ -s $src ] && tail -1 $src > profiles/${T}_${f/bench_n2/bench_n2_shared_gpu}.json && echo "  profiles/${T}_${f}.json"
done
cpif gpurun_out/final_${T}/ab_k3.txt profiles/${T}_ab_k3.txt
cpif gpurun_out/final_${T}/ab_k1.txt profiles/${T}_ab_k1.txt
cpif gpurun_out/final_${T}/ab_scannet.txt profiles/${T}_ab_scannet.txt
cpif gpurun_out/final_${T}/heads.md profiles/${T}_heads_throughput.md
cpif gpurun_out/final_${T}/ingest_bench.txt profiles/${T}_ingest_bench.txt
args=""
for kv in corr:fast:vc=corr_vc corr:fast:low=corr_low corr:fast:high=corr_high compact:fast:vc=compact_vc compact:fast:low=compact_low compact:fast:high=compact_high dense_xyz:fast:vc=dense_xyz_vc minimal:fast:vc=minimal_vc; do
  key=${kv%%=*}; leg=${kv##*=}
  [ -s gpurun_out/prof_${T}_${leg}/traffic_entry.json ] && args="$args $key=${T}_${leg}:profiles/${T}_k3_${leg}_pmc.md"
done
[ -n "$args" ] && python tools/emit_traffic.py $args
# K1's entry of profiles/traffic.json from this round's scene PMC summary (read + written MB of the compacted K1 kernel)
python - <<'PY'
import json, re, os
src = "profiles/r05_scene_kernels_pmc.md"
if os.path.exists(src):
    for line in open(src):
        if "vertex_visibility_compact_kernel" in line:
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            rd, wr = float(cells[-2]), float(cells[-1])
            t = json.load(open("profiles/traffic.json"))
            t["K1_vertex_visibility"] = {"kernel": "mspa::vertex_visibility_compact_kernel<true>", "images": 320, "vertices": 131072,
                                         "hbm_bytes_per_launch": int((rd + wr) * 1e6),
                                         "source": f"{src} ({rd:.1f} MB read + {wr:.1f} MB written per 320-image scene; mean over the bench's "
                                                   "launches on the shuffled and on the Morton-ordered cloud)"}
            json.dump(t, open("profiles/traffic.json", "w"), indent=1)
            print("K1_vertex_visibility", t["K1_vertex_visibility"])
            break
PY
