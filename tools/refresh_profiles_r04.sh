#!/bin/bash
# Round 4, second half of the refresh: rocprofv3 summaries (kernel-trace stats + PMC passes) of the K3 legs, ScanNet-shape A/B.
T=r04
O=gpurun_out/final_$T; mkdir -p $O
timeout 200 python tools/ab_scannet.py --steps 30 > $O/ab_scannet.txt 2>&1
for leg in "corr_vc" "compact_vc --variant compact" "minimal_vc --variant minimal" "dense_xyz_vc --variant dense_xyz" "corr_low --workload low" "corr_high --workload high" "compact_low --variant compact --workload low" "compact_high --variant compact --workload high"; do
  set -- $leg
  tag=$1; shift
  timeout 400 bash tools/profile.sh ${T}_$tag "$@" > /dev/null 2>&1
done
cat $O/ab_scannet.txt
for leg in corr_vc compact_vc minimal_vc dense_xyz_vc corr_low corr_high compact_low compact_high; do echo "== $leg"; head -12 gpurun_out/prof_${T}_$leg/summary.md | cut -c1-220; done
