#!/bin/bash
# Round 5, GPU call D: PMC passes of the rectangular-tile kernel at ScanNet's shape (minimal on 64-row tiles, corr), and the
# from-disk sweep with 2 / 3 / 4 scenes in flight on the host.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
bash tools/pmc.sh r05_rect_minimal pair_fast_tight python $ROOT/tools/ab_scannet.py --legs minimal:rect --rounds 1 --steps 6 > /dev/null 2>&1
bash tools/pmc.sh r05_rect_corr pair_fast_tight python $ROOT/tools/ab_scannet.py --legs corr:rect --rounds 1 --steps 6 > /dev/null 2>&1
head -8 gpurun_out/pmc_r05_rect_minimal/summary.md
python - <<'PY' > gpurun_out/r05_dropin_lookahead.txt 2>&1
import json, os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "multi-spatialmllm_amd")]
import bench
from spatial_engine.utils.scannet_utils.handler import info_handler as IH
orig = IH.SceneInfoHandler.prefetched_scenes
for la in (2, 3, 4, 2, 4):
    def patched(self, scene_ids, num_workers=8, device="cuda", timings=None, with_points=True, lookahead=None, _la=la):
        return orig(self, scene_ids, num_workers, device, timings, with_points, _la)
    IH.SceneInfoHandler.prefetched_scenes = patched
    d = bench.time_dropin_sweep()
    print("lookahead", la, "scenes/s", d["scenes_per_s"], "seconds", d["seconds"], json.dumps(d["stage_busy_s"]))
PY
cat gpurun_out/r05_dropin_lookahead.txt | grep lookahead
