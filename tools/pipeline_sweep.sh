#!/bin/bash
# bench.py's pipeline leg (host memory -> pair table, 24 scenes x 320 frames) over the staging knobs of mspa/upload.py.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for cfg in ${CFGS:-"2 1 320" "3 4 160" "3 8 160" "3 8 64" "3 8 320" "4 8 160" "3 12 160"}; do
  set -- $cfg
  MSPA_UPLOAD_SLOTS=$1 MSPA_STAGE_THREADS=$2 MSPA_STAGE_CHUNK_FRAMES=$3 python - <<PY
import sys
sys.argv = ["bench.py"]
import bench
r = [bench.time_scene_pipeline("cuda:0")["scenes_per_s"] for _ in range(3)]
print("slots $1 threads $2 chunk $3:", r, "scenes/s")
PY
done
