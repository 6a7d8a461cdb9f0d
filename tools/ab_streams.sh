#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/streams
timeout 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_device_ingest.py -x -q -m gpu 2>&1 | tail -2
tl() { # prepare hwq
  echo -n "timeline prepare=$1 hwq=$2: "
  GPU_MAX_HW_QUEUES=$2 MSPA_PREPARE_ON_LOADER=$1 timeout 300 python tools/sweep_timeline.py --scenes 192 --brief 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('scenes_per_s', 'mid_region_scenes_per_s', 'slot_held_ms', 'stage_ms', 'h2d_ms', 'inflate_ms', 'unfilter_ms', 'consumer_waits_in_finish_decode_ms', 'consumer_holds_ms')})"
}
for i in 1 2 3; do tl 0 16; tl 1 16; tl 1 24; tl 0 24; done
for i in 1 2; do
  timeout 400 python tools/dropin_ranks.py --ranks 1 --scenes 96 --workers 8 --decode device --passes 6 --per-rank 8 > gpurun_out/streams/p6_$i.json 2> gpurun_out/streams/p6_$i.err
  python tools/show_ranks.py gpurun_out/streams/p6_$i.json | grep -v "^#" | grep -v "rank0\|cfs" | cut -c1-220
done
