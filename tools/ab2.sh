#!/bin/bash
# A/B with custom bench args: tools/ab2.sh <bench args...>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in 1 2 3; do
  for lib in tools/ab/libmspa_base.so multi-spatialmllm_amd/libmspa.so; do
    MSPA_LIB=$ROOT/$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-live-traffic --no-scene-legs --also none "$@" 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', j['roofline']['kernel_ms'], j['visible_fraction'])"
  done
done
