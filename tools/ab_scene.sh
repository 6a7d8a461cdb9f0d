#!/bin/bash
# A/B of the scene-level kernels (K1 / K2 / K4 legs of bench.py) across tools/ab/libmspa_*.so and the in-tree library.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for i in 1 2; do
  for lib in tools/ab/libmspa_*.so multi-spatialmllm_amd/libmspa.so; do
    MSPA_LIB=$ROOT/$lib python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-traffic --no-sweep --also none 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read())
s=j['variants']['scene']
print('%-28s' % '$lib'.split('/')[-1], ' '.join('%s %.4f' % (k, v.get('kernel_ms', v.get('ms'))) for k,v in s.items()))"
  done
done
