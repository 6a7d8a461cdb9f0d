#!/bin/bash
# A/B builds of libmspa.so on the same GPU box: every tools/ab/libmspa_*.so and the in-tree library, interleaved runs of the
# bench's three-point sweep (+ the variant legs named in $1).  Prints kernel ms per leg.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
ARGS="--steps 30 --warmup 3 --no-cpu-baseline --no-scene-legs --also ${1:-minimal:fast,dense:fast}"
for i in 1 2; do
  for lib in tools/ab/libmspa_*.so multi-spatialmllm_amd/libmspa.so; do
    MSPA_LIB=$ROOT/$lib python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read())
print('%-34s' % '$lib'.split('/')[-1], ' '.join('%s %.4f' % (k, v['kernel_ms']) for k,v in j['sweep'].items()), ' '.join('%s %.4f' % (k, v['kernel_ms']) for k,v in j['variants'].items() if v))"
  done
done
