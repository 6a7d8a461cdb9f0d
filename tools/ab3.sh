#!/bin/bash
# A/B builds on one GPU box, variant legs only: every tools/ab/libmspa_*.so and the in-tree library, interleaved, two rounds.
# usage: tools/ab3.sh "compact:fast,corr:fast" [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
LEGS=${1:-compact:fast}; shift
ARGS="--steps 12 --warmup 3 --no-cpu-baseline --no-scene-legs --no-sweep --variant minimal --also $LEGS $*"
for i in 1 2; do
  for lib in tools/ab/libmspa_*.so multi-spatialmllm_amd/libmspa.so; do
    MSPA_LIB=$ROOT/$lib python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read())
print('%-34s' % '$lib'.split('/')[-1], 'minimal %.4f' % j['roofline']['kernel_ms'], ' '.join('%s %.4f' % (k, v['kernel_ms']) for k,v in j['variants'].items() if v))"
  done
done
