#!/bin/bash
# The 2 KB ring as the default: device-ingest tests, the out-of-suite fuzz (three seeds), the gate benchmark per streams in flight
# for both kinds of frames, beside the 4 KB build (tools/ab/libmspa_ring4k.so).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/ringv
timeout 600 python -m pytest tests/test_gpu_device_ingest.py -x -q -m gpu 2>&1 | tail -2
for seed in 11 12 13; do timeout 600 python tools/fuzz_device_inflate.py --rounds 24 --streams 256 --seed $seed 2>/dev/null | tail -1 | cut -c1-600; done
for n in 2560 3584 4096; do
  for lib in ring2k ring4k; do
    L=$ROOT/multi-spatialmllm_amd/libmspa.so; [ $lib = ring4k ] && L=$ROOT/tools/ab/libmspa_ring4k.so
    for kind in "" "--smooth"; do
      echo -n "gate $lib $n $kind: "; MSPA_LIB=$L timeout 200 python tools/device_ingest_bench.py --streams $n --reps 3 $kind 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); v = d['device']
print(v['inflate_adler_ms'], 'ms +', v['unfilter_ms'], 'ms', v['frames_per_s'], 'frames/s; device/host', d.get('device_over_host'), 'identical', d['bit_identical_to_the_rendered_frames'], 'bad', d['status_nonzero'])"
    done
  done
done
