#!/bin/bash
# (Run when the 4 KB ring was the default and tools/ab/libmspa_ring2k.so the variant; today the default is 2 KB and the variant is
# tools/build_variant.sh ring4k -DMSPA_INFLATE_RING=4096 -- tools/ring_validate.sh is the current form of this A/B.)
# A/B inside one box: the inflate kernel with a 2 KB ring (9 KB of LDS per wave) against the 4 KB ring (11 KB): the gate benchmark
# per streams in flight, the device-ingest tests on the variant, then the from-disk sweep per slots / frames in flight.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/ring
V=$ROOT/tools/ab/libmspa_ring2k.so
echo "== tests on the variant"; MSPA_LIB=$V timeout 600 python -m pytest tests/test_gpu_device_ingest.py -x -q -m gpu 2>&1 | tail -2
for n in 2560 3072 3584 4096; do
  for lib in base ring2k; do
    L=$ROOT/multi-spatialmllm_amd/libmspa.so; [ $lib = ring2k ] && L=$V
    echo -n "gate $lib $n: "; MSPA_LIB=$L timeout 200 python tools/device_ingest_bench.py --streams $n --reps 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); v = d['device']
print(v['inflate_adler_ms'], 'ms', v['frames_per_s'], 'frames/s', 'identical', d['bit_identical_to_the_rendered_frames'], 'bad', d['status_nonzero'])"
  done
done
echo -n "gate ring2k smooth 3584: "; MSPA_LIB=$V timeout 200 python tools/device_ingest_bench.py --streams 3584 --reps 3 --smooth 2>/dev/null | tail -1 | cut -c1-400
run() {  # tag lib slots cap
  L=$ROOT/multi-spatialmllm_amd/libmspa.so; [ $2 = ring2k ] && L=$V
  echo "== sweep $1: $2 slots $3 cap $4"
  MSPA_LIB=$L MSPA_DECODE_SLOTS=$3 MSPA_DECODE_MAX_FRAMES=$4 timeout 300 python tools/dropin_ranks.py --ranks 1 --scenes 96 --workers 8 --decode device --passes 3 --per-rank 8 > gpurun_out/ring/$1.json 2> gpurun_out/ring/$1.err
  python tools/show_ranks.py gpurun_out/ring/$1.json | grep -v "^#" | grep -v "rank0\|cfs" | cut -c1-200
}
run a base 8 2560
run b ring2k 8 2560
run c ring2k 10 3200
run d ring2k 11 3520
run e ring2k 12 3840
run f base 10 3200
run g base 8 2560
