#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/{stats,sq,sq2,fetch,write,tcc}
# PMC passes never combine with --sys-trace / hip/hsa traces (see task notes); one counter group per pass.
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-live-traffic --no-scene-legs --also none --no-sweep $*"
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace "$@" --output-format csv -d $OUT/$name -o p -- $B > $OUT/$name.log 2>&1; }
run stats --stats
run sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcc --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum
python $ROOT/tools/rocprof_summary.py $OUT ${MSPA_PROF_PATTERN:-pair_} | tee $OUT/summary.md
# the raw CSVs are tens of MB per pass and gpurun brings back 64 MiB in all: keep the summary and the two numbers
# tools/emit_traffic.py needs, drop the rest
MSPA_PROF_PATTERN=${MSPA_PROF_PATTERN:-pair_} python $ROOT/tools/emit_traffic.py --entry $OUT > $OUT/traffic_entry.json
for d in stats sq sq2 fetch write tcc; do rm -rf $OUT/$d; done
