#!/bin/bash
# Build a variant of libmspa.so with extra compiler flags into tools/ab/libmspa_<name>.so (A/B runs: tools/ab.sh).
# usage: tools/build_variant.sh <name> [flags...]     e.g.  tools/build_variant.sh r32w6 -DMSPA_TIGHT_ROWS=32 -DMSPA_TIGHT_WAVES=6
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
B=/tmp/mspa_variant_$NAME
mkdir -p $B $ROOT/tools/ab
cd $ROOT/multi-spatialmllm_amd/csrc
for f in api pair_reproject vertex_visibility pair_overlap pose_tracks samples object_extents bitset_index host_ingest device_ingest format_lists; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result "$@" -c $f.hip -o $B/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/libmspa_$NAME.so $B/*.o -lz
echo built tools/ab/libmspa_$NAME.so
