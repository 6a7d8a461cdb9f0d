#!/bin/bash
# A/B baselines: the kernels of an earlier revision compiled against round 4's frame-record stride (8 slots), so that
# tools/ab_k3.py / ab_k1.py can time them in the same process, on the same inputs, as the in-tree library.
#   tools/build_rev_variant.sh [REV [NAME]]   -> tools/ab/libmspa_NAME.so   (default: round 3's final kernels, "r03guard")
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-7182b9d}
NAME=${2:-r03guard}
B=/tmp/mspa_src_$NAME
rm -rf $B && mkdir -p $B/multi-spatialmllm_amd/csrc $B/include $ROOT/tools/ab
cd $ROOT
for f in api pair_reproject vertex_visibility pair_overlap pose_tracks samples object_extents bitset_index; do
  git show $REV:multi-spatialmllm_amd/csrc/$f.hip > $B/multi-spatialmllm_amd/csrc/$f.hip
done
git show $REV:multi-spatialmllm_amd/csrc/mspa_common.h > $B/multi-spatialmllm_amd/csrc/mspa_common.h
git show $REV:include/mspa.h | sed 's/#define MSPA_FRAME_MATS 7/#define MSPA_FRAME_MATS 8/' > $B/include/mspa.h
# ... and against round 4's image-record stride of K1 / K6b (3 slots of 16 instead of 2)
sed -i 's/(int64_t)\(img\|(img0 + im)\|(img0 + q)\|(img0 + (int)(threadIdx.x - 128))\|(img0 + tid - 128)\) \* 32\b/(int64_t)\1 * 48/g' $B/multi-spatialmllm_amd/csrc/vertex_visibility.hip $B/multi-spatialmllm_amd/csrc/samples.hip
cd $B/multi-spatialmllm_amd/csrc
for f in api pair_reproject vertex_visibility pair_overlap pose_tracks samples object_extents bitset_index; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-result -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/libmspa_$NAME.so *.o -lz
echo built tools/ab/libmspa_$NAME.so
