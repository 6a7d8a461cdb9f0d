// K2: all-pairs overlap of visibility bitsets (calculate_camera_overlap, CFR:102-137).
//
// One wave per pair: lanes stride over the two bitsets 16 bytes at a time (coalesced 1 KiB per
// wave-instruction), popcount a&b and a|b, reduce across the wave, lane 0 forms the float64
// percentage exactly as NumPy does: int64 / int64 (true divide) * 100.  A scene's bitsets
// (F x N/8 bytes: 1 MB at F = 64, N = 131072) sit in L2 / Infinity Cache after the first touch.
#include "mspa_common.h"

namespace mspa {

struct OverlapArgs {
    const uint64_t *bits;
    int64_t n_words;
    const int32_t *pairs;
    int64_t n_pairs;
    double *overlap;
    int32_t *inter;
    int32_t *uni;
};

constexpr int kOThreads = 256;

__global__ __launch_bounds__(kOThreads) void pair_overlap_kernel(OverlapArgs a) {
    const int64_t p = (int64_t)blockIdx.x * (kOThreads / kWave) + (threadIdx.x >> 6);
    if (p >= a.n_pairs) return;
    const int lane = threadIdx.x & 63;
    const uint64_t *__restrict__ ra = a.bits + (int64_t)a.pairs[2 * p + 0] * a.n_words;
    const uint64_t *__restrict__ rb = a.bits + (int64_t)a.pairs[2 * p + 1] * a.n_words;
    int inter = 0, uni = 0;
    const int64_t n2 = a.n_words >> 1;
    const bool aligned = ((a.n_words & 1) == 0);   // rows stay 16-byte aligned only for even n_words
    if (aligned) {
        const ulonglong2 *__restrict__ va = reinterpret_cast<const ulonglong2 *>(ra);
        const ulonglong2 *__restrict__ vb = reinterpret_cast<const ulonglong2 *>(rb);
        for (int64_t w = lane; w < n2; w += kWave) {
            const ulonglong2 x = va[w], y = vb[w];
            inter += __popcll(x.x & y.x) + __popcll(x.y & y.y);
            uni += __popcll(x.x | y.x) + __popcll(x.y | y.y);
        }
    } else {
        for (int64_t w = lane; w < a.n_words; w += kWave) {
            const uint64_t x = ra[w], y = rb[w];
            inter += __popcll(x & y);
            uni += __popcll(x | y);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        inter += __shfl_down(inter, off);
        uni += __shfl_down(uni, off);
    }
    if (lane == 0) {
        a.overlap[p] = (double)inter / (double)uni * 100.0;   // 0/0 -> NaN as in CFR:136
        if (a.inter) a.inter[p] = inter;
        if (a.uni) a.uni[p] = uni;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Tiled form (all pairs of a scene, or an [objects x images] rectangle): the reference's HOT LOOP 2 (CFR:176-189)
// visits F(F-1)/2 pairs, and one wave per pair re-reads both 16 KB rows from L2 for each of them (1.67 GB through L2
// for a 320-frame scene).  Here a one-wave workgroup owns a 32 x 32 block of (a, b) rows and a slice of the words:
// 16-word chunks of the 64 rows are staged in LDS (coalesced 16-byte loads, 128 contiguous bytes per row), every lane
// keeps a 4 x 4 sub-block of pair counters in registers and reads 4 + 4 row fragments per two words (ds_read_b128) for
// 32 pair-words -- 4 bytes of LDS traffic per pair-word instead of 16 bytes of L2 traffic, and only |a & b| is counted:
// the union is |a| + |b| - |a & b| with |a| = the diagonal entry.  The word dimension is split over workgroups so that a
// scene fills the chip; partial counts go to a caller-provided workspace (no atomics: device-scope atomics on 10^5
// addresses would serialise at the fabric) and a second kernel reduces the slices and forms the float64 percentage
// exactly as NumPy does (int / int true divide, * 100; 0 / 0 -> NaN, CFR:136).
// ---------------------------------------------------------------------------------------------------------
constexpr int kTile = 32;                  // rows of a and of b per workgroup
constexpr int kChunkWords = 16;            // bitset words staged per step
constexpr int kRowDw = kChunkWords * 2 + 4;   // LDS row stride in dwords: 16-byte aligned, bank-skewed (36)

struct TileArgs {
    const uint64_t *bits_a;
    const uint64_t *bits_b;
    int n_a, n_b;
    int64_t n_words;
    int tiles_b;
    int slices;
    int64_t words_per_slice;       // multiple of kChunkWords
    int symmetric;                 // bits_a == bits_b: tiles below the diagonal are skipped
    int32_t *partial;              // [tiles_a * tiles_b][slices][kTile * kTile]
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool ALIGNED16>
__global__ __launch_bounds__(kWave) void overlap_tile_kernel(TileArgs a) {
    const int tile = blockIdx.x;
    const int ta = tile / a.tiles_b, tb = tile - ta * a.tiles_b;
    if (a.symmetric && tb < ta) return;
    const int slice = blockIdx.y;
    const int lane = threadIdx.x;
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * kTile * kRowDw];
    // staging: 8 lanes cover one row's 128-byte chunk, a wave instruction covers 8 rows
    const int lrow = lane >> 3, lcol = lane & 7;
    const int ti = lane >> 3, tj = lane & 7;          // this lane's rows: a: ti + 8r, b: tj + 8c
    int acc[4][4] = {};
    const int64_t w_begin = (int64_t)slice * a.words_per_slice;
    const int64_t w_end = min(w_begin + a.words_per_slice, a.n_words);
    for (int64_t w0 = w_begin; w0 < w_end; w0 += kChunkWords) {
        u32x4 stage[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int r = k * 8 + lrow;                                  // 0..31 = a rows, 32..63 = b rows
            const bool is_b = r >= kTile;
            const int row = is_b ? tb * kTile + (r - kTile) : ta * kTile + r;
            const bool row_ok = row < (is_b ? a.n_b : a.n_a);
            const uint64_t *src = (is_b ? a.bits_b : a.bits_a) + (int64_t)row * a.n_words;
            const int64_t w = w0 + lcol * 2;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (row_ok) {
                if (ALIGNED16) {
                    if (w < w_end) v = *reinterpret_cast<const u32x4 *>(src + w);   // n_words even: w + 1 exists too
                } else {
                    const uint64_t x = w < w_end ? src[w] : 0ull, y = (w + 1) < w_end ? src[w + 1] : 0ull;
                    v = u32x4{(uint32_t)x, (uint32_t)(x >> 32), (uint32_t)y, (uint32_t)(y >> 32)};
                }
            }
            stage[k] = v;
        }
        __syncthreads();                                                 // the previous chunk has been consumed
#pragma unroll
        for (int k = 0; k < 8; ++k)
            *reinterpret_cast<u32x4 *>(&lds[(k * 8 + lrow) * kRowDw + lcol * 4]) = stage[k];
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < kChunkWords / 2; ++s2) {
            u32x4 va[4], vb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) va[r] = *reinterpret_cast<const u32x4 *>(&lds[(ti + 8 * r) * kRowDw + s2 * 4]);
#pragma unroll
            for (int c = 0; c < 4; ++c) vb[c] = *reinterpret_cast<const u32x4 *>(&lds[(kTile + tj + 8 * c) * kRowDw + s2 * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    acc[r][c] += __popc(va[r].x & vb[c].x) + __popc(va[r].y & vb[c].y) + __popc(va[r].z & vb[c].z) +
                                 __popc(va[r].w & vb[c].w);
        }
    }
    int32_t *dst = a.partial + ((int64_t)tile * a.slices + slice) * (kTile * kTile);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[(ti + 8 * r) * kTile + (tj + 8 * c)] = acc[r][c];
}

__device__ __forceinline__ int reduce_slices(const int32_t *partial, int tiles_b, int slices, int i, int j) {
    const int tile = (i / kTile) * tiles_b + (j / kTile);
    const int32_t *p = partial + (int64_t)tile * slices * (kTile * kTile) + (i % kTile) * kTile + (j % kTile);
    int sum = 0;
    for (int s = 0; s < slices; ++s) sum += p[(int64_t)s * (kTile * kTile)];
    return sum;
}

// [n_a, n_b] intersection counts; symmetric input: entries below the diagonal are mirrored
__global__ __launch_bounds__(256) void overlap_reduce_kernel(const int32_t *partial, int n_a, int n_b, int tiles_b,
                                                             int slices, int symmetric, int32_t *out) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n_b) return;
    const bool flip = symmetric && (j / kTile) < (i / kTile);
    out[(int64_t)i * n_b + j] = flip ? reduce_slices(partial, tiles_b, slices, j, i)
                                     : reduce_slices(partial, tiles_b, slices, i, j);
}

// all pairs i < j in the reference's nested-loop order (CFR:176-178): p = i*F - i(i+1)/2 + (j - i - 1)
__global__ __launch_bounds__(256) void scene_overlap_finalize_kernel(const int32_t *partial, int F, int tiles_b, int slices,
                                                                     double *overlap, int32_t *inter_out,
                                                                     int32_t *union_out) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= F || j <= i) return;
    const int inter = reduce_slices(partial, tiles_b, slices, i, j);
    const int uni = reduce_slices(partial, tiles_b, slices, i, i) + reduce_slices(partial, tiles_b, slices, j, j) - inter;
    const int64_t p = (int64_t)i * F - (int64_t)i * (i + 1) / 2 + (j - i - 1);
    overlap[p] = (double)inter / (double)uni * 100.0;          // 0/0 -> NaN as in CFR:136
    if (inter_out) inter_out[p] = inter;
    if (union_out) union_out[p] = uni;
}

struct TilePlan {
    int tiles_a, tiles_b, slices;
    int64_t words_per_slice, bytes;
};

static TilePlan plan_tiles(int n_a, int n_b, int64_t n_words, bool symmetric) {
    TilePlan p;
    p.tiles_a = (n_a + kTile - 1) / kTile;
    p.tiles_b = (n_b + kTile - 1) / kTile;
    const int64_t active = symmetric ? (int64_t)p.tiles_a * (p.tiles_a + 1) / 2 : (int64_t)p.tiles_a * p.tiles_b;
    const int64_t chunks = (n_words + kChunkWords - 1) / kChunkWords;
    int64_t slices = (2048 + active - 1) / active;            // ~2 waves per SIMD of one-wave workgroups
    if (slices < 1) slices = 1;
    if (slices > chunks) slices = chunks;
    const int64_t chunks_per_slice = (chunks + slices - 1) / slices;
    p.words_per_slice = chunks_per_slice * kChunkWords;
    p.slices = (int)((chunks + chunks_per_slice - 1) / chunks_per_slice);
    p.bytes = (int64_t)p.tiles_a * p.tiles_b * p.slices * (kTile * kTile) * (int64_t)sizeof(int32_t);
    return p;
}

static int launch_tiles(const uint64_t *bits_a, int n_a, const uint64_t *bits_b, int n_b, int64_t n_words, bool symmetric,
                        const TilePlan &p, int32_t *partial, hipStream_t s) {
    TileArgs t{bits_a, bits_b, n_a, n_b, n_words, p.tiles_b, p.slices, p.words_per_slice, symmetric ? 1 : 0, partial};
    const dim3 grid((uint32_t)(p.tiles_a * p.tiles_b), (uint32_t)p.slices);
    const bool aligned = (n_words % 2 == 0) && (((uintptr_t)bits_a & 15u) == 0) && (((uintptr_t)bits_b & 15u) == 0);
    if (aligned) hipLaunchKernelGGL(overlap_tile_kernel<true>, grid, dim3(kWave), 0, s, t);
    else hipLaunchKernelGGL(overlap_tile_kernel<false>, grid, dim3(kWave), 0, s, t);
    return check_hip(hipGetLastError(), "overlap_tile_kernel launch");
}

}  // namespace mspa

using namespace mspa;

extern "C" int64_t mspa_overlap_workspace_bytes(int32_t n_a, int32_t n_b, int64_t n_words) {
    if (n_a <= 0 || n_b <= 0 || n_words <= 0) return 0;
    // the symmetric plan never needs more than the rectangular one of the same size
    const TilePlan r = plan_tiles(n_a, n_b, n_words, false), q = plan_tiles(n_a, n_b, n_words, n_a == n_b);
    return r.bytes > q.bytes ? r.bytes : q.bytes;
}

extern "C" int mspa_overlap_matrix(const uint64_t *bits_a, int32_t n_a, const uint64_t *bits_b, int32_t n_b,
                                   int64_t n_words, void *workspace, int64_t workspace_bytes, int32_t *out_inter,
                                   mspa_stream_t stream) {
    if (n_a < 0 || n_b < 0 || n_words <= 0) return fail(MSPA_EINVAL, "mspa_overlap_matrix: bad size");
    if (n_a == 0 || n_b == 0) return MSPA_OK;
    if (!bits_a || !bits_b || !out_inter || !workspace) return fail(MSPA_EINVAL, "mspa_overlap_matrix: null pointer");
    if (n_words > (1LL << 25)) return fail(MSPA_EINVAL, "mspa_overlap_matrix: bitset too long for int32 counts");
    if (n_a > 65535 * kTile / 2 || n_b > 65535) return fail(MSPA_EINVAL, "mspa_overlap_matrix: too many rows; split the batch");
    const bool symmetric = (bits_a == bits_b) && (n_a == n_b);
    const TilePlan p = plan_tiles(n_a, n_b, n_words, symmetric);
    if (workspace_bytes < p.bytes) return fail(MSPA_EINVAL, "mspa_overlap_matrix: workspace smaller than mspa_overlap_workspace_bytes()");
    if (((uintptr_t)workspace & 3u) != 0) return fail(MSPA_EINVAL, "mspa_overlap_matrix: workspace must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_tiles(bits_a, n_a, bits_b, n_b, n_words, symmetric, p, (int32_t *)workspace, s);
    if (rc) return rc;
    hipLaunchKernelGGL(overlap_reduce_kernel, dim3((uint32_t)((n_b + 255) / 256), (uint32_t)n_a), dim3(256), 0, s,
                       (const int32_t *)workspace, n_a, n_b, p.tiles_b, p.slices, symmetric ? 1 : 0, out_inter);
    return check_hip(hipGetLastError(), "overlap_reduce_kernel launch");
}

extern "C" int mspa_scene_overlap(const uint64_t *bits, int32_t n_images, int64_t n_words, void *workspace,
                                  int64_t workspace_bytes, double *out_overlap, int32_t *out_inter, int32_t *out_union,
                                  mspa_stream_t stream) {
    if (n_images < 0 || n_words <= 0) return fail(MSPA_EINVAL, "mspa_scene_overlap: bad size");
    if (n_images < 2) return MSPA_OK;                               // no pair
    if (!bits || !out_overlap || !workspace) return fail(MSPA_EINVAL, "mspa_scene_overlap: null pointer");
    if (n_words > (1LL << 25)) return fail(MSPA_EINVAL, "mspa_scene_overlap: bitset too long for int32 counts");
    if (n_images > 65535) return fail(MSPA_EINVAL, "mspa_scene_overlap: too many images; split the scene");
    const TilePlan p = plan_tiles(n_images, n_images, n_words, true);
    if (workspace_bytes < p.bytes) return fail(MSPA_EINVAL, "mspa_scene_overlap: workspace smaller than mspa_overlap_workspace_bytes()");
    if (((uintptr_t)workspace & 3u) != 0) return fail(MSPA_EINVAL, "mspa_scene_overlap: workspace must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_tiles(bits, n_images, bits, n_images, n_words, true, p, (int32_t *)workspace, s);
    if (rc) return rc;
    hipLaunchKernelGGL(scene_overlap_finalize_kernel, dim3((uint32_t)((n_images + 255) / 256), (uint32_t)n_images), dim3(256),
                       0, s, (const int32_t *)workspace, n_images, p.tiles_b, p.slices, out_overlap, out_inter, out_union);
    return check_hip(hipGetLastError(), "scene_overlap_finalize_kernel launch");
}

extern "C" int mspa_pair_overlap(const uint64_t *bits, int32_t n_images, int64_t n_words, const int32_t *pairs,
                                 int64_t n_pairs, double *out_overlap, int32_t *out_inter, int32_t *out_union,
                                 mspa_stream_t stream) {
    if (!bits || !pairs || !out_overlap) return fail(MSPA_EINVAL, "mspa_pair_overlap: null pointer");
    if (n_images <= 0 || n_words <= 0 || n_pairs < 0) return fail(MSPA_EINVAL, "mspa_pair_overlap: bad size");
    if (n_words > (1LL << 25)) return fail(MSPA_EINVAL, "mspa_pair_overlap: bitset too long for int32 counts");
    if (n_pairs == 0) return MSPA_OK;
    OverlapArgs a{bits, n_words, pairs, n_pairs, out_overlap, out_inter, out_union};
    const int64_t blocks = (n_pairs + (kOThreads / kWave) - 1) / (kOThreads / kWave);
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_pair_overlap: too many pairs; split the batch");
    hipLaunchKernelGGL(pair_overlap_kernel, dim3((uint32_t)blocks), dim3(kOThreads), 0, (hipStream_t)stream, a);
    return check_hip(hipGetLastError(), "pair_overlap_kernel launch");
}
