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
// for a 320-frame scene) and popcounts a & b AND a | b.  Here a workgroup owns a 32 x 32 block of (a, b) rows and a
// 64-word slice of the bitsets; each of its four waves takes one 16-word chunk of that slice: the 64 rows' chunk is staged
// in the wave's own LDS region (coalesced 16-byte loads, 128 contiguous bytes per row), every lane keeps a 4 x 4 sub-block
// of pair counters in registers and reads 4 + 4 row fragments per two words (ds_read_b128) for 32 pair-words -- 4 bytes of
// LDS traffic per pair-word instead of 16 bytes of L2 traffic -- and only |a & b| is counted: the union is
// |a| + |b| - |a & b| with |a| the diagonal entry.  The four waves' counters are summed through LDS, and the block writes
// ONE partial 32 x 32 table to a caller-provided workspace (no atomics: device-scope atomics on 10^5 addresses serialise
// at the fabric); small kernels reduce the slices and form the float64 percentage exactly as NumPy does (int / int true
// divide, * 100; 0 / 0 -> NaN, CFR:136).  Work is cut this finely because the popcounts are VALU-bound (v_and + v_bcnt per
// 32 bits, ~7 M wave instructions for 51 040 pairs of 131 072 bits): a 320-frame scene must become several waves per SIMD.
// ---------------------------------------------------------------------------------------------------------
#ifndef MSPA_K2_MFMA
#define MSPA_K2_MFMA 1
#endif
#ifndef MSPA_K2_DRY
#define MSPA_K2_DRY 0          // timing ablations only (results wrong by construction): 1 = no spreading, no MFMA; 2 = spreading only
#endif
#ifndef MSPA_K2_ACC2
#define MSPA_K2_ACC2 0
#endif
constexpr int kTile = 32;                  // rows of a and of b per workgroup
#ifndef MSPA_K2_CHUNK_WORDS
#define MSPA_K2_CHUNK_WORDS 8
#endif
#ifndef MSPA_K2_WAVE_CHUNKS
#define MSPA_K2_WAVE_CHUNKS 2
#endif
#ifndef MSPA_K2_MIN_WAVES
#define MSPA_K2_MIN_WAVES 5
#endif
constexpr int kChunkWords = MSPA_K2_CHUNK_WORDS;          // bitset words staged per step (8: 5 KB of LDS per wave)
constexpr int kRowDw = kChunkWords * 2 + 4;               // LDS row stride in dwords: 16-byte aligned, bank-skewed
constexpr int kTileWaves = 4;                             // waves per workgroup
constexpr int kWaveChunks = MSPA_K2_WAVE_CHUNKS;          // chunks per wave; the next chunk's loads fly while this one is counted
constexpr int kSliceWords = kChunkWords * kTileWaves * kWaveChunks;
constexpr int kLoadsPerChunk = kChunkWords / 2;           // 16-byte loads per lane and chunk (64 rows x kChunkWords x 8 B / 64 lanes)
constexpr int kLanesPerRow = kChunkWords / 2;             // lanes covering one row's chunk
constexpr int kRowsPerLoad = kWave / kLanesPerRow;        // rows a wave-wide load covers

struct TileArgs {
    const uint64_t *bits_a;
    const uint64_t *bits_b;
    int n_a, n_b;
    int64_t n_words;
    int tiles_b;
    int slices;
    int symmetric;                 // bits_a == bits_b: tiles below the diagonal are skipped
    int32_t *partial;              // [tiles_a * tiles_b][slices][kTile * kTile]
    int32_t *diag;                 // scene form: [slices][n_a] row popcounts of the slice (the diagonal, compact); else null
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// one chunk of the tile's 64 rows: kLoadsPerChunk 16-byte loads per lane (kLanesPerRow lanes cover a row's chunk)
template <bool ALIGNED16>
__device__ __forceinline__ void load_chunk(const TileArgs &a, int ta, int tb, int64_t w0, int lrow, int lcol,
                                           u32x4 (&v)[kLoadsPerChunk]) {
#pragma unroll
    for (int k = 0; k < kLoadsPerChunk; ++k) {
        const int r = k * kRowsPerLoad + lrow;                           // 0..31 = a rows, 32..63 = b rows
        const bool is_b = r >= kTile;
        const int row = is_b ? tb * kTile + (r - kTile) : ta * kTile + r;
        const bool row_ok = row < (is_b ? a.n_b : a.n_a);
        const uint64_t *src = (is_b ? a.bits_b : a.bits_a) + (int64_t)row * a.n_words;
        const int64_t w = w0 + lcol * 2;
        u32x4 x = {0u, 0u, 0u, 0u};
        if (row_ok && w0 < a.n_words) {
            if (ALIGNED16) {
                if (w < a.n_words) x = *reinterpret_cast<const u32x4 *>(src + w);      // n_words even: w + 1 exists too
            } else {
                const uint64_t p = w < a.n_words ? src[w] : 0ull, q = (w + 1) < a.n_words ? src[w + 1] : 0ull;
                x = u32x4{(uint32_t)p, (uint32_t)(p >> 32), (uint32_t)q, (uint32_t)(q >> 32)};
            }
        }
        v[k] = x;
    }
}

__device__ __forceinline__ void count_chunk(uint32_t *my, int lrow, int lcol, int ti, int tj, const u32x4 (&v)[kLoadsPerChunk],
                                            int (&acc)[4][4]) {
#pragma unroll
    for (int k = 0; k < kLoadsPerChunk; ++k)                             // the wave's own region: no workgroup barrier needed
        *reinterpret_cast<u32x4 *>(&my[(k * kRowsPerLoad + lrow) * kRowDw + lcol * 4]) = v[k];
    wave_lds_fence();                                                    // other lanes' rows are read below
#pragma unroll
    for (int s2 = 0; s2 < kChunkWords / 2; ++s2) {
        u32x4 va[4], vb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) va[r] = *reinterpret_cast<const u32x4 *>(&my[(ti + 8 * r) * kRowDw + s2 * 4]);
#pragma unroll
        for (int c = 0; c < 4; ++c) vb[c] = *reinterpret_cast<const u32x4 *>(&my[(kTile + tj + 8 * c) * kRowDw + s2 * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[r][c] += __popc(va[r].x & vb[c].x) + __popc(va[r].y & vb[c].y) + __popc(va[r].z & vb[c].z) +
                             __popc(va[r].w & vb[c].w);
    }
    wave_lds_fence();                                                    // the next chunk overwrites what was just read
}

// Occupancy is the point of the sizes above: a wave issues one VALU instruction every ~5.7 cycles on its own
// (profiles/r02_valu_rates.txt: 5.7 cycles per instruction at one wave per SIMD, 2.2 at eight), so this popcount-bound
// kernel needs many resident waves, i.e. few registers (one staged chunk at a time, the next one in flight) and little LDS.
template <bool ALIGNED16>
__global__ __launch_bounds__(kTileWaves *kWave, MSPA_K2_MIN_WAVES) void overlap_tile_kernel(TileArgs a) {
    // tile list: all (ta, tb) for a rectangle; for a symmetric problem only ta <= tb, enumerated row by row of the triangle
    int ta, tb;
    if (a.symmetric) {
        const int T = a.tiles_b, t = blockIdx.x;
        // largest ta with ta * T - ta (ta - 1) / 2 <= t  (float estimate, then exact fix-up)
        const float disc = (float)(2 * T + 1) * (float)(2 * T + 1) - 8.0f * (float)t;
        ta = (int)(((float)(2 * T + 1) - __builtin_sqrtf(disc > 0.f ? disc : 0.f)) * 0.5f);
        ta = min(max(ta, 0), T - 1);
        while (ta > 0 && ta * T - ta * (ta - 1) / 2 > t) --ta;
        while (ta + 1 < T && (ta + 1) * T - (ta + 1) * ta / 2 <= t) ++ta;
        tb = ta + (t - (ta * T - ta * (ta - 1) / 2));
    } else {
        ta = blockIdx.x / a.tiles_b;
        tb = blockIdx.x - ta * a.tiles_b;
    }
    const int tile = ta * a.tiles_b + tb;
    const int slice = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per wave: 64 staged rows (kRowDw dwords each); the first 4 096 bytes are reused for the 32 x 32 counter table
    constexpr int kRegionDw = (2 * kTile * kRowDw) > (kTile * kTile) ? (2 * kTile * kRowDw) : (kTile * kTile);
    __shared__ __attribute__((aligned(16))) uint32_t lds[kTileWaves][kRegionDw];
    uint32_t *const my = lds[wave];
    const int lrow = lane / kLanesPerRow, lcol = lane % kLanesPerRow;
    const int ti = lane >> 3, tj = lane & 7;          // this lane's rows: a: ti + 8r, b: tj + 8c
    int acc[4][4] = {};
    const int64_t w0 = (int64_t)slice * kSliceWords + (int64_t)wave * (kChunkWords * kWaveChunks);
    u32x4 cur[kLoadsPerChunk], nxt[kLoadsPerChunk];
    load_chunk<ALIGNED16>(a, ta, tb, w0, lrow, lcol, cur);
#pragma unroll
    for (int q = 0; q < kWaveChunks; ++q) {
        if (q + 1 < kWaveChunks) load_chunk<ALIGNED16>(a, ta, tb, w0 + (q + 1) * kChunkWords, lrow, lcol, nxt);
        if (w0 + q * kChunkWords < a.n_words) count_chunk(my, lrow, lcol, ti, tj, cur, acc);      // wave-uniform condition
#pragma unroll
        for (int k = 0; k < kLoadsPerChunk; ++k) cur[k] = nxt[k];
    }
    // the wave's 32 x 32 counters into its region (element (i, j) at i * 32 + j), then 256 threads sum the four tables
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) my[(ti + 8 * r) * kTile + (tj + 8 * c)] = (uint32_t)acc[r][c];
    __syncthreads();
    int32_t *dst = a.partial + ((int64_t)tile * a.slices + slice) * (kTile * kTile);
#pragma unroll
    for (int q = 0; q < (kTile * kTile) / (kTileWaves * kWave); ++q) {
        const int e = q * (kTileWaves * kWave) + (int)threadIdx.x;
        dst[e] = (int32_t)(lds[0][e] + lds[1][e] + lds[2][e] + lds[3][e]);
    }
    if (a.diag && ta == tb && threadIdx.x < kTile && ta * kTile + (int)threadIdx.x < a.n_a) {
        const int e = (int)threadIdx.x * (kTile + 1);
        a.diag[(int64_t)slice * a.n_a + ta * kTile + (int)threadIdx.x] = (int32_t)(lds[0][e] + lds[1][e] + lds[2][e] + lds[3][e]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// MFMA form of the same tile: |a_i & b_j| over a slice IS a 32 x 32 x K matrix product of 0/1 values, and gfx950's
// v_mfma_i32_32x32x32_i8 does 32 768 multiply-adds per issue where v_and + v_bcnt do 2 048.  One wave per (tile, slice):
// lane (r = lane % 32, g = lane / 32) reads 64 bytes of row r of the a-tile and of the b-tile straight from L2 (the two
// halves g of a 128-byte line; no LDS), spreads 16 bits at a time into the 16 int8 of an MFMA operand (v_bfe + v_mul_u32_u24
// + v_and per 4 bits: nibble * 0x204081 & 0x01010101 puts bit k into byte k) and accumulates in int32.  Which k of the
// product a given bit lands on does not matter as long as the a and the b operand agree, and they do: both are built by the
// same function of (g, position in the lane's 64 bytes).  Counts are exact integers -- the result is bit-identical to the
// popcount form.  A workgroup = four waves, each a quarter of the slice.  The partial table keeps the layout the reduction kernels read (element (i, j) at i * 32 + j).
// The expansion (24 VALU issues per MFMA) is what bounds the kernel, not the matrix pipe; it still does the 51 040 pairs
// of a 320-frame scene in a fraction of the popcount kernel's time (profiles/r02_scene_kernels_stats.md).
// ---------------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
constexpr int kMfmaChunkWords = 16;                        // one 128-byte line of a row per chunk: 64 bytes per lane half
static_assert(kSliceWords % kMfmaChunkWords == 0, "a slice is a whole number of 128-byte lines");

template <int HALF>
__device__ __forceinline__ v4i spread16(uint32_t d) {     // 16 bits of d -> 16 bytes of 0 / 1
    v4i o;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        o[q] = (int)(__umul24(__builtin_amdgcn_ubfe(d, 16 * HALF + 4 * q, 4), 0x00204081u) & 0x01010101u);
    return o;
}

// the lane's 64 bytes of one row's line: four 16-byte loads (zero beyond the row count / the bitset's end)
__device__ __forceinline__ void load_line(const uint64_t *__restrict__ row, bool row_ok, int64_t w, int64_t n_words, u32x4 (&v)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32x4 x = {0u, 0u, 0u, 0u};
        if (row_ok && w + 2 * q < n_words) x = *reinterpret_cast<const u32x4 *>(row + w + 2 * q);   // n_words even
        v[q] = x;
    }
}

__global__ __launch_bounds__(kTileWaves *kWave) void overlap_tile_mfma_kernel(TileArgs a) {
    int ta, tb;
    if (a.symmetric) {
        const int T = a.tiles_b, t = blockIdx.x;
        const float disc = (float)(2 * T + 1) * (float)(2 * T + 1) - 8.0f * (float)t;
        ta = (int)(((float)(2 * T + 1) - __builtin_sqrtf(disc > 0.f ? disc : 0.f)) * 0.5f);
        ta = min(max(ta, 0), T - 1);
        while (ta > 0 && ta * T - ta * (ta - 1) / 2 > t) --ta;
        while (ta + 1 < T && (ta + 1) * T - (ta + 1) * ta / 2 <= t) ++ta;
        tb = ta + (t - (ta * T - ta * (ta - 1) / 2));
    } else {
        ta = blockIdx.x / a.tiles_b;
        tb = blockIdx.x - ta * a.tiles_b;
    }
    const int tile = ta * a.tiles_b + tb;
    const int slice = blockIdx.y;
    const int lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_a = ta * kTile + r, row_b = tb * kTile + r;
    const bool ok_a = row_a < a.n_a, ok_b = row_b < a.n_b;
    const uint64_t *__restrict__ pa = a.bits_a + (int64_t)(ok_a ? row_a : 0) * a.n_words;
    const uint64_t *__restrict__ pb = a.bits_b + (int64_t)(ok_b ? row_b : 0) * a.n_words;
    // As in the popcount form the work is cut finely -- a wave takes kWaveLines 128-byte lines of the slice -- because the
    // spreading is VALU work and a SIMD issues it at full rate only with many resident waves.
    constexpr int kWaveLines = kSliceWords / kMfmaChunkWords / kTileWaves;
    static_assert(kWaveLines * kMfmaChunkWords * kTileWaves == kSliceWords, "the slice splits into whole lines per wave");
    const int64_t w0 = (int64_t)slice * kSliceWords + (int64_t)wave * (kWaveLines * kMfmaChunkWords) + g * (kMfmaChunkWords / 2);
    v16i acc = {};
#if MSPA_K2_ACC2
    v16i acc2 = {};
#endif
    u32x4 ca[kWaveLines][4], cb[kWaveLines][4];
#pragma unroll
    for (int c = 0; c < kWaveLines; ++c) {
        load_line(pa, ok_a, w0 + c * kMfmaChunkWords, a.n_words, ca[c]);
        load_line(pb, ok_b, w0 + c * kMfmaChunkWords, a.n_words, cb[c]);
    }
#pragma unroll
    for (int c = 0; c < kWaveLines; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t da[4] = {ca[c][q].x, ca[c][q].y, ca[c][q].z, ca[c][q].w};
            const uint32_t db[4] = {cb[c][q].x, cb[c][q].y, cb[c][q].z, cb[c][q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#if MSPA_K2_DRY == 1
                acc[0] += (int)(da[e] ^ db[e]);
#elif MSPA_K2_DRY == 2
                acc[0] += spread16<0>(da[e])[0] + spread16<0>(db[e])[1] + spread16<1>(da[e])[2] + spread16<1>(db[e])[3] + spread16<0>(da[e])[2] + spread16<0>(db[e])[3] + spread16<1>(da[e])[0] + spread16<1>(db[e])[1];
#elif MSPA_K2_ACC2
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(spread16<0>(da[e]), spread16<0>(db[e]), acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(spread16<1>(da[e]), spread16<1>(db[e]), acc2, 0, 0, 0);
#else
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(spread16<0>(da[e]), spread16<0>(db[e]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(spread16<1>(da[e]), spread16<1>(db[e]), acc, 0, 0, 0);
#endif
            }
        }
#if MSPA_K2_ACC2
    acc += acc2;
#endif
    // accumulator register v of lane (r, g) is element (i = 8 (v / 4) + 4 g + v % 4, j = r) of the 32 x 32 product;
    // the four waves' tables are summed through LDS and the block writes one partial table, as the popcount form does
    __shared__ int32_t lds[kTileWaves][kTile * kTile];
#pragma unroll
    for (int v = 0; v < 16; ++v) lds[wave][(8 * (v / 4) + 4 * g + (v % 4)) * kTile + r] = acc[v];
    __syncthreads();
    int32_t *dst = a.partial + ((int64_t)tile * a.slices + slice) * (kTile * kTile);
#pragma unroll
    for (int q = 0; q < (kTile * kTile) / (kTileWaves * kWave); ++q) {
        const int e = q * (kTileWaves * kWave) + (int)threadIdx.x;
        dst[e] = lds[0][e] + lds[1][e] + lds[2][e] + lds[3][e];
    }
    if (a.diag && ta == tb && threadIdx.x < kTile && ta * kTile + (int)threadIdx.x < a.n_a) {
        const int e = (int)threadIdx.x * (kTile + 1);
        a.diag[(int64_t)slice * a.n_a + ta * kTile + (int)threadIdx.x] = lds[0][e] + lds[1][e] + lds[2][e] + lds[3][e];
    }
}

// sum over the slices of element (i, j); the loads are independent (unrolled), a lane's neighbours read neighbouring ints
__device__ __forceinline__ int reduce_slices(const int32_t *__restrict__ partial, int tiles_b, int slices, int i, int j) {
    const int tile = (i / kTile) * tiles_b + (j / kTile);
    const int32_t *p = partial + (int64_t)tile * slices * (kTile * kTile) + (i % kTile) * kTile + (j % kTile);
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int s = 0;
    for (; s + 4 <= slices; s += 4) {
        s0 += p[(int64_t)(s + 0) * (kTile * kTile)];
        s1 += p[(int64_t)(s + 1) * (kTile * kTile)];
        s2 += p[(int64_t)(s + 2) * (kTile * kTile)];
        s3 += p[(int64_t)(s + 3) * (kTile * kTile)];
    }
    for (; s < slices; ++s) s0 += p[(int64_t)s * (kTile * kTile)];
    return (s0 + s1) + (s2 + s3);
}

// [n_a, n_b] intersection counts; symmetric input: entries below the diagonal are mirrored
__global__ __launch_bounds__(256) void overlap_reduce_kernel(const int32_t *__restrict__ partial, int n_a, int n_b,
                                                             int tiles_b, int slices, int symmetric, int32_t *out) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= n_b) return;
    const bool flip = symmetric && (j / kTile) < (i / kTile);
    out[(int64_t)i * n_b + j] = flip ? reduce_slices(partial, tiles_b, slices, j, i)
                                     : reduce_slices(partial, tiles_b, slices, i, j);
}

// all pairs i < j in the reference's nested-loop order (CFR:176-178): p = i*F - i(i+1)/2 + (j - i - 1).
// The row popcounts |a_i| are the diagonal of the intersection table; the tile kernel leaves them as a compact
// [slices][F] table beside the partial tables, so |a_i| is one wave's sum per block and |a_j| a coalesced sum per thread,
// in the same round trips as |a_i & a_j| (a separate diagonal kernel cost a 5 us launch for 320 numbers).
__global__ __launch_bounds__(256) void scene_overlap_finalize_kernel(const int32_t *__restrict__ partial,
                                                                     const int32_t *__restrict__ diag, int F, int tiles_b,
                                                                     int slices, double *overlap, int32_t *inter_out,
                                                                     int32_t *union_out) {
    const int i = blockIdx.y;
    if ((int)(blockIdx.x * 256 + 255) <= i) return;            // the whole block lies on or below the diagonal (uniform)
    __shared__ int cnt_i;
    if (threadIdx.x < kWave) {
        int sum = 0;
        for (int s = threadIdx.x; s < slices; s += kWave) sum += diag[(int64_t)s * F + i];
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
        if (threadIdx.x == 0) cnt_i = sum;
    }
    const int j = blockIdx.x * 256 + threadIdx.x;
    const bool active = j < F && j > i;
    int inter = 0, cnt_j = 0;
    if (active) {
        inter = reduce_slices(partial, tiles_b, slices, i, j);
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0, s = 0;
        for (; s + 4 <= slices; s += 4) {
            c0 += diag[(int64_t)(s + 0) * F + j];
            c1 += diag[(int64_t)(s + 1) * F + j];
            c2 += diag[(int64_t)(s + 2) * F + j];
            c3 += diag[(int64_t)(s + 3) * F + j];
        }
        for (; s < slices; ++s) c0 += diag[(int64_t)s * F + j];
        cnt_j = (c0 + c1) + (c2 + c3);
    }
    __syncthreads();
    if (!active) return;
    const int uni = cnt_i + cnt_j - inter;
    const int64_t p = (int64_t)i * F - (int64_t)i * (i + 1) / 2 + (j - i - 1);
    overlap[p] = (double)inter / (double)uni * 100.0;          // 0/0 -> NaN as in CFR:136
    if (inter_out) inter_out[p] = inter;
    if (union_out) union_out[p] = uni;
}

struct TilePlan {
    int tiles_a, tiles_b, slices;
    int64_t partial_bytes, bytes;      // partial tables; + the [slices][rows] diagonal table of the scene form
};

static TilePlan plan_tiles(int n_a, int n_b, int64_t n_words) {
    TilePlan p;
    p.tiles_a = (n_a + kTile - 1) / kTile;
    p.tiles_b = (n_b + kTile - 1) / kTile;
    p.slices = (int)((n_words + kSliceWords - 1) / kSliceWords);
    p.partial_bytes = (int64_t)p.tiles_a * p.tiles_b * p.slices * (kTile * kTile) * (int64_t)sizeof(int32_t);
    p.bytes = p.partial_bytes + (int64_t)p.slices * (((n_a > n_b ? n_a : n_b) + 3) / 4 * 4) * (int64_t)sizeof(int32_t);
    return p;
}

static int launch_tiles(const uint64_t *bits_a, int n_a, const uint64_t *bits_b, int n_b, int64_t n_words, bool symmetric,
                        const TilePlan &p, int32_t *partial, int32_t *diag, hipStream_t s) {
    TileArgs t{bits_a, bits_b, n_a, n_b, n_words, p.tiles_b, p.slices, symmetric ? 1 : 0, partial, diag};
    const int64_t n_tiles = symmetric ? (int64_t)p.tiles_a * (p.tiles_a + 1) / 2 : (int64_t)p.tiles_a * p.tiles_b;
    const dim3 grid((uint32_t)n_tiles, (uint32_t)p.slices);
    const bool aligned = (n_words % 2 == 0) && (((uintptr_t)bits_a & 15u) == 0) && (((uintptr_t)bits_b & 15u) == 0);
#if MSPA_K2_MFMA
    if (aligned) {                                   // the matrix-core form needs whole 16-byte pieces of every row
        hipLaunchKernelGGL(overlap_tile_mfma_kernel, grid, dim3(kTileWaves * kWave), 0, s, t);
        return check_hip(hipGetLastError(), "overlap_tile_mfma_kernel launch");
    }
#endif
    if (aligned) hipLaunchKernelGGL(overlap_tile_kernel<true>, grid, dim3(kTileWaves * kWave), 0, s, t);
    else hipLaunchKernelGGL(overlap_tile_kernel<false>, grid, dim3(kTileWaves * kWave), 0, s, t);
    return check_hip(hipGetLastError(), "overlap_tile_kernel launch");
}

}  // namespace mspa

using namespace mspa;

extern "C" int64_t mspa_overlap_workspace_bytes(int32_t n_a, int32_t n_b, int64_t n_words) {
    if (n_a <= 0 || n_b <= 0 || n_words <= 0) return 0;
    return plan_tiles(n_a, n_b, n_words).bytes;
}

extern "C" int mspa_overlap_matrix(const uint64_t *bits_a, int32_t n_a, const uint64_t *bits_b, int32_t n_b,
                                   int64_t n_words, void *workspace, int64_t workspace_bytes, int32_t *out_inter,
                                   mspa_stream_t stream) {
    if (n_a < 0 || n_b < 0 || n_words <= 0) return fail(MSPA_EINVAL, "mspa_overlap_matrix: bad size");
    if (n_a == 0 || n_b == 0) return MSPA_OK;
    if (!bits_a || !bits_b || !out_inter || !workspace) return fail(MSPA_EINVAL, "mspa_overlap_matrix: null pointer");
    if (n_words > (1LL << 25)) return fail(MSPA_EINVAL, "mspa_overlap_matrix: bitset too long for int32 counts");
    const TilePlan p = plan_tiles(n_a, n_b, n_words);
    if ((int64_t)p.tiles_a * p.tiles_b > 0x7fffffffLL || p.slices > 65535 || n_a > 65535)
        return fail(MSPA_EINVAL, "mspa_overlap_matrix: too many rows / words; split the batch");
    if (workspace_bytes < p.bytes) return fail(MSPA_EINVAL, "mspa_overlap_matrix: workspace smaller than mspa_overlap_workspace_bytes()");
    if (((uintptr_t)workspace & 3u) != 0) return fail(MSPA_EINVAL, "mspa_overlap_matrix: workspace must be 4-byte aligned");
    const bool symmetric = (bits_a == bits_b) && (n_a == n_b);
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_tiles(bits_a, n_a, bits_b, n_b, n_words, symmetric, p, (int32_t *)workspace, nullptr, s);
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
    const TilePlan p = plan_tiles(n_images, n_images, n_words);
    if (p.slices > 65535) return fail(MSPA_EINVAL, "mspa_scene_overlap: bitset too long; split the scene");
    if (workspace_bytes < p.bytes) return fail(MSPA_EINVAL, "mspa_scene_overlap: workspace smaller than mspa_overlap_workspace_bytes()");
    if (((uintptr_t)workspace & 3u) != 0) return fail(MSPA_EINVAL, "mspa_scene_overlap: workspace must be 4-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int32_t *partial = (int32_t *)workspace;
    int32_t *diag = (int32_t *)((char *)workspace + p.partial_bytes);
    int rc = launch_tiles(bits, n_images, bits, n_images, n_words, true, p, partial, diag, s);
    if (rc) return rc;
    hipLaunchKernelGGL(scene_overlap_finalize_kernel, dim3((uint32_t)((n_images + 255) / 256), (uint32_t)n_images), dim3(256),
                       0, s, (const int32_t *)partial, (const int32_t *)diag, n_images, p.tiles_b, p.slices, out_overlap, out_inter,
                       out_union);
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
