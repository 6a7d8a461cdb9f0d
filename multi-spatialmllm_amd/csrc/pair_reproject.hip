// K3: frame-pair back-projection -> reprojection -> depth-buffer visibility (see include/mspa.h).
//
// Common to all four kernels.  Blocks of one pair are numbered so that they land on one XCD (the
// hardware dispatches block b to XCD b % 8): the frame-2 depth image the pair gathers from (600 KB)
// then stays in that XCD's 4 MB L2.  Camera matrices are read through wave-uniform addresses, i.e.
// as scalar loads into SGPRs -- the one operand v_fma_f64 takes for free -- while the per-pixel
// chain lives in VGPRs.  No MFMA: this is point geometry, bound by HBM and VALU issue.
//
// Four kernels behind one entry point (mspa_pair_correspondences adds the compacted output set of the third):
//   * pair_exact_kernel -- the reference's own operation order (five sequential 3x4 products, IEEE
//     division); a workgroup owns a strip of 4096 consecutive pixels, lanes take consecutive pixels.
//     Float64 outputs are bit-identical to the C oracle.  ~100 FP64 VALU ops / pixel: issue bound.
//   * pair_fast_kernel (MSPA_PAIR_FAST, any image shape) -- one composed 3x4 product + reciprocal;
//     lane <-> image column, a wave walks down a 64 x 16 tile.  Every lane whose integer decisions
//     (half-to-even rounding of the pixel index, the image bounds, the strict depth comparison) sit
//     within a guard band of a decision boundary is re-evaluated with the exact chain, so masks, pixel
//     indices and counters stay bit-exact; the band is derived per tile from a BOUND on the difference between
//     the two evaluation orders (mspa_common.h guard_from_bounds, slot MSPA_MAT_BOUNDS).
//   * pair_fast_tight_kernel (MSPA_PAIR_FAST, whole-tile images such as 640x480) -- the benchmarked
//     one: same arithmetic, plus LDS-DMA depth tiles, tile- and group-level culling, buffer-resource
//     addressing and 16-byte stores (see the comment above it).
//   * pair_fast_scaled_kernel (MSPA_PAIR_FAST, ScanNet's 1296 x 968 colour grid over 640 x 480 depth) -- the same machinery
//     on stripes that follow the bitset's word boundaries, both grid scalings off the per-pixel path.
#include "mspa_common.h"

#include <type_traits>

namespace mspa {

// Inputs are separate `const T *__restrict__` kernel parameters (not struct members) on purpose:
// only then can the compiler prove the wave-uniform matrix/pair reads are never clobbered by the
// kernel's own stores and issue them as scalar loads (s_load_dwordx8/x16 -> SGPRs).
struct PairArgs {
    int64_t n_pairs;
    int dh, dw, H, W;
    uint32_t P;            // H*W
    uint32_t div_magic;    // floor(2^32 / W) + 1 : i / W == umulhi(i, magic) for i*W < 2^32
    double sx, sy;         // dw / W, dh / H  (IH:359-360, OPS:272-273)
    int strips;            // exact kernel: 4096-pixel strips per pair;  fast kernel: groups of 4 wave tiles
    int n_stripes, n_tiles;   // fast kernel: 64-column stripes per row band, wave tiles per pair
    int n_wave_tiles;         // tight kernel: tile GROUPS per pair (a wave walks TPW vertically adjacent tiles); == n_tiles when TPW == 1
    uint32_t stripe_magic;    // floor(2^32 / n_stripes) + 1
    uint64_t *vis_bits;
    uint8_t *vis_u8;
    uint8_t *valid_u8;
    int16_t *pix_i16;
    float *xyz_f32;
    uint32_t *rgba;
    double *xyz_f64;
    double *uv_f64;
    double *depth_f64;
    int32_t *counts;
    int16_t *cpix;            // compacted correspondences: [n_pairs][n_tiles][kTileCap][2] (fused tight kernel only)
    int32_t *tile_counts;     // [n_pairs][n_tiles] entries per tile segment
    uint32_t xcd_shift;       // log2 of the XCDs workgroups are dealt over (3 on an MI355X in SPX mode, 0 otherwise)
    double wm1, hm1, wh_max;  // W - 1, H - 1, max(W, H) as float64 (guard band: mspa_common.h guard_from_bounds)
    // tight kernel: u, v are taken relative to the centre of the grid they live on (the depth grid), so that "inside the image
    // up to the guard" is ONE compare per axis: |u - W/2| < W/2 + g
    double hw, hh, khw, khh;  // dw / 2, dh / 2 (integers), and the same + kGuardPx
    int hwi, hhi;
};

// Output sets.  A kernel instantiated with GENERIC = true tests every output pointer at run time
// (wave-uniform branches); the specialised instances know their set at compile time, which frees
// ~20 SGPRs of pointers and removes the dead stores' address arithmetic.
enum : uint32_t {
    O_VIS_BITS = 1u << 0, O_VIS_U8 = 1u << 1, O_VALID_U8 = 1u << 2, O_PIX = 1u << 3, O_XYZ32 = 1u << 4,
    O_RGBA = 1u << 5, O_XYZ64 = 1u << 6, O_UV64 = 1u << 7, O_DEPTH64 = 1u << 8, O_COUNTS = 1u << 9, O_CPIX = 1u << 10,
};
constexpr uint32_t kSetCorr = O_VIS_BITS | O_PIX | O_COUNTS;                         // correspondence
constexpr uint32_t kSetDense = O_VIS_U8 | O_PIX | O_XYZ32 | O_RGBA | O_COUNTS;       // coloured point cloud
constexpr uint32_t kSetDenseXyz = O_VIS_U8 | O_PIX | O_XYZ32 | O_COUNTS;              // point cloud without colour (rgb = 0)
constexpr uint32_t kSetMinimal = O_VIS_BITS | O_COUNTS;                              // overlap only
constexpr uint32_t kSetCompact = O_VIS_BITS | O_CPIX | O_COUNTS;                     // correspondences of the visible pixels only

template <uint32_t SET, bool GENERIC>
struct Outs {
    template <uint32_t BIT, typename T>
    static __device__ __forceinline__ bool has(T *ptr) {
        return GENERIC ? (ptr != nullptr) : ((SET & BIT) != 0);
    }
};

constexpr int kThreads = 256;
constexpr int kIters = 16;                       // 4096 pixels per workgroup
constexpr int kStrip = kThreads * kIters;
constexpr int kGroup = 4;                        // pixels per lane processed between matrix reloads

// guard band of the fast kernels: kGuardPx and guard_from_bounds in mspa_common.h

// Makes a wave-uniform pointer opaque to LICM: loads through the result cannot be hoisted out of
// the pixel loop, so at most one stage's matrices are live in SGPRs at a time (all five would
// need 120 SGPRs; beyond ~100 the compiler spills them to VGPR lanes and every FMA pays a
// v_readlane).
// (The asm launders an SGPR *offset*, not the pointer: the pointer keeps its kernel-argument
// provenance -- global address space, noalias -- so the loads stay scalar.)
// `after` is a value the previous stage produced: the (empty) asm cannot issue before it exists,
// which pins the matrix loads between the two stages instead of at the top of the loop body.
__device__ __forceinline__ const double *opaque(const double *p, double after) {
    int zero = 0;
    asm volatile("" : "+s"(zero) : "v"(after));
    return p + zero;
}

struct Pixel {
    double ax, ay, az;     // aligned world point
    double u, v, qz;       // projection into frame 2, camera-2 depth
    int xi, yi;            // clipped depth-2 pixel index
    bool vis;
    bool inview;           // inside frame 2 and in front of camera 2 (before the depth-buffer test)
};

// The exact chain (OPS:303-320, IH:57-69), one 3x4 product at a time over a group of kGroup pixels
// so that only ONE matrix (24 SGPRs) has to be live at any point of the loop.
template <int G>
__device__ __forceinline__ void apply_affine(const double *m, double (&x)[G], double (&y)[G], double (&z)[G]) {
    // sched_barrier: the machine scheduler must not interleave two products, or it keeps several
    // matrices live and spills SGPRs to VGPR lanes (v_writelane/v_readlane on every operand).
    __builtin_amdgcn_sched_barrier(0);
    const double *__restrict__ mm = opaque(m, x[G - 1]);
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const double nx = affine_row(mm + 0, x[j], y[j], z[j]);
        const double ny = affine_row(mm + 4, x[j], y[j], z[j]);
        const double nz = affine_row(mm + 8, x[j], y[j], z[j]);
        x[j] = nx;
        y[j] = ny;
        z[j] = nz;
    }
    // pin the results here: without this the optimiser sinks the products of pixels 1..G-1 past the
    // next stages (down to their first use), which again keeps every matrix live across the body
#pragma unroll
    for (int j = 0; j < G; ++j) asm volatile("" : "+v"(x[j]), "+v"(y[j]), "+v"(z[j]));
    __builtin_amdgcn_sched_barrier(0);
}

// single-pixel forms (cold path of the fast kernel)
__device__ __forceinline__ void exact_unproject(const double *__restrict__ m1, double mxd, double myd, double d,
                                                double &ax, double &ay, double &az) {
    double x[1] = {mxd * d}, y[1] = {myd * d}, z[1] = {d};
    apply_affine<1>(m1 + MSPA_MAT_KINV * 16, x, y, z);      // OPS:313
    apply_affine<1>(m1 + MSPA_MAT_E * 16, x, y, z);         // OPS:316
    apply_affine<1>(m1 + MSPA_MAT_A * 16, x, y, z);         // OPS:319-320
    ax = x[0];
    ay = y[0];
    az = z[0];
}

__device__ __forceinline__ void exact_project(const double *__restrict__ m2, double ax, double ay, double az,
                                              double &u, double &v, double &qz) {
    double x[1] = {ax}, y[1] = {ay}, z[1] = {az};
    apply_affine<1>(m2 + MSPA_MAT_EINV_ALIGNED * 16, x, y, z);   // IH:57-60
    qz = z[0];                                                    // IH:63
    apply_affine<1>(m2 + MSPA_MAT_K * 16, x, y, z);              // IH:66
    u = x[0] / z[0];                                              // IH:69
    v = y[0] / z[0];
}

struct Ctx {
    const uint16_t *__restrict__ depth1;
    const uint16_t *__restrict__ depth2;
    const uint8_t *__restrict__ rgb1;
    int64_t obase;
    int64_t words_per_pair;
    int64_t pair;
    int lane;
};

// All per-pixel stores.  Every branch on an output pointer is wave-uniform (or compile-time).
template <typename O, bool SKIP_BITS = false>
__device__ __forceinline__ void store_pixel(const PairArgs &a, const Ctx &c, uint32_t i, bool in_img, bool valid,
                                            const Pixel &p) {
    if (!SKIP_BITS && O::template has<O_VIS_BITS>(a.vis_bits)) {
        const unsigned long long vmask = __ballot(p.vis);
        if (c.lane == 0 && (i - c.lane) < a.P) a.vis_bits[c.pair * c.words_per_pair + ((i - c.lane) >> 6)] = vmask;
    }
    if (!in_img) return;
    const int64_t o = c.obase + i;
    if (O::template has<O_VIS_U8>(a.vis_u8)) a.vis_u8[o] = p.vis ? 1 : 0;
    if (O::template has<O_VALID_U8>(a.valid_u8)) a.valid_u8[o] = valid ? 1 : 0;
    if (O::template has<O_PIX>(a.pix_i16)) {
        const uint32_t packed =
            (valid && p.inview) ? ((uint32_t)(uint16_t)p.xi | ((uint32_t)(uint16_t)p.yi << 16)) : 0xFFFFFFFFu;
        reinterpret_cast<uint32_t *>(a.pix_i16)[o] = packed;
    }
    if (O::template has<O_XYZ32>(a.xyz_f32)) {
        float *q = a.xyz_f32 + 3 * o;
        const float fn = __builtin_nanf("");
        q[0] = valid ? (float)p.ax : fn;
        q[1] = valid ? (float)p.ay : fn;
        q[2] = valid ? (float)p.az : fn;
    }
    if (O::template has<O_RGBA>(a.rgba)) {
        uint32_t col = 0;
        if (c.rgb1) {
            const uint8_t *s = c.rgb1 + 3 * (int64_t)i;
            col = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
        }
        a.rgba[o] = col | (valid ? 0xFF000000u : 0u);
    }
    const double nan = __builtin_nan("");
    if (O::template has<O_XYZ64>(a.xyz_f64)) {
        double *q = a.xyz_f64 + 3 * o;
        q[0] = valid ? p.ax : nan;
        q[1] = valid ? p.ay : nan;
        q[2] = valid ? p.az : nan;
    }
    if (O::template has<O_UV64>(a.uv_f64)) {
        a.uv_f64[2 * o + 0] = valid ? p.u : nan;
        a.uv_f64[2 * o + 1] = valid ? p.v : nan;
    }
    if (O::template has<O_DEPTH64>(a.depth_f64)) a.depth_f64[o] = valid ? p.qz : nan;
}

// wave reduce (bpermute shuffles), one LDS step, two atomics per workgroup
template <typename O>
__device__ __forceinline__ void flush_counts(const PairArgs &a, int64_t pair, int lane, int n_valid, int n_vis) {
    if (!O::template has<O_COUNTS>(a.counts)) return;
    for (int off = 32; off > 0; off >>= 1) {
        n_valid += __shfl_down(n_valid, off);
        n_vis += __shfl_down(n_vis, off);
    }
    __shared__ int red[2][kThreads / kWave];
    const int w = threadIdx.x >> 6;
    if (lane == 0) {
        red[0][w] = n_valid;
        red[1][w] = n_vis;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int sv = 0, ss = 0;
        for (int j = 0; j < kThreads / kWave; ++j) {
            sv += red[0][j];
            ss += red[1][j];
        }
        atomicAdd(a.counts + 2 * pair + 0, sv);
        atomicAdd(a.counts + 2 * pair + 1, ss);
    }
}

// XCD-aware decode: the hardware deals workgroup b to XCD b % n_xcd (n_xcd = 1 << xcd_shift: 8 on an MI355X in SPX mode; the
// host passes shift 0 for partitioned devices and other parts, which is the plain linear decode).  xcd picks the pair within
// a group of n_xcd pairs, the rest walks the strips: all workgroups of one pair share one L2.  Results never depend on it.
__device__ __forceinline__ bool decode_block(const PairArgs &a, int64_t &pair, uint32_t &strip) {
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & ((1u << a.xcd_shift) - 1u);
    const uint32_t k = b >> a.xcd_shift;
    strip = k % (uint32_t)a.strips;
    pair = ((int64_t)(k / (uint32_t)a.strips) << a.xcd_shift) + xcd;
    return pair < a.n_pairs;
}

// 16-byte buffer store + guard.  Observed on gfx950: when a VALU instruction writes one of the store's four data VGPRs
// right behind the store, lanes 12..15 of every 16 store the NEW value -- the store is still reading its data.  LLVM's
// hazard recognizer inserts the wait state for > 64-bit stores only when soffset is an immediate; with an SGPR soffset it
// assumes none is needed.  The guard is an s_nop that NAMES the data registers as an input: that keeps them live (nothing
// else can be allocated to them) until the wait states have passed, and being a volatile asm with a memory clobber it
// stays behind the store.  (Round 2's guard was a bare `s_nop 1`: the scheduler was free to slip a v_cndmask whose result
// had been given one of the data registers between the store and the nop, and in round 3 it did -- the dense set's rgba
// words came out as the next payload's float bits in exactly those lanes; tests/test_gpu_tight.py, 72 pairs at 640x480.)
// The store carries the non-temporal hint (aux 2 = nt): the outputs are full 128-byte lines that this launch never reads
// back; measured -2.5..-3 % on the headline kernel, with distinct as well as with heavily re-used input frames.
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buffer_store_b128_guarded(u32x4_t q, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(q, rs, voff, soff, 2);
    asm volatile("s_nop 1" ::"v"(q) : "memory");
}

// depth sample of frame 1 under colour pixel (mx, my): OPS:272-294
template <bool IDENT>
__device__ __forceinline__ uint32_t sample_depth1(const PairArgs &a, const uint16_t *__restrict__ depth1, uint32_t ic,
                                                  uint32_t mx, uint32_t my) {
    if (IDENT) return depth1[ic];
    const int dy = round_clip((double)my * a.sy, a.dh - 1);
    const int dx = round_clip((double)mx * a.sx, a.dw - 1);
    return depth1[dy * a.dw + dx];
}

template <bool IDENT>
__global__ __launch_bounds__(kThreads) void pair_exact_kernel(const uint16_t *__restrict__ depth,
                                                              const uint8_t *__restrict__ rgb,
                                                              const double *__restrict__ mats,
                                                              const int32_t *__restrict__ pairs, PairArgs a) {
    int64_t pair;
    uint32_t strip;
    if (!decode_block(a, pair, strip)) return;
    const int f1 = pairs[2 * pair + 0];
    const int f2 = pairs[2 * pair + 1];
    const double *m1 = mats + (int64_t)f1 * (MSPA_FRAME_MATS * 16);
    const double *m2 = mats + (int64_t)f2 * (MSPA_FRAME_MATS * 16);
    const int64_t dpix = (int64_t)a.dh * a.dw;
    Ctx c;
    c.depth1 = depth + (int64_t)f1 * dpix;
    c.depth2 = depth + (int64_t)f2 * dpix;
    c.rgb1 = rgb ? rgb + (int64_t)f1 * a.P * 3 : nullptr;
    c.obase = pair * (int64_t)a.P;
    c.words_per_pair = (a.P + 63) >> 6;
    c.pair = pair;
    c.lane = threadIdx.x & 63;

    int n_valid = 0, n_vis = 0;
    const uint32_t i0 = strip * (uint32_t)kStrip + threadIdx.x;
    for (int g = 0; g < kIters; g += kGroup) {
        double x[kGroup], y[kGroup], z[kGroup];
        bool in_img[kGroup], valid[kGroup];
        Pixel px[kGroup];
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            const uint32_t i = i0 + (uint32_t)(g + j) * kThreads;
            in_img[j] = i < a.P;
            const uint32_t ic = in_img[j] ? i : a.P - 1;
            const uint32_t my = __umulhi(ic, a.div_magic);
            const uint32_t mx = ic - my * (uint32_t)a.W;
            const uint32_t d16 = sample_depth1<IDENT>(a, c.depth1, ic, mx, my);
            const double d = (double)d16 * 0.001;                     // OPS:292-294
            valid[j] = in_img[j] && (d > 0.0);                        // OPS:297
            x[j] = (double)mx * d;                                    // OPS:303-310
            y[j] = (double)my * d;
            z[j] = d;
        }
        apply_affine<kGroup>(m1 + MSPA_MAT_KINV * 16, x, y, z);      // OPS:313
        apply_affine<kGroup>(m1 + MSPA_MAT_E * 16, x, y, z);         // OPS:316
        apply_affine<kGroup>(m1 + MSPA_MAT_A * 16, x, y, z);         // OPS:319-320
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            px[j].ax = x[j];
            px[j].ay = y[j];
            px[j].az = z[j];
        }
        apply_affine<kGroup>(m2 + MSPA_MAT_EINV_ALIGNED * 16, x, y, z);   // IH:57-60
#pragma unroll
        for (int j = 0; j < kGroup; ++j) px[j].qz = z[j];                  // IH:63
        apply_affine<kGroup>(m2 + MSPA_MAT_K * 16, x, y, z);              // IH:66
#pragma unroll
        for (int j = 0; j < kGroup; ++j) {
            px[j].u = x[j] / z[j];                                         // IH:69
            px[j].v = y[j] / z[j];
            px[j].vis = depth_test(valid[j], px[j].u, px[j].v, px[j].qz, c.depth2, a.dh, a.dw, a.H, a.W, a.sx, a.sy,
                                   px[j].xi, px[j].yi, &px[j].inview);
            n_valid += valid[j] ? 1 : 0;
            n_vis += px[j].vis ? 1 : 0;
            store_pixel<Outs<0, true>>(a, c, i0 + (uint32_t)(g + j) * kThreads, in_img[j], valid[j], px[j]);
        }
    }
    flush_counts<Outs<0, true>>(a, pair, c.lane, n_valid, n_vis);
}

// --------------------------------------------------------------------------------------------
// fast path
// --------------------------------------------------------------------------------------------
// row r of (3x4 affine X) * (3x4 affine Y), both row-major 4x4 storage; plain float64 FMAs
__device__ __forceinline__ void compose_row(const double *__restrict__ X, const double *__restrict__ Y, int r,
                                            double out[4]) {
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx) {
        double acc = X[4 * r + 0] * Y[0 + cidx];
        acc = __builtin_fma(X[4 * r + 1], Y[4 + cidx], acc);
        acc = __builtin_fma(X[4 * r + 2], Y[8 + cidx], acc);
        if (cidx == 3) acc += X[4 * r + 3];
        out[cidx] = acc;
    }
}

__device__ __forceinline__ double uniform(double v) {   // VGPR holding a wave-uniform value -> SGPR pair
    const unsigned long long b = __double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

constexpr int kTileRows = 16;                    // a wave owns a 64-column x 16-row tile
constexpr int kRowGroup = 4;                     // rows whose depth-2 gathers are in flight together

// Fast kernel.
//  * Mapping: lane <-> image column, the wave walks down kTileRows rows.  The column is loop-invariant
//    per lane and the row is wave-uniform, so the pixel-coordinate half of the composed product
//    (M[:,0]*mx + M[:,2]) is computed once per lane, the row half is one add per row, and no integer
//    division is needed.
//  * Memory-level parallelism: all kTileRows depth-1 samples of the tile are requested before any
//    arithmetic (one 128-byte line per wave-row, kTileRows lines in flight per wave); depth-2 gathers
//    are issued kRowGroup rows at a time.  With one row at a time a wave pays two dependent memory
//    round trips per row and the kernel sits at ~30 % of HBM bandwidth (profiles/r01).
//  * The hot loop is branch-free.  Lanes whose decisions fall inside the guard band only set a bit in
//    a per-lane row mask; one cold loop after the tile re-evaluates those pixels with the exact chain
//    and patches the outputs (pixel index store, bit flip by atomicXor, counter delta).
//  * TIGHT = the image is a whole number of tiles (W % 64 == 0, H % kTileRows == 0, true for 640x480):
//    no edge predicates, bitset words coincide with wave rows.
template <bool IDENT, bool TIGHT, uint32_t SET, bool GENERIC, bool LINEAR = false>
__global__ __launch_bounds__(kThreads) void pair_fast_kernel(const uint16_t *__restrict__ depth,
                                                             const uint8_t *__restrict__ rgb,
                                                             const double *__restrict__ mats,
                                                             const int32_t *__restrict__ pairs, PairArgs a) {
    using O = Outs<SET, GENERIC>;
    constexpr bool WANT_XYZ = GENERIC || (SET & O_XYZ32);
    int64_t pair;
    uint32_t tgroup;
    if (!decode_block(a, pair, tgroup)) return;
    const int f1 = pairs[2 * pair + 0];
    const int f2 = pairs[2 * pair + 1];
    const double *m1 = mats + (int64_t)f1 * (MSPA_FRAME_MATS * 16);
    const double *m2 = mats + (int64_t)f2 * (MSPA_FRAME_MATS * 16);
    const int64_t dpix = (int64_t)a.dh * a.dw;
    Ctx c;
    c.depth1 = depth + (int64_t)f1 * dpix;
    c.depth2 = depth + (int64_t)f2 * dpix;
    c.rgb1 = rgb ? rgb + (int64_t)f1 * a.P * 3 : nullptr;
    c.obase = pair * (int64_t)a.P;
    c.words_per_pair = (a.P + 63) >> 6;
    c.pair = pair;
    c.lane = threadIdx.x & 63;

    // Per-pair composed matrices: U = A*E1*Kinv (host, per frame), M = (K*inv(A*E2)) * U.
    // 36 FMAs per wave, once per tile; results parked in SGPRs.  The depth scale 0.001 (OPS:292) is
    // folded into the columns that multiply the raw millimetre sample.
    const double *__restrict__ U = m1 + MSPA_MAT_UNPROJ * 16;
    const double *__restrict__ N = m2 + MSPA_MAT_REPROJ * 16;
    double M[3][4], Us[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double row[4];
        compose_row(N, U, r, row);
#pragma unroll
        for (int k = 0; k < 4; ++k) M[r][k] = uniform(k < 3 ? row[k] * 0.001 : row[k]);
        if (WANT_XYZ) {
#pragma unroll
            for (int k = 0; k < 4; ++k) Us[r][k] = k < 3 ? U[4 * r + k] * 0.001 : U[4 * r + k];
        }
    }

    // guard band from the pair's bound coefficients, over the whole image and the full sample range (this kernel serves the
    // shapes without a tile pre-pass); its composed matrix works in metres
    const Guard gd = guard_from_bounds(m1 + MSPA_MAT_BOUNDS * 16, m2 + MSPA_MAT_BOUNDS * 16, a.wm1, a.hm1, 65535.0, a.wh_max);
    const double zmin_m = uniform(gd.zmin * 0.001), gz_m = uniform(gd.gz * 0.001);
    const double tnear_m = uniform(2.0 * a.wh_max * gd.zmin * 0.001);   // near the plane AND near the optical axis (tight kernel)

    // wave tile -> (row band, column stripe); both wave-uniform
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = tgroup * (kThreads / kWave) + wave;
    const uint32_t band = a.n_stripes == 1 ? tile : __umulhi(tile, a.stripe_magic);   // magic overflows for 1
    const uint32_t stripe = tile - band * (uint32_t)a.n_stripes;
    const bool tile_ok = tile < (uint32_t)a.n_tiles;
    const uint32_t col = stripe * 64u + (uint32_t)c.lane;
    const bool col_ok = TIGHT ? true : (tile_ok && col < (uint32_t)a.W);
    const uint32_t colc = col_ok ? col : 0u;
    const uint32_t row0 = band * (uint32_t)kTileRows;
    const bool words_aligned = TIGHT || LINEAR || (a.W & 63) == 0;
    // LINEAR mapping (images whose width is not a multiple of 64, when a bitset is wanted): the tile is kTileRows runs
    // of 64 CONSECUTIVE pixel indices, so that a wave's ballot is exactly one word of the bitset whatever the width;
    // (row, column) come from a magic division per lane and the affine terms are evaluated per pixel, not advanced per row.
    const uint32_t lin_base = tile * (uint32_t)(kTileRows * 64);
    auto lin_pixel = [&](int g, uint32_t &i, uint32_t &row, uint32_t &colx) -> bool {
        const uint32_t raw = lin_base + (uint32_t)g * 64u + (uint32_t)c.lane;
        const bool ok = tile_ok && raw < (uint32_t)a.P;
        i = ok ? raw : 0u;
        row = __umulhi(i, a.div_magic);
        colx = i - row * (uint32_t)a.W;
        return ok;
    };

    int n_valid = 0, n_vis = 0;          // wave totals, kept uniform (SALU popcounts of ballots)
    if (tile_ok) {
        int dx1 = 0;
        if (!IDENT) dx1 = round_clip((double)colc * a.sx, a.dw - 1);   // OPS:286-290, column part
        // depth-1 sample of (tile row g, this lane's column); rows past the image are clamped
        auto load_d1 = [&](int g) -> uint32_t {
            if (LINEAR) {
                uint32_t i, row, colx;
                lin_pixel(g, i, row, colx);
                if (IDENT) return c.depth1[i];
                return c.depth1[round_clip((double)row * a.sy, a.dh - 1) * a.dw + round_clip((double)colx * a.sx, a.dw - 1)];
            }
            const uint32_t row = TIGHT ? row0 + (uint32_t)g : min(row0 + (uint32_t)g, (uint32_t)a.H - 1u);
            if (IDENT) return c.depth1[row * (uint32_t)a.W + colc];
            const int dy = round_clip((double)row * a.sy, a.dh - 1);
            return c.depth1[dy * a.dw + dx1];
        };
        uint32_t d16n[kRowGroup];                   // samples of the NEXT row group (software prefetch)
#pragma unroll
        for (int j = 0; j < kRowGroup; ++j) d16n[j] = load_d1(j);
        const double mxd = (double)colc;
        const double myd0 = (double)row0;
        // t_r = M[r][0]*mx + M[r][1]*my + M[r][2]; advanced by M[r][1] per row
        double t0 = __builtin_fma(M[0][1], myd0, __builtin_fma(M[0][0], mxd, M[0][2]));
        double t1 = __builtin_fma(M[1][1], myd0, __builtin_fma(M[1][0], mxd, M[1][2]));
        double t2 = __builtin_fma(M[2][1], myd0, __builtin_fma(M[2][0], mxd, M[2][2]));
        double s0 = 0, s1 = 0, s2 = 0;
        if (WANT_XYZ) {
            s0 = __builtin_fma(Us[0][1], myd0, __builtin_fma(Us[0][0], mxd, Us[0][2]));
            s1 = __builtin_fma(Us[1][1], myd0, __builtin_fma(Us[1][0], mxd, Us[1][2]));
            s2 = __builtin_fma(Us[2][1], myd0, __builtin_fma(Us[2][0], mxd, Us[2][2]));
        }
        const double Wd = (double)a.W, Hd = (double)a.H;
        uint32_t risky_rows = 0, vis_rows = 0;       // per-lane row masks for the cold loop

#pragma unroll 1
        for (int r0 = 0; r0 < kTileRows; r0 += kRowGroup) {
            uint32_t d16[kRowGroup];
#pragma unroll
            for (int j = 0; j < kRowGroup; ++j) d16[j] = d16n[j];
            if (r0 + kRowGroup < kTileRows) {       // wave-uniform: request the next group's samples now
#pragma unroll
                for (int j = 0; j < kRowGroup; ++j) d16n[j] = load_d1(r0 + kRowGroup + j);
            }
            double qz[kRowGroup];
            float fx[kRowGroup], fy[kRowGroup], fz[kRowGroup];
            int pix[kRowGroup];
            uint32_t dv16[kRowGroup];
            bool valid[kRowGroup], test[kRowGroup], risky[kRowGroup];
            // ---- projection + gather issue ---------------------------------------------------
#pragma unroll
            for (int j = 0; j < kRowGroup; ++j) {
                const int g = r0 + j;
                bool in_img = TIGHT ? true : (col_ok && (row0 + (uint32_t)g) < (uint32_t)a.H);
                if (LINEAR) {
                    uint32_t i, row, colx;
                    in_img = lin_pixel(g, i, row, colx);
                    const double mxl = (double)colx, myl = (double)row;
                    t0 = __builtin_fma(M[0][1], myl, __builtin_fma(M[0][0], mxl, M[0][2]));
                    t1 = __builtin_fma(M[1][1], myl, __builtin_fma(M[1][0], mxl, M[1][2]));
                    t2 = __builtin_fma(M[2][1], myl, __builtin_fma(M[2][0], mxl, M[2][2]));
                    if (WANT_XYZ) {
                        s0 = __builtin_fma(Us[0][1], myl, __builtin_fma(Us[0][0], mxl, Us[0][2]));
                        s1 = __builtin_fma(Us[1][1], myl, __builtin_fma(Us[1][0], mxl, Us[1][2]));
                        s2 = __builtin_fma(Us[2][1], myl, __builtin_fma(Us[2][0], mxl, Us[2][2]));
                    }
                }
                valid[j] = in_img & (d16[j] != 0u);                          // OPS:297
                const double dmm = (double)d16[j];
                // (ix, iy, iz) = M * (mx*d, my*d, d, 1) = d * (M[:, :3] * (mx, my, 1)) + M[:, 3]
                const double ix = __builtin_fma(t0, dmm, M[0][3]);
                const double iy = __builtin_fma(t1, dmm, M[1][3]);
                const double iz = __builtin_fma(t2, dmm, M[2][3]);          // camera-2 depth
                if (WANT_XYZ) {
                    fx[j] = (float)__builtin_fma(s0, dmm, Us[0][3]);
                    fy[j] = (float)__builtin_fma(s1, dmm, Us[1][3]);
                    fz[j] = (float)__builtin_fma(s2, dmm, Us[2][3]);
                    s0 += Us[0][1];
                    s1 += Us[1][1];
                    s2 += Us[2][1];
                }
                t0 += M[0][1];
                t1 += M[1][1];
                t2 += M[2][1];
                // reciprocal: hardware estimate + one Newton step (squares the relative error; the
                // guard band only needs ~1e-9 relative)
                double rz = __builtin_amdgcn_rcp(iz);
                rz = __builtin_fma(__builtin_fma(-iz, rz, 1.0), rz, rz);
                const double u = ix * rz, v = iy * rz;
                const double us = IDENT ? u : u * a.sx;
                const double vs = IDENT ? v : v * a.sy;
                const double ru = __builtin_rint(us), rv = __builtin_rint(vs);
                const bool inb = (u >= 0.0) & (u < Wd) & (v >= 0.0) & (v < Hd);
                // saturating conversion, then an integer clamp (NaN -> 0; overflow is caught by the guard)
                const int xi = min(max((int)ru, 0), a.dw - 1);
                const int yi = min(max((int)rv, 0), a.dh - 1);
                test[j] = valid[j] & inb & (iz > 0.0);           // bitwise: no short-circuit branches
                // unconditional gather (pixel 0 for lanes that cannot pass) keeps the code straight-line
                dv16[j] = c.depth2[test[j] ? (yi * a.dw + xi) : 0];
                pix[j] = (int)((uint32_t)xi | ((uint32_t)yi << 16));
                qz[j] = iz;
                // Guard band, coordinate part.  With t = us - rint(us) in [-0.5, 0.5], a decision can
                // flip only if |t| is within the guard of 0.5 (rounding tie) or of 0 (u at an integer:
                // the image bounds 0 and W are integers), i.e. unless guard < |t| < 0.5 - guard  <=>
                // ||t| - 0.25| < 0.25 - guard.  Huge |us| gives t = 0 and NaN fails the ordered
                // compare, so both end up "risky".
                const double wu = __builtin_fabs(us - ru) - 0.25;
                const double wv = __builtin_fabs(vs - rv) - 0.25;
                bool rk = !(__builtin_fmax(__builtin_fabs(wu), __builtin_fabs(wv)) < 0.25 - kGuardPx);
                // below zmin the projection itself is not trusted; only a lane that is also within tnear of the optical axis
                // (homogeneous x, y) can be accepted by any evaluation order
                rk = rk | (!(__builtin_fabs(iz) > zmin_m) & (__builtin_fabs(ix) < tnear_m) & (__builtin_fabs(iy) < tnear_m));
                if (!IDENT) {   // bounds live in colour-pixel units here, not at integers of the depth grid
                    const double bu = __builtin_fmin(__builtin_fabs(u), __builtin_fabs(u - Wd));
                    const double bv = __builtin_fmin(__builtin_fabs(v), __builtin_fabs(v - Hd));
                    rk = rk | !(__builtin_fmin(bu, bv) > kGuardPx);
                }
                risky[j] = rk;
            }
            // ---- depth test, outputs ----------------------------------------------------------
#pragma unroll
            for (int j = 0; j < kRowGroup; ++j) {
                const int g = r0 + j;
                uint32_t row = row0 + (uint32_t)g;
                bool row_ok = TIGHT ? true : row < (uint32_t)a.H;             // wave-uniform
                bool in_img = col_ok && row_ok;
                uint32_t lin_i = 0, lin_col = 0;
                if (LINEAR) {
                    in_img = lin_pixel(g, lin_i, row, lin_col);
                    row_ok = tile_ok && (lin_base + (uint32_t)g * 64u) < (uint32_t)a.P;   // the run's first pixel exists
                }
                const double dv = (double)dv16[j] * 0.001;
                const bool vis = test[j] & (qz[j] < dv);
                const bool rk = valid[j] & (risky[j] | (test[j] & !(__builtin_fabs(qz[j] - dv) > gz_m)));
                risky_rows |= (uint32_t)rk << g;
                vis_rows |= (uint32_t)vis << g;
                const unsigned long long vmask = __ballot(vis);
                n_vis += __popcll(vmask);
                n_valid += __popcll(__ballot(valid[j]));
                if (!row_ok) continue;
                if (O::template has<O_VIS_BITS>(a.vis_bits)) {
                    const uint64_t bit0 = LINEAR ? (uint64_t)(lin_base + (uint32_t)g * 64u)
                                                 : (uint64_t)row * (uint64_t)a.W + (uint64_t)stripe * 64u;
                    uint64_t *wp = a.vis_bits + pair * c.words_per_pair + (int64_t)(bit0 >> 6);
                    if (words_aligned) {
                        if (c.lane == 0) *wp = vmask;
                    } else if (c.lane == 0 && vmask) {   // stripes straddle words: OR into the zeroed bitset
                        const uint32_t sh = (uint32_t)(bit0 & 63u);
                        atomicOr((unsigned long long *)wp, vmask << sh);
                        if (sh && (vmask >> (64u - sh))) atomicOr((unsigned long long *)wp + 1, vmask >> (64u - sh));
                    }
                }
                if (in_img) {
                    const uint32_t pix_i = LINEAR ? lin_i : row * (uint32_t)a.W + colc;
                    const int64_t o = c.obase + (int64_t)pix_i;
                    if (O::template has<O_VIS_U8>(a.vis_u8)) a.vis_u8[o] = vis ? 1 : 0;
                    if (O::template has<O_VALID_U8>(a.valid_u8)) a.valid_u8[o] = valid[j] ? 1 : 0;
                    if (O::template has<O_PIX>(a.pix_i16))
                        reinterpret_cast<int *>(a.pix_i16)[o] = test[j] ? pix[j] : -1;
                    if (O::template has<O_XYZ32>(a.xyz_f32)) {
                        float *q = a.xyz_f32 + 3 * o;
                        const float fn = __builtin_nanf("");
                        q[0] = valid[j] ? fx[j] : fn;
                        q[1] = valid[j] ? fy[j] : fn;
                        q[2] = valid[j] ? fz[j] : fn;
                    }
                    if (O::template has<O_RGBA>(a.rgba)) {
                        uint32_t colr = 0;
                        if (c.rgb1) {
                            const uint8_t *s = c.rgb1 + 3 * (int64_t)pix_i;
                            colr = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
                        }
                        a.rgba[o] = colr | (valid[j] ? 0xFF000000u : 0u);
                    }
                }
            }
        }

        // ---- cold loop: exact re-evaluation of the guarded pixels ---------------------------------
        if (__ballot(risky_rows != 0u)) {
            __builtin_amdgcn_s_waitcnt(0);            // the fast path's stores have reached L2
            int delta = 0;
            while (risky_rows) {
                const int g = __builtin_ctz(risky_rows);
                risky_rows &= risky_rows - 1u;
                uint32_t row = row0 + (uint32_t)g;
                uint32_t i = row * (uint32_t)a.W + colc;
                uint32_t colx = col;
                double mxe = mxd;
                if (LINEAR) {
                    lin_pixel(g, i, row, colx);
                    mxe = (double)colx;
                }
                uint32_t dd;
                if (IDENT) {
                    dd = c.depth1[i];
                } else {
                    const int dy = round_clip((double)row * a.sy, a.dh - 1);
                    dd = c.depth1[dy * a.dw + (LINEAR ? round_clip((double)colx * a.sx, a.dw - 1) : dx1)];
                }
                Pixel p;
                exact_unproject(m1, mxe, (double)row, (double)dd * 0.001, p.ax, p.ay, p.az);
                exact_project(m2, p.ax, p.ay, p.az, p.u, p.v, p.qz);
                p.vis = depth_test(true, p.u, p.v, p.qz, c.depth2, a.dh, a.dw, a.H, a.W, a.sx, a.sy, p.xi, p.yi, &p.inview);
                const bool was = (vis_rows >> g) & 1u;
                if (p.vis != was) {
                    delta += p.vis ? 1 : -1;
                    if (O::template has<O_VIS_BITS>(a.vis_bits)) {
                        const uint64_t bit = LINEAR ? (uint64_t)i : (uint64_t)row * (uint64_t)a.W + (uint64_t)col;
                        atomicXor((unsigned long long *)(a.vis_bits + pair * c.words_per_pair + (int64_t)(bit >> 6)),
                                  1ull << (bit & 63u));
                    }
                }
                store_pixel<O, true>(a, c, i, true, true, p);
            }
            // wave-reduce the visible-count corrections (rare path)
            for (int off = 32; off > 0; off >>= 1) delta += __shfl_down(delta, off);
            n_vis += __builtin_amdgcn_readfirstlane(delta);
        }
    }
    // one LDS step and two atomics per workgroup (lane 0 of each wave holds the wave totals)
    if (O::template has<O_COUNTS>(a.counts)) {
        __shared__ int red[2][kThreads / kWave];
        if (c.lane == 0) {
            red[0][wave] = n_valid;
            red[1][wave] = n_vis;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int sv = 0, ss = 0;
            for (int j = 0; j < kThreads / kWave; ++j) {
                sv += red[0][j];
                ss += red[1][j];
            }
            atomicAdd(a.counts + 2 * pair + 0, sv);
            atomicAdd(a.counts + 2 * pair + 1, ss);
        }
    }
}

// --------------------------------------------------------------------------------------------
// fast path, whole-tile images (the BASELINE shape 640x480: W % 64 == 0, H % kTightRows == 0)
// --------------------------------------------------------------------------------------------
// Same arithmetic and guard as pair_fast_kernel, trimmed for VALU issue, which is what bounds the
// fast path (profiles/r01c: VALU pipe 75 % busy; profiles/r04_valu_rates.md: every float64, compare, conversion, VOP3 or packed
// instruction issues at 4 cycles per wave and SIMD, v_rcp_f64 at 16, only v_add_u32 / v_sub_u32 / v_and_b32 / v_mov_b32 and the
// plain float32 add / mul / fma at 2 -- so the lever is the instruction COUNT per pixel):
//   * depth reads, gathers and pixel-index stores go through buffer resources (SGPR base + SGPR row
//     offset + one loop-invariant VGPR column offset): no 64-bit address arithmetic in the vector pipe;
//   * the pixel index comes out of a rounding ADDITION, its clamp out of one saturating pack for both axes, the byte offset
//     of the gather out of one v_dot2_u32_u16; lanes that are not in view gather at an offset the resource drops;
//   * decisions stay in SGPR masks (s_and/s_or are free next to the
//     vector pipe); risky rows are recorded per wave (ballot -> VGPR lane) instead of per-lane bit masks;
//   * 48 rows per wave tile amortise the per-tile matrix composition;
//   * a group of 4 rows none of whose 256 pixels can land in frame 2 skips gather, guard and depth test.
#ifndef MSPA_TIGHT_ROWS
#define MSPA_TIGHT_ROWS 48
#endif
#ifndef MSPA_TIGHT_BLOCK_WAVES
#define MSPA_TIGHT_BLOCK_WAVES 0       // 0 = per output set (tight_bw_of)
#endif
#ifndef MSPA_TIGHT_DMA16
#define MSPA_TIGHT_DMA16 1
#endif
#ifndef MSPA_STAGE2_ROW_BARRIER
#define MSPA_STAGE2_ROW_BARRIER 0      // 0 never, 1 after every row, 2 after the second row only, 3 after the second row for the compacted set only
#endif
#ifndef MSPA_SCALED_ROW_BARRIER
#define MSPA_SCALED_ROW_BARRIER 1      // ScanNet-shape kernel: 100 -> 73 VGPRs (4 -> 6 waves per SIMD), 2.30 -> 2.11 ms per 1 000 pairs
#endif
#ifndef MSPA_SCALED_TILE_CULL
#define MSPA_SCALED_TILE_CULL 1        // ScanNet-shape kernel: frustum test of a tile's colour box against frame 2 before projecting it
#endif
#ifndef MSPA_SCALED_FULL_WAIT
#define MSPA_SCALED_FULL_WAIT 1        // ScanNet-shape kernel: one vmcnt(0) for a group's gathers instead of a counted wait per row (-1..4 %)
#endif
#ifndef MSPA_PRIO
#define MSPA_PRIO 1                    // s_setprio: a tile's prologue above the row loops (correspondence and minimal sets; 0 = off)
#endif
#ifndef MSPA_COMPACT_LDS_PAD
#define MSPA_COMPACT_LDS_PAD 0
#endif
#ifndef MSPA_TIGHT_WAVES_PER_EU
#define MSPA_TIGHT_WAVES_PER_EU 0      // > 0: ask the register allocator for that many waves per SIMD
#endif
#if MSPA_TIGHT_WAVES_PER_EU > 0
#define MSPA_TIGHT_ATTR __attribute__((amdgpu_waves_per_eu(MSPA_TIGHT_WAVES_PER_EU, MSPA_TIGHT_WAVES_PER_EU)))
#else
#define MSPA_TIGHT_ATTR
#endif
constexpr int kTightRows = MSPA_TIGHT_ROWS;            // tile height of the correspondence family (= MSPA_CORR_TILE_H)
#ifndef MSPA_TIGHT_ROWS_DENSE
#define MSPA_TIGHT_ROWS_DENSE 32
#endif
// The dense sets carry a 3 KB transpose stage per wave: with 48-row tiles that is 36 KB of LDS per workgroup (four per CU),
// with 32-row tiles 28 KB (five): measured 1.22 -> 1.17 ms per 1 000 pairs (dense without colour words, tools/ab_k3.py)
constexpr int kTightRowsDense = MSPA_TIGHT_ROWS_DENSE;
#ifndef MSPA_TIGHT_RG_LIGHT
#define MSPA_TIGHT_RG_LIGHT 4
#endif
// rows whose depth-2 gathers are in flight together: the sets without a transpose stage (minimal, compact) may take more
constexpr int tight_rg_of(uint32_t set) { return (set & (O_PIX | O_XYZ32 | O_RGBA | O_VIS_U8)) ? kRowGroup : MSPA_TIGHT_RG_LIGHT; }
// Two A/B knobs of round 5, both measured at noise level and left off (tools/ab_scannet.py / ab_k3.py, one box, ms per 1 000
// pairs, shipped -> knob): MSPA_WG_COMPOSE = 1 -- ScanNet's shape corr 1.717 -> 1.701, minimal 1.343 -> 1.336, compact 1.798 ->
// 1.814; 640x480 corr 0.4494 -> 0.4514, compact 0.3489 -> 0.3497, minimal 0.2855 -> 0.2834: the workgroup barrier costs what
// the ~65 saved issues per tile save.  MSPA_SCALED_CULL_BATCH = 16 -- corr 1.717 -> 1.716, minimal 1.343 -> 1.361, compact
// 1.798 -> 1.793: a culled tile's twelve serial round trips are hidden by the other waves already.
#ifndef MSPA_WG_COMPOSE
#define MSPA_WG_COMPOSE 0              // 1: the pair's composed matrix once per WORKGROUP (wave 0 -> LDS -> the others), not per wave
#endif
#ifndef MSPA_SCALED_CULL_BATCH
#define MSPA_SCALED_CULL_BATCH 4       // SCALED form, culled tile: rows whose samples are requested together for the valid-sample count
#endif
// Tile height of the SCALED form (no LDS tile: height costs no occupancy; one lane per tile row holds its visibility word, so
// 64 at most).  A tile's fixed work -- box scan, composition, culling test, ~400 VALU issues and their round trips -- is paid
// per tile whatever it goes on to do, and at ScanNet's shape about half the tiles are culled: the minimal set runs 10 % faster
// on 64-row tiles (1.298 vs 1.445 ms per 1 000 pairs), the correspondence set 7 % slower (1.821 vs 1.708: coarser culling
// writes more of its index table), tools/ab_scannet.py round 5.  The compacted set's tile is part of the API (48).
#ifndef MSPA_SCALED_ROWS
#define MSPA_SCALED_ROWS 48
#endif
#ifndef MSPA_SCALED_ROWS_NOPIX
#define MSPA_SCALED_ROWS_NOPIX 64
#endif
// Tiles per wave of the SCALED form (MSPA_SCALED_TPW_*, A/B knob, 1 = off): a wave walks this many vertically adjacent tiles --
// launch, the pair's matrices, the column's depth-grid offsets and the other per-stripe constants are paid once for them, the
// culling stays per tile.  Measured at ScanNet's shape (tools/ab_scannet.py, round 5, ms per 1 000 pairs at 1 / 2 / 3 / 4 tiles):
// corr 1.745 / 2.102 / 2.125 / 2.256, minimal 1.349 / 2.021 / 2.055 / 1.996 -- bit-identical and 20-50 % SLOWER: what is live
// across the tile loop costs the row loop its registers (minimal: 42 VGPRs spilled to scratch in it).  Not for the compacted
// set at all: with its spills the SGPR quad of the inline-asm `idxen` store came back wrong (memory fault; static_assert below).
#ifndef MSPA_SCALED_TPW_CORR
#define MSPA_SCALED_TPW_CORR 1
#endif
#ifndef MSPA_SCALED_TPW_COMPACT
#define MSPA_SCALED_TPW_COMPACT 1
#endif
#ifndef MSPA_SCALED_TPW_NOPIX
#define MSPA_SCALED_TPW_NOPIX 1
#endif
constexpr int tight_tpw_of(uint32_t set, bool scaled) {
    return !scaled ? 1 : (set & O_CPIX) ? MSPA_SCALED_TPW_COMPACT : (set & O_PIX) ? MSPA_SCALED_TPW_CORR : MSPA_SCALED_TPW_NOPIX;
}
constexpr int tight_rows_of(uint32_t set, bool scaled = false) {
    return (set & (O_XYZ32 | O_RGBA | O_VIS_U8)) ? kTightRowsDense
           : (scaled && !(set & O_CPIX)) ? ((set & O_PIX) ? MSPA_SCALED_ROWS : MSPA_SCALED_ROWS_NOPIX) : kTightRows;
}
// waves (= tiles) per workgroup, per output set (tools/ab_k3.py, one box, ms per 1 000 pairs at 1 / 2 / 4 / 8 waves): the sets
// without an index table like small workgroups -- a workgroup's LDS is released only when its slowest tile is done --
// minimal 0.325 / 0.309 / 0.319 / 0.351, compact 0.388 / 0.377 / 0.388 / 0.424; corr 0.525 / 0.505 / 0.503 / 0.558; the dense
// point set without colour a large one, dense_xyz 1.303 / 1.200 / 1.149 / 1.132 (with colour words 4 stay better than 8:
// 1.572 vs 1.587).  MSPA_TIGHT_BLOCK_WAVES > 0 forces one size for all (A/B builds).
// The SCALED form (no LDS depth tile: a workgroup's size costs no occupancy) runs best at four for every set
// (tools/ab_scannet.py at ScanNet's shape, ms per 1 000 pairs at 1 / 2 / 4 / 8 waves: minimal 2.76 / 1.94 / 1.55-1.63 / 1.63,
// compact 3.03 / 2.18 / 1.93 / 2.01, corr 2.55 / - / 1.72-1.80 / 1.79).
constexpr int tight_bw_of(uint32_t set, bool scaled = false) {
    return MSPA_TIGHT_BLOCK_WAVES > 0 ? MSPA_TIGHT_BLOCK_WAVES
           : scaled ? 4
           : (set & O_RGBA) ? 4
           : (set & (O_XYZ32 | O_VIS_U8)) ? 8
           : (set & O_PIX) ? 4 : 2;
}
// waves per SIMD the register allocator must leave room for (second argument of __launch_bounds__): the sets without an
// index table run best at six (80 VGPRs); round 4's guard-band bookkeeping had pushed `minimal` to 82 = five waves, +5 %
// (the SCALED correspondence set sat at 97 VGPRs = four waves: held at five)
#ifndef MSPA_CORR_MINWAVES
#define MSPA_CORR_MINWAVES 1           // A/B knob: waves per SIMD the correspondence-table instantiation is held to (register budget)
#endif
#ifndef MSPA_COMPACT_FRACT
#define MSPA_COMPACT_FRACT 0
#endif
#ifndef MSPA_COMPACT_MINWAVES
#define MSPA_COMPACT_MINWAVES MSPA_NOPIX_MINWAVES
#endif
#ifndef MSPA_NOPIX_MINWAVES
#define MSPA_NOPIX_MINWAVES 6
#endif
#ifndef MSPA_SCALED_NOPIX_MINWAVES
#define MSPA_SCALED_NOPIX_MINWAVES 6
#endif
#ifndef MSPA_SCALED_CORR_MINWAVES
#define MSPA_SCALED_CORR_MINWAVES 5
#endif
constexpr int tight_minwaves_of(uint32_t set, bool scaled) {
    return (set & (O_XYZ32 | O_RGBA | O_VIS_U8)) ? 1
           : (set & O_PIX) ? (scaled ? MSPA_SCALED_CORR_MINWAVES : MSPA_CORR_MINWAVES)
           : (scaled ? MSPA_SCALED_NOPIX_MINWAVES : (set & O_CPIX) ? MSPA_COMPACT_MINWAVES : MSPA_NOPIX_MINWAVES);
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int med3_0(int x, int hi) {   // clamp(x, 0, hi), hi wave-uniform
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

// Drop a wave-uniform 64-bit value (a ballot) into lane `lane` (wave-uniform) of a VGPR pair: 2 VALU issues.
// v_writelane_b32 may name only ONE SGPR besides M0 (constant-bus rule of gfx9), so the lane select travels in M0.
// Nothing else in the loop uses M0 (gfx9 LDS instructions do not; the tile's LDS-DMA requests were issued before it).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void writelane64(unsigned long long value, int lane, uint32_t &lo, uint32_t &hi) {
    asm("s_mov_b32 m0, %4\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
        : "+v"(lo), "+v"(hi)
        : "s"((uint32_t)value), "s"((uint32_t)(value >> 32)), "s"(__builtin_amdgcn_readfirstlane(lane))
        : "m0");
}
#pragma clang diagnostic pop

// Ballot of a lane predicate as the i1 it is.  (HIP's __ballot takes an int: the predicate is widened to 0/1 in a VGPR and
// compared against zero again -- a v_cndmask + v_cmp per use that the optimiser does not always fold away.)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Keeps a ballot opaque: no instruction, but the optimiser can no longer see that it came from a lane predicate.
__device__ __forceinline__ unsigned long long opaque_mask(unsigned long long m) {
    asm("" : "+s"(m));
    return m;
}

__device__ __forceinline__ unsigned long long readlane64(uint32_t lo, uint32_t hi, int lane) {
    return (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)lo, lane) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, lane) << 32);
}

// lanes below this one whose bit is set in the wave-uniform mask m, plus base
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m, uint32_t base) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
}

// Compacted correspondences (SET has O_CPIX, see include/mspa.h mspa_pair_correspondences): a tile writes the (xi, yi) of its
// VISIBLE pixels only, in (row, column) order, into its own 3072-entry segment: rank of a lane = entries of the tile so far
// (a scalar, the store's SGPR offset) + set bits below the lane (v_mbcnt), one exec-masked dword store per row with a visible
// lane -- and nothing at all for culled tiles and groups, which is where the dense table spends three quarters of its bytes
// on (-1, -1).  Three forms that staged the entries in LDS were measured slower (tools/ab_k3.py, one box): a 512-entry ring
// and a 320-entry buffer with carry, storing 1 KB chunks -- their 1.25 - 2 KB per wave cost the kernel a workgroup per CU
// (0.586 / 0.536 ms; profiles/r03a_k3_compact_fulltile_ring512_pmc.md) -- and a 128-entry buffer in the tile's pad storing
// 64-entry chunks at the same occupancy as the direct form (0.483 vs 0.468 ms); a fourth that ranks a whole row group into the
// dead depth rows and stores it as one aligned 16-byte-per-lane piece (0.7 M store instructions instead of 2.2 M) came out
// equal (0.3916 vs 0.3917 ms) -- the stores were never the problem.  What was: the rewrite of every tile with a guarded row
// (see the cold loop), 0.07 of 0.46 ms.

// SCALED (round 4): the same kernel on a colour grid that differs from the depth grid, or whose width is not a multiple of 64
// -- ScanNet's 1296 x 968 colour over 640 x 480 depth -- with the SAME rectangular 64 x ROWS tiles (so the compacted output
// needs no dense table at that shape either; the wobbling-stripe kernel below cannot rank a rectangular tile's entries, its
// waves own word-aligned stripes).  What changes, all of it off the per-pixel path:
//   * no LDS depth tile: a row's depth-1 samples come through the reference's colour -> depth pixel map (OPS:285-290) as one
//     2-byte gather per lane and row, requested one row group ahead (byte offsets: the lane's column once per tile, the rows'
//     offsets in a VGPR read lane by lane into the load's scalar offset); the tile's sample range comes from the depth BOX
//     its corners map to (a superset of its samples: the culling and the guard band only get more cautious);
//   * the composed matrix's rows 0 / 1 carry sx / sy (IH:359-366 folded in): u, v are depth-grid coordinates, as in the
//     wobbling-stripe kernel;
//   * ragged right edge (W = 1296: 16 live columns in the 21st stripe): dead lanes load through an out-of-range offset (the
//     buffer returns 0 = "no sample") and store through one (dropped); ragged bottom band: a run-time row count;
//   * the bitset: W % 16 == 0 makes a tile row's 64 bits four ALIGNED 16-bit pieces of the row-major bitset whatever the
//     row -- four 2-byte stores per tile (lane = row) instead of one 8-byte store, no atomics, no wobble.
template <uint32_t SET, bool STREAM, int ROWS, int RG, bool SCALED = false>
__global__ __launch_bounds__(tight_bw_of(SET, SCALED) * kWave, tight_minwaves_of(SET, SCALED)) MSPA_TIGHT_ATTR void pair_fast_tight_kernel(const uint16_t *__restrict__ depth,
                                                                   const uint8_t *__restrict__ rgb,
                                                                   const double *__restrict__ mats,
                                                                   const int32_t *__restrict__ pairs, PairArgs a) {
    using O = Outs<SET, false>;
    constexpr int kTightBW = tight_bw_of(SET, SCALED);
    constexpr bool WANT_XYZ = (SET & O_XYZ32) != 0;
    constexpr bool COMPACT = (SET & O_CPIX) != 0;
    constexpr bool FRACT_GUARD = (!COMPACT || (MSPA_COMPACT_FRACT && !SCALED)) && !(SCALED && (SET & O_PIX));   // stage 2's tie / bound test by v_fract (see there)
    constexpr double kUV = FRACT_GUARD ? 2.0 : 1.0;                           // ... on doubled image coordinates
    static_assert(RG == 4 || !(SET & (O_PIX | O_XYZ32 | O_RGBA | O_VIS_U8)), "the transpose stages move 4-row blocks");
    static_assert(ROWS % RG == 0, "whole row groups");
    static_assert(!COMPACT || !(SET & (O_PIX | O_XYZ32 | O_RGBA | O_VIS_U8)), "the compacted set shares the transpose stage's LDS");
    static_assert(!SCALED || !(SET & (O_XYZ32 | O_RGBA | O_VIS_U8 | O_VALID_U8)), "SCALED: correspondence / minimal / compacted sets");
    int64_t pair;
    uint32_t tgroup;
    if (!decode_block(a, pair, tgroup)) return;
    // A new wave gets its requests and its per-tile work out ahead of the row loops of the others (s_setprio; back to 0 in front
    // of the row loop): corr -1.5 %, minimal -0.5 % (ScanNet's shape: -2.7 %), compact +2 % -- so not for the compacted set
    // (tools/ab_k3.py, ab_scannet.py, one box; raising stage 2 above the rest as well changed nothing).
    constexpr bool PROLOGUE_PRIO = MSPA_PRIO && !(SET & (O_CPIX | O_XYZ32 | O_RGBA | O_VIS_U8));
    if (PROLOGUE_PRIO) __builtin_amdgcn_s_setprio(2);
    const int f1 = pairs[2 * pair + 0];
    const int f2 = pairs[2 * pair + 1];
    const double *m1 = mats + (int64_t)f1 * (MSPA_FRAME_MATS * 16);
    const double *m2 = mats + (int64_t)f2 * (MSPA_FRAME_MATS * 16);
    const int64_t dpix = (int64_t)a.dh * a.dw;
    Ctx c;
    c.depth1 = depth + (int64_t)f1 * dpix;
    c.depth2 = depth + (int64_t)f2 * dpix;
    c.rgb1 = (rgb && (SET & O_RGBA)) ? rgb + (int64_t)f1 * a.P * 3 : nullptr;
    c.obase = pair * (int64_t)a.P;
    c.words_per_pair = (a.P + 63) >> 6;
    c.pair = pair;
    c.lane = threadIdx.x & 63;

    // wave tile -> (row band, column stripe); both wave-uniform
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int TPW = tight_tpw_of(SET, SCALED);
    static_assert(TPW == 1 || !(SET & O_CPIX), "tiles per wave > 1 is not available for the compacted set (see tight_tpw_of)");
    const uint32_t wtile = tgroup * kTightBW + wave;                 // this wave's tile group: TPW vertically adjacent tiles
    const uint32_t sband = a.n_stripes == 1 ? wtile : __umulhi(wtile, a.stripe_magic);
    const uint32_t stripe = wtile - sband * (uint32_t)a.n_stripes;
    const bool tile_ok = wtile < (uint32_t)a.n_wave_tiles;
    const uint32_t col = stripe * 64u + (uint32_t)c.lane;
    uint32_t row0 = sband * (uint32_t)(TPW * ROWS);                  // of the group's first tile; advanced per tile below
    uint32_t tile = sband * (uint32_t)TPW * (uint32_t)a.n_stripes + stripe;   // tile index of the API (band-major)

    // The tile's depth-1 samples (48 rows x 128 B) go HBM -> LDS by LDS-DMA, two rows per wave
    // instruction (lanes 0-31 fetch row 2k, lanes 32-63 row 2k+1, 4 bytes = 2 pixels each): all 24
    // requests are in flight before any arithmetic and cost no VGPRs; their latency hides behind the
    // matrix composition below.  (A register prefetch one row group ahead left the kernel latency
    // bound once skipped groups made an iteration shorter than a memory round trip.)
    // The correspondence set's transpose stage (the 4 x 64 pixel indices of a row group, 1 KB) needs no LDS of its own: the
    // depth-1 rows of a group are dead once the group has loaded its samples, so group g's stage is the 1 KB that ends with
    // its own four rows -- bytes [512 g, 512 g + 1024) of a per-wave region that starts with a 512-byte pad (group 0 has no
    // predecessor).  The wave's LDS operations execute in order; the next group reads ITS rows before its stage overwrites them.
    constexpr bool PX_IN_TILE = !SCALED && (SET & O_PIX) && !(SET & (O_XYZ32 | O_RGBA));
    constexpr int kPadPx = PX_IN_TILE ? 256 : 0;                        // uint16 units: 512 bytes
    // (MSPA_COMPACT_LDS_PAD: A/B knob -- bytes of LDS added per wave of the compacted set to cap its waves per SIMD)
    constexpr int kOccPad = (COMPACT && !SCALED) ? MSPA_COMPACT_LDS_PAD / 2 : 0;
    __shared__ __attribute__((aligned(16))) uint16_t lds_w[kTightBW][SCALED ? 16 : kPadPx + ROWS * 64 + kOccPad];   // SCALED: counter slots only
    uint16_t *const lds_d1w = &lds_w[wave][kPadPx];                     // this wave's 48 x 64 depth-1 samples
    if (tile_ok && !SCALED) {
        typedef __attribute__((address_space(1))) const void gvoid_t;
        typedef __attribute__((address_space(3))) void lvoid_t;
#if MSPA_TIGHT_DMA16
        // gfx950's 16-byte LDS-DMA: eight lanes fetch a row's 128 bytes, one wave instruction lands eight rows (1 KB,
        // contiguous in LDS: lane L writes bytes 16 L .. 16 L + 15 past the base) -- 6 requests per tile instead of 24
        static_assert(ROWS % 8 == 0, "eight rows per 16-byte LDS-DMA request");
        const uint16_t *src = c.depth1 + (int64_t)(row0 + (uint32_t)(c.lane >> 3)) * a.W + stripe * 64u +
                              (uint32_t)(c.lane & 7) * 8u;
#pragma unroll
        for (int k = 0; k < ROWS / 8; ++k)
            __builtin_amdgcn_global_load_lds((gvoid_t *)(src + (int64_t)(8 * k) * a.W),
                                             (lvoid_t *)&lds_d1w[k * 512], 16, 0, STREAM ? 2 : 0);  // aux 2 = nt
#else
        const uint16_t *src = c.depth1 + (int64_t)(row0 + (uint32_t)(c.lane >> 5)) * a.W + stripe * 64u +
                              (uint32_t)(c.lane & 31) * 2u;
#pragma unroll
        for (int k = 0; k < ROWS / 2; ++k)
            __builtin_amdgcn_global_load_lds((gvoid_t *)(src + (int64_t)(2 * k) * a.W),
                                             (lvoid_t *)&lds_d1w[k * 128], 4, 0, STREAM ? 2 : 0);   // aux 2 = nt
#endif
    }

    const double *__restrict__ U = m1 + MSPA_MAT_UNPROJ * 16;
    const double *__restrict__ N = m2 + MSPA_MAT_REPROJ * 16;
    double M[3][4], Us[3][4];
    // millimetre-scaled homogeneous coordinates: M maps (mx*d, my*d, d) with d the RAW millimetre sample to 1000 x the
    // image-space triple, so u and v are unchanged and the third coordinate is the camera-2 depth in millimetres --
    // directly comparable with the raw depth-2 sample (no 0.001 multiply per pixel)
    // (Forming the twelve entries lane-parallel -- one entry per lane, 24 v_readlane -- saves ~60 of the ~100 issue slots this
    // costs per tile, and measured SLOWER: compact +10 %, corr +2 %, dense_xyz +4 %, tools/ab_k3.py round 4: its per-lane
    // matrix loads go through the vector memory path and sit in front of everything else a tile does, where the wave-uniform
    // form reads the scalar cache all tiles of a pair share.)
    auto compose = [&]() {
        double raw[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double row[4];
            compose_row(N, U, r, row);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                raw[r][k] = k < 3 ? row[k] : row[k] * 1000.0;
                if (SCALED) raw[r][k] *= (r == 0 ? a.sx : r == 1 ? a.sy : 1.0);
            }
            if (WANT_XYZ) {
#pragma unroll
                for (int k = 0; k < 4; ++k) Us[r][k] = k < 3 ? U[4 * r + k] * 0.001 : U[4 * r + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // rows 0 / 1 minus (W/2, H/2) times row 2: u and v come out relative to the image centre (W/2, H/2 are integers:
            // rounding ties and integer bounds keep their fractional parts; the extra rounding per entry is inside MSPA_GUARD_C)
            // -- and, for the sets whose stage 2 tests ties and bounds with v_fract, TWICE that (an exact scaling): their row loop
            // works on 2u, 2v, whose distance from an integer says "rounding tie OR integer bound" in one instruction
            M[0][k] = uniform(kUV * __builtin_fma(-a.hw, raw[2][k], raw[0][k]));
            M[1][k] = uniform(kUV * __builtin_fma(-a.hh, raw[2][k], raw[1][k]));
            M[2][k] = uniform(raw[2][k]);
        }
    };
    // MSPA_WG_COMPOSE: the tiles of a workgroup belong to one pair, so its waves would all form the same twelve numbers (~100
    // VALU issues of a tile's ~400 of fixed work).  Wave 0 forms them, parks them in LDS, the others pick them up behind a
    // workgroup barrier (12 broadcast reads + 24 v_readfirstlane).  `all_here`: every wave of the workgroup reaches this point
    // (block-uniform) -- otherwise each wave composes for itself as before.
    constexpr bool WG_COMPOSE = MSPA_WG_COMPOSE && kTightBW > 1 && !WANT_XYZ;
    __shared__ __attribute__((aligned(16))) double lds_mat[WG_COMPOSE ? 12 : 1];
    auto obtain_matrix = [&](bool all_here) {
        if (!WG_COMPOSE || !all_here) {
            compose();
            return;
        }
        if (wave == 0) {
            compose();
            if (c.lane == 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k) lds_mat[4 * r + k] = M[r][k];
            }
        }
        __syncthreads();
        if (wave != 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) M[r][k] = uniform(lds_mat[4 * r + k]);
        }
    };
    // The whole-tile form composes here, behind its LDS-DMA requests.  The SCALED form has no such requests in flight yet: it
    // composes further down, behind the loads of its depth box and of its first row group.
    if (!SCALED) obtain_matrix(true);

    // transpose stage: the group's pixel indices, then (dense set) its rgba words, then its 4 x 64 x 3 point coordinates
    __shared__ __attribute__((aligned(16))) uint32_t lds_pxs[kTightBW][(PX_IN_TILE || !(SET & (O_PIX | O_XYZ32 | O_RGBA))) ? 4 : RG * 64 * (WANT_XYZ ? 3 : 1)];
    static_assert(ROWS <= 64, "one lane per tile row holds that row's visibility word");
#ifndef MSPA_EXPERIMENT_ROWS   // timing-only builds with another tile height (tools/build_variant.sh): the compacted set is then wrong
    static_assert(!COMPACT || (ROWS * 64 == MSPA_CORR_TILE_CAP && ROWS == MSPA_CORR_TILE_H), "tile segment of the compacted set");
#endif
    int n_valid = 0, n_vis = 0;
#pragma unroll 1
    for (int sub = 0; tile_ok && sub < TPW; ++sub) {
        if (sub) {                                       // the next tile down (wave-uniform)
            row0 += (uint32_t)ROWS;
            tile += (uint32_t)a.n_stripes;
            if (row0 >= (uint32_t)a.H) break;
        }
        const int n_vis_before = n_vis;
        const uint32_t Wb = (uint32_t)a.W;
        // buffer resources: SGPR base + byte count; raw addressing = base + voffset (VGPR) + soffset (SGPR)
        const int kRsrcFlags = 0x00020000;
        __amdgpu_buffer_rsrc_t rs_d2 = __builtin_amdgcn_make_buffer_rsrc((void *)c.depth2, 0, (int)(dpix * 2), kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_pix = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.pix_i16 ? a.pix_i16 + 2 * c.obase : nullptr), 0, O::template has<O_PIX>(a.pix_i16) ? (int)(a.P * 4) : 0,
            kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.vis_bits ? a.vis_bits + pair * c.words_per_pair : nullptr), 0,
            O::template has<O_VIS_BITS>(a.vis_bits) ? (int)(c.words_per_pair * 8) : 0, kRsrcFlags);
        // Pixel indices are stored per ROW GROUP, 16 bytes per lane: four dword-per-lane stores per group left the
        // kernel store-issue bound (~3.6 B/clk/CU).  The 4 x 64 pixel indices of a group are transposed
        // through 1 KB of LDS so that lane L owns 4 consecutive pixels of row L / 16.
        // (SCALED: the 4-pixel groups to the right of the image's last column store through an offset the range check drops)
        constexpr int kDropOffset = 0x7FFFFFF0;
        const bool px4_ok = !SCALED || (stripe * 64u + ((uint32_t)c.lane & 15u) * 4u) < Wb;
        const int pix_voff = px4_ok ? (int)((((uint32_t)c.lane >> 4) * Wb + stripe * 64u + ((uint32_t)c.lane & 15u) * 4u) * 4u) : kDropOffset;
        const uint32_t wpr = Wb >> 6;                                   // bitset words per image row
        // SCALED: rows of this tile (ragged bottom band), depth-1 addressing through the colour -> depth pixel map (OPS:285-290)
        const int n_rows = SCALED ? (int)min((uint32_t)ROWS, (uint32_t)a.H - row0) : ROWS;
        __amdgpu_buffer_rsrc_t rs_d1 = __builtin_amdgcn_make_buffer_rsrc((void *)c.depth1, 0, SCALED ? (int)(dpix * 2) : 0, kRsrcFlags);
        int dxb = 0, dyb = 0;                          // byte offsets: this lane's column; lane r: tile row r
        if (SCALED) {
            dxb = col < Wb ? 2 * round_clip((double)col * a.sx, a.dw - 1) : kDropOffset;            // dead lanes read 0
            dyb = round_clip((double)min(row0 + (uint32_t)c.lane, (uint32_t)a.H - 1u) * a.sy, a.dh - 1) * (a.dw * 2);
        }
        auto load_d1_row = [&](int r) -> uint32_t {     // SCALED: depth-1 sample of (tile row r, this lane's column)
            return __builtin_amdgcn_raw_buffer_load_b16(rs_d1, dxb, __builtin_amdgcn_readlane(dyb, r), STREAM ? 2 : 0);
        };
        // the reference's sample of colour pixel (row, col) for the cold paths
        auto sample1 = [&](uint32_t row, uint32_t colx) -> uint32_t {
            if (!SCALED) return c.depth1[row * Wb + colx];
            return c.depth1[round_clip((double)row * a.sy, a.dh - 1) * a.dw + round_clip((double)colx * a.sx, a.dw - 1)];
        };
        // SCALED: the depth BOX the tile's corners map to (both maps are monotone) is scanned for its smallest / largest sample
        // below, in 8-byte pieces, 16 lanes across a depth row, 4 rows per pass (dw % 4 == 0: pieces never straddle a row; the
        // host checks).  ScanNet's 64 x 48 colour tile maps to a 33 x 25 box: one piece column, seven passes.  Up to eight
        // passes are requested HERE, with the first row group's samples, and the matrix composition runs behind them (a
        // tile's first vector-memory round trip used to sit bare between the composition and the culling test).
        constexpr int kBoxPre = ROWS > 48 ? (ROWS + 7) / 8 + 1 : 8;     // 4 depth rows per pass; a colour tile of ROWS rows maps to <= ROWS / 2 + 1 of them at ScanNet's scale
        u32x2 boxw[kBoxPre] = {};
        uint32_t d16_first[RG] = {};
        bool box_pre = false;
        uint32_t box_dyA = 0, box_xb0 = 0, box_nxb = 1, box_nrow = 1;
        const uint32_t cB = SCALED ? min(stripe * 64u + 63u, Wb - 1u) : stripe * 64u + 63u;     // the tile's last live column / row
        const uint32_t rB = row0 + (uint32_t)n_rows - 1u;
        if (SCALED) {
            const uint32_t dw2s = (uint32_t)a.dw * 2u;
            const uint32_t dxA = 2u * (uint32_t)round_clip((double)(stripe * 64u) * a.sx, a.dw - 1);
            const uint32_t dxB = 2u * (uint32_t)round_clip((double)cB * a.sx, a.dw - 1);
            box_dyA = __builtin_amdgcn_readfirstlane((uint32_t)round_clip((double)row0 * a.sy, a.dh - 1) * dw2s);
            const uint32_t dyB = (uint32_t)round_clip((double)rB * a.sy, a.dh - 1) * dw2s;
            box_xb0 = __builtin_amdgcn_readfirstlane(dxA & ~7u);
            box_nxb = __builtin_amdgcn_readfirstlane((dxB + 2u - box_xb0 + 7u) >> 3);
            box_nrow = __builtin_amdgcn_readfirstlane((dyB - box_dyA) / dw2s + 1u);
            box_pre = box_nxb <= 16u && box_nrow <= 4u * kBoxPre;        // wave-uniform
            if (box_pre) {
                const uint32_t piece = min((uint32_t)c.lane & 15u, box_nxb - 1u);
#pragma unroll
                for (int k = 0; k < kBoxPre; ++k) {                      // passes past the box repeat its last row
                    const uint32_t r = min(4u * (uint32_t)k + ((uint32_t)c.lane >> 4), box_nrow - 1u);
                    boxw[k] = __builtin_amdgcn_raw_buffer_load_b64(rs_d1, (int)(box_dyA + r * dw2s + box_xb0 + piece * 8u), 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < RG; ++j) d16_first[j] = load_d1_row(j);
            if (sub == 0) obtain_matrix((tgroup * (uint32_t)kTightBW + (uint32_t)(kTightBW - 1)) < (uint32_t)a.n_wave_tiles);
        }
        // dense payload: byte mask (lane L: 4 pixels of row L >> 4), colour in / rgba out, points (16-byte pieces of the
        // group's 4 x 768 bytes: piece 64 k + L lies in row (16 (64 k + L)) / 768)
        __amdgpu_buffer_rsrc_t rs_vis = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.vis_u8 ? a.vis_u8 + c.obase : nullptr), 0, (SET & O_VIS_U8) ? (int)a.P : 0, kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_rgb = __builtin_amdgcn_make_buffer_rsrc(
            (void *)c.rgb1, 0, ((SET & O_RGBA) && c.rgb1) ? (int)(a.P * 3) : 0, kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_rgba = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.rgba ? a.rgba + c.obase : nullptr), 0, (SET & O_RGBA) ? (int)(a.P * 4) : 0, kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_xyz = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.xyz_f32 ? a.xyz_f32 + 3 * c.obase : nullptr), 0, (SET & O_XYZ32) ? (int)(a.P * 12) : 0, kRsrcFlags);
        // compacted set: the tile's own segment of MSPA_CORR_TILE_CAP entries (4 bytes each)
        // -- a STRUCTURED resource (stride 4): an entry is addressed by its index (`idxen`), the address unit does the x 4
        __amdgpu_buffer_rsrc_t rs_cpix = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(COMPACT ? a.cpix + ((pair * (int64_t)a.n_tiles + (int64_t)tile) * MSPA_CORR_TILE_CAP) * 2 : nullptr), 4,
            COMPACT ? MSPA_CORR_TILE_CAP : 0, kRsrcFlags);
        auto store_entry = [&](uint32_t value, uint32_t index, uint32_t soffset_bytes) {   // no builtin takes an index operand
            asm volatile("buffer_store_dword %0, %1, %2, %3 idxen" ::"v"(value), "v"(index), "s"(rs_cpix), "s"(soffset_bytes) : "memory");
        };
        uint32_t cfill = 0;                          // compacted set, wave-uniform: entries of the tile so far
        // one row's visible lanes store their pixel index at their rank: entries so far (scalar offset) + set bits below the lane
        auto compact_row = [&](unsigned long long vmask, int pixv) {
            if (__builtin_amdgcn_inverse_ballot_w64(vmask)) store_entry((uint32_t)pixv, mbcnt64(vmask, 0u), cfill * 4u);
            cfill += (uint32_t)__popcll(vmask);
        };
        const int vis_voff = (int)(((uint32_t)c.lane >> 4) * Wb + stripe * 64u + ((uint32_t)c.lane & 15u) * 4u);
        const int rgb_voff = (int)(stripe * 192u + (c.lane == 0 ? 0u : (uint32_t)c.lane * 3u - 1u));
        int xyz_voff[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const uint32_t o = (uint32_t)(64 * k + c.lane) * 16u;
            xyz_voff[k] = (int)((o / 768u) * Wb * 12u + stripe * 768u + (o % 768u));
        }
        const int hi_x = a.dw - 1, hi_y = a.dh - 1;
        const uint32_t dw2 = (uint32_t)a.dw * 2u;
        // stage 2's rounding constants (see there): 1.5 * 2^52 + half the grid + the bias that saturates at the last column / row
        const uint32_t bias_x = 32767u - (uint32_t)hi_x, bias_y = 32767u - (uint32_t)hi_y;
        const uint32_t pk_bias = bias_x | (bias_y << 16);
        const uint32_t dot_k = 2u | (dw2 << 16);                                   // (2, 2 DW): DW <= 32 767
        const uint32_t dot_c = 0u - (2u * bias_x + dw2 * bias_y);
        const double k2hw = kUV * a.khw, k2hh = kUV * a.khh;                       // bounds of kUV (u - W / 2), kUV (v - H / 2)
        const double magic_u = 6755399441055744.0 + (a.hw + (double)bias_x), magic_v = 6755399441055744.0 + (a.hh + (double)bias_y);   // 1.5 * 2^52 + half the grid

        const double mxd = (double)col;
        const double myd0 = (double)row0;
        double t0 = __builtin_fma(M[0][1], myd0, __builtin_fma(M[0][0], mxd, M[0][2]));
        double t1 = __builtin_fma(M[1][1], myd0, __builtin_fma(M[1][0], mxd, M[1][2]));
        double t2 = __builtin_fma(M[2][1], myd0, __builtin_fma(M[2][0], mxd, M[2][2]));
        double s0 = 0, s1 = 0, s2 = 0;
        if (WANT_XYZ) {
            s0 = __builtin_fma(Us[0][1], myd0, __builtin_fma(Us[0][0], mxd, Us[0][2]));
            s1 = __builtin_fma(Us[1][1], myd0, __builtin_fma(Us[1][0], mxd, Us[1][2]));
            s2 = __builtin_fma(Us[2][1], myd0, __builtin_fma(Us[2][0], mxd, Us[2][2]));
        }
        const double Wd = SCALED ? (double)a.dw : (double)a.W, Hd = SCALED ? (double)a.dh : (double)a.H;   // bounds in u, v's units
        unsigned long long risky_rows = 0;           // wave-uniform: rows with at least one guarded lane
        // Lane r keeps the visibility word of tile row r: ballots are SGPR pairs and v_writelane drops them into one lane
        // for 2 VALU issues per row; the tile's 48 words leave with ONE store at the end, after the cold loop has patched
        // them in registers.
        uint32_t bits_lo = 0, bits_hi = 0;
        uint32_t rb_lo = 0, rb_hi = 0;               // lane r: guarded-lane ballot of tile row r (flagged rows only; rare path)
        if (!SCALED) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tile's depth-1 samples have landed in LDS

        // ---- tile-level culling -------------------------------------------------------------------
        // The tile's pixels with depths in [dlo, dhi] span a frustum {d*(mx, my, 1)}, the convex hull of 8
        // corner points; the composed map is affine in (mx*d, my*d, d) and "in view" (z > 0, 0 <= x < W*z,
        // 0 <= y < H*z in homogeneous image coordinates) is an intersection of half-spaces.  If all 8
        // projected corners violate ONE of those half-spaces by a margin (>= 1e-4 px, 7 orders above the
        // rounding differences between evaluation orders), no pixel of the tile can land in frame 2:
        // the wave writes "nothing visible" for 48 x 64 pixels without projecting any of them.
        // (Homogeneous coordinates are in pixel * millimetre here: M maps the raw millimetre sample.)
        // The same pass yields the tile's guard band (guard_from_bounds with the tile's far corner and largest sample) and
        // whether the camera-2 plane can be near any of its pixels: if all 8 corners lie in front of it by more than zsafe,
        // every pixel of the tile does (the depth coordinate is affine too), no lane can be "near the plane", and the row
        // loop runs without its two depth-sign compares (`all_front`; the other tiles take the careful instantiation).
        bool culled = false;
        bool all_front = false;
        const double *__restrict__ bnd1 = m1 + MSPA_MAT_BOUNDS * 16;
        const double *__restrict__ bnd2 = m2 + MSPA_MAT_BOUNDS * 16;
        const double tile_xmax = (double)(stripe * 64u + 63u), tile_ymax = (double)(row0 + (uint32_t)(ROWS - 1));
        double zmin = 0.0, gz = kGuardZmmFloor, tnear2 = 0.0;  // wave-uniform (SGPR pairs) once set below
        // the dense payload sets write every pixel, so their tiles are never culled -- but they take the pass all the same: the
        // band comes from the tile's own largest sample (with the format's 65 535 mm it is ~13 x wider, and on distant views
        // whole rows of lanes "near the camera-2 plane" went to the reference chain: dense_xyz 0.98 -> 1.38 ms on `low`)
        constexpr bool CAN_CULL = !WANT_XYZ && !(SET & (O_VALID_U8 | O_RGBA | O_VIS_U8));
        {
            typedef unsigned short us2 __attribute__((ext_vector_type(2)));
            const us2 *wds = reinterpret_cast<const us2 *>(lds_d1w);
            const us2 one = {1, 1};
            int lo, hi;
            if (!SCALED) {
                us2 mn = {0xFFFF, 0xFFFF}, mxv = {0, 0};
#pragma unroll
                for (int k = 0; k < ROWS / 2; ++k) {
                    const us2 x = wds[c.lane + 64 * k];
                    mn = __builtin_elementwise_min(mn, (us2)(x - one));      // 0 (invalid) wraps to 0xFFFF
                    mxv = __builtin_elementwise_max(mxv, x);
                }
                lo = min((int)mn.x, (int)mn.y);
                hi = max((int)mxv.x, (int)mxv.y);
            } else {
                typedef unsigned short us4 __attribute__((ext_vector_type(4)));
                us4 mn = {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF}, mxv = {0, 0, 0, 0};
                const us4 one4 = {1, 1, 1, 1};
                if (box_pre) {                                           // wave-uniform: the pieces requested above
#pragma unroll
                    for (int k = 0; k < kBoxPre; ++k) {
                        us4 x;
                        __builtin_memcpy(&x, &boxw[k], 8);
                        mn = __builtin_elementwise_min(mn, (us4)(x - one4));
                        mxv = __builtin_elementwise_max(mxv, x);
                    }
                } else {
                    const uint32_t q = (uint32_t)c.lane >> 4;
                    for (uint32_t k = 0; k < (box_nrow + 3u) >> 2; ++k) {
                        const uint32_t r = min(4u * k + q, box_nrow - 1u);
                        for (uint32_t j = 0; j < (box_nxb + 15u) >> 4; ++j) {
                            const uint32_t piece = min(16u * j + ((uint32_t)c.lane & 15u), box_nxb - 1u);
                            const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs_d1, (int)(box_dyA + r * dw2 + box_xb0 + piece * 8u), 0, 0);
                            us4 x;
                            __builtin_memcpy(&x, &w, 8);
                            mn = __builtin_elementwise_min(mn, (us4)(x - one4));
                            mxv = __builtin_elementwise_max(mxv, x);
                        }
                    }
                }
                lo = min(min((int)mn.x, (int)mn.y), min((int)mn.z, (int)mn.w));
                hi = max(max((int)mxv.x, (int)mxv.y), max((int)mxv.z, (int)mxv.w));
            }
            {   // one packed reduction for both: (largest sample, 0xFFFF - smallest sample - 1) as two u16 halves, v_pk_max_u16
                us2 key = {(unsigned short)hi, (unsigned short)(0xFFFF - lo)};
                for (int off = 32; off > 0; off >>= 1) {
                    uint32_t kb;
                    __builtin_memcpy(&kb, &key, 4);
                    kb = (uint32_t)__shfl_xor((int)kb, off);
                    us2 other;
                    __builtin_memcpy(&other, &kb, 4);
                    key = __builtin_elementwise_max(key, other);
                }
                hi = (int)key.x;
                lo = 0xFFFF - (int)key.y;
            }
            if (hi == 0) {
                culled = CAN_CULL;                                        // no valid depth sample at all
                all_front = true;                                         // (and nothing the row loop could get wrong)
            } else {
                const int k = c.lane & 7;
                const double cx = (double)((k & 1) ? cB : stripe * 64u);
                const double cy = (double)((k & 2) ? rB : row0);
                const double cd = (double)((k & 4) ? hi : lo + 1);
                double hx = __builtin_fma(__builtin_fma(M[0][1], cy, __builtin_fma(M[0][0], cx, M[0][2])), cd, M[0][3]);
                double hy = __builtin_fma(__builtin_fma(M[1][1], cy, __builtin_fma(M[1][0], cx, M[1][2])), cd, M[1][3]);
                const double hz = __builtin_fma(__builtin_fma(M[2][1], cy, __builtin_fma(M[2][0], cx, M[2][2])), cd, M[2][3]);
                hx = __builtin_fma(a.hw, hz, (1.0 / kUV) * hx);      // rows 0 / 1 are centred (and doubled): back to image coordinates
                hy = __builtin_fma(a.hh, hz, (1.0 / kUV) * hy);
                // margins in homogeneous units (pixel * millimetre; millimetres for the depth); a tile about to be culled
                // checks that they are at least four times what two evaluation orders can differ by (mspa_common.h)
                const bool all_behind = ballot64(hz <= -kCullMarginZ) == ~0ull;
                const bool all_left = ballot64(hx < -kCullMarginXY) == ~0ull;
                const bool all_right = ballot64(hx - Wd * hz > kCullMarginXY) == ~0ull;
                const bool all_above = ballot64(hy < -kCullMarginXY) == ~0ull;
                const bool all_below = ballot64(hy - Hd * hz > kCullMarginXY) == ~0ull;
                culled = CAN_CULL && (all_behind | all_left | all_right | all_above | all_below);
                if (culled) {
                    culled = ballot64(cull_margins_hold(bnd1, bnd2, a.wm1, a.hm1, a.wh_max)) == ~0ull;
                }
                if (!culled) {
                    const Guard gd = guard_from_bounds(bnd1, bnd2, tile_xmax, tile_ymax, (double)hi, a.wh_max);
                    all_front = ballot64(hz > gd.zsafe) == ~0ull;
                    zmin = uniform(gd.zmin);
                    gz = uniform(gd.gz);
                    tnear2 = uniform(kUV * 2.0 * a.wh_max * gd.zmin);      // against the (doubled) homogeneous x, y
                }
            }
            if (culled) {
                if (!SCALED) {
                    // valid samples of the tile: two ballots per LDS dword (low half != 0, high half != 0), counted on the
                    // scalar side -- no cross-lane reduction (the packed-add form + six shuffle steps cost 151 VALU issues of a
                    // culled tile's ~490)
                    const uint32_t *wd32 = reinterpret_cast<const uint32_t *>(lds_d1w);
#pragma unroll
                    for (int k = 0; k < ROWS / 2; ++k) {
                        const uint32_t w = wd32[c.lane + 64 * k];
                        n_valid += __popcll(ballot64((w & 0xFFFFu) != 0u)) + __popcll(ballot64(w > 0xFFFFu));
                    }
                } else {                                   // the tile's own samples (the box only bounded their range)
                    // MSPA_SCALED_CULL_BATCH rows per round trip: a culled tile does nothing else, so its registers are free
                    // for the requests (4 per trip made a 48-row tile twelve serial round trips)
                    constexpr int CB = MSPA_SCALED_CULL_BATCH;
#pragma unroll 1
                    for (int r0 = 0; r0 < n_rows; r0 += CB) {
                        uint32_t d[CB];
#pragma unroll
                        for (int j = 0; j < CB; ++j) d[j] = load_d1_row(min(r0 + j, ROWS - 1));     // rows past the band: counted below
#pragma unroll                                                                                    // only if they exist
                        for (int j = 0; j < CB; ++j)
                            if (r0 + j < n_rows) n_valid += __popcll(ballot64(d[j] != 0u));
                    }
                }
                if (O::template has<O_PIX>(a.pix_i16)) {
                    const u32x4_t none = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                    for (int r0 = 0; r0 < n_rows; r0 += RG)
                        buffer_store_b128_guarded(none, rs_pix, pix_voff, (int)((row0 + (uint32_t)r0) * Wb * 4u));
                }
            }
        }

        // The row loop once per kind of tile (compile-time CAREFUL, as the ScanNet-shape kernel does for its last stripe): a
        // run-time flag inside the loop would be if-converted -- the compares executed regardless.
        auto run_rows = [&](auto careful_c) {
            constexpr bool CAREFUL = decltype(careful_c)::value;
            uint32_t d16n[RG] = {};                  // SCALED: the next row group's samples, requested one group ahead
            if (SCALED) {
#pragma unroll
                for (int j = 0; j < RG; ++j) d16n[j] = d16_first[j];
            }
#pragma unroll 1
            for (int r0 = 0; r0 < n_rows; r0 += RG) {
                uint32_t d16[RG];
#pragma unroll
                for (int j = 0; j < RG; ++j) {
                    d16[j] = SCALED ? d16n[j] : (uint32_t)lds_d1w[(r0 + j) * 64 + c.lane];
                    asm("" : "+v"(d16[j]));      // a plain 32-bit value from here on (ds_read_u16 zero-extends): otherwise the
                }                                // compare below is narrowed to 16 bits and the conversion pays a v_and
                if (SCALED && r0 + RG < n_rows) {    // wave-uniform
#pragma unroll
                    for (int j = 0; j < RG; ++j) d16n[j] = load_d1_row(r0 + RG + j);
                }
                // ---- stage 1: project; "in view" with the guard band folded into the comparison constants:
                // a lane the reference would accept (0 <= u < W, 0 <= v < H, depth > 0) always passes, and a lane
                // that passes without being accepted sits inside a guard band and is re-evaluated exactly.
                // Lane predicates live as 64-bit ballots (SGPR pairs) from here on: carried as `bool` across the
                // branch below the compiler parks them in 0/1 VGPRs and re-compares them (8 VALU issues per row);
                // `opaque_mask` keeps it from folding inverse_ballot(ballot(x)) back into such a bool.
                double u[RG], v[RG], qz[RG];
                float fx[RG], fy[RG], fz[RG];
                unsigned long long vmk[RG], ivm[RG];
                unsigned long long any = 0;
#pragma unroll
                for (int j = 0; j < RG; ++j) {
                    const double dmm = (double)d16[j];
                    const double ix = __builtin_fma(t0, dmm, M[0][3]);
                    const double iy = __builtin_fma(t1, dmm, M[1][3]);
                    const double iz = __builtin_fma(t2, dmm, M[2][3]);          // camera-2 depth, millimetres
                    if (WANT_XYZ) {
                        fx[j] = (float)__builtin_fma(s0, dmm, Us[0][3]);
                        fy[j] = (float)__builtin_fma(s1, dmm, Us[1][3]);
                        fz[j] = (float)__builtin_fma(s2, dmm, Us[2][3]);
                        s0 += Us[0][1];
                        s1 += Us[1][1];
                        s2 += Us[2][1];
                    }
                    t0 += M[0][1];
                    t1 += M[1][1];
                    t2 += M[2][1];
                    double rz = __builtin_amdgcn_rcp(iz);
                    rz = __builtin_fma(__builtin_fma(-iz, rz, 1.0), rz, rz);
                    u[j] = ix * rz;                       // kUV (u - W / 2), kUV (v - H / 2)
                    v[j] = iy * rz;
                    qz[j] = iz;
                    // every ballot is the SGPR result of ONE compare; the conjunctions are scalar ANDs of those words
                    // (a ballot of an AND of predicates is lowered to v_cndmask 0/1 + v_cmp again)
                    vmk[j] = ballot64(d16[j] != 0u);                             // OPS:297
                    // u, v relative to the image centre: one compare per axis
                    ivm[j] = vmk[j] & ballot64(__builtin_fabs(u[j]) < k2hw) & ballot64(__builtin_fabs(v[j]) < k2hh);
                    // CAREFUL tiles (the camera-2 plane may cut the tile's frustum): in front of the plane by more than zmin,
                    // or within zmin of it -- there u and v mean nothing, the lane is a candidate whatever they say and stage 2
                    // hands it to the reference chain.  (NaN depth lands in the second set.)
                    // Of the lanes near the plane only those within 2 max(W, H) zmin of the optical axis (homogeneous x, y) can be
                    // accepted by ANY evaluation order (0 <= x < W z with z <= zmin + B_2): a point within millimetres of the
                    // camera-2 CENTRE, not merely of its plane -- without this cut every row that crosses the plane sent a
                    // lane to the reference chain (dense_xyz on distant views: +14 %).
                    if (CAREFUL)
                        ivm[j] = vmk[j] & ((ivm[j] & ballot64(iz > zmin)) |
                                           (ballot64(!(__builtin_fabs(iz) > zmin)) & ballot64(__builtin_fabs(ix) < tnear2) & ballot64(__builtin_fabs(iy) < tnear2)));
                    any |= ivm[j];
                }
                const uint32_t rowg = row0 + (uint32_t)r0;
                // transpose stage of this group (see the LDS layout above); the dense sets keep a stage of their own
                uint32_t *const lds_px = PX_IN_TILE ? reinterpret_cast<uint32_t *>(&lds_w[wave][0]) + (r0 / RG) * 128 : &lds_pxs[wave][0];
                unsigned long long vm[RG] = {};       // visibility words of the group's rows
                if (any == 0) {
                    // ---- nothing of these 4 x 64 pixels can land in frame 2: no gather, no depth test ----
                    if (O::template has<O_PIX>(a.pix_i16)) {
                        const u32x4_t none = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                        buffer_store_b128_guarded(none, rs_pix, pix_voff, (int)(rowg * Wb * 4u));
                    }
                } else {
                    // ---- stage 2: pixel index, gather, guard ---------------------------------------------
                    int pix[RG];
                    uint32_t dv16[RG];
                    unsigned long long rkc[RG];
#pragma unroll
                    for (int j = 0; j < RG; ++j) {
                        // Rounding by addition: u + (1.5 * 2^52 + n) -- with doubled coordinates 0.5 (2u) + ..., one FMA, one rounding
                        // either way -- is rounded to an integer by
                        // the adder (ties fall in the guard band, whichever way they go) and the sum's low dword IS rint(u) + n.
                        // n = W / 2 (u is relative to the image centre) + a bias that puts column W - 1 at 32 767: the
                        // reference's clip (IH:362-365: u in [W - 0.5, W) is in bounds and reads column W - 1) then is the
                        // saturation of ONE v_cvt_pk_i16_i32 for both axes, and the byte offset of the sample ONE
                        // v_dot2_u32_u16 of the packed pair with (2, 2 DW).  Five issues per row where rint, conversion,
                        // centring add and clamp per axis, shift and multiply-add were ten (+ the pack of the index pair).  A lane
                        // that is not in view gathers at an offset the buffer resource drops: no memory access for it.
                        const double tu = FRACT_GUARD ? __builtin_fma(u[j], 0.5, magic_u) : u[j] + magic_u;
                        const double tv = FRACT_GUARD ? __builtin_fma(v[j], 0.5, magic_v) : v[j] + magic_v;
                        const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(__double2loint(tu), __double2loint(tv)));
                        int goff = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, pk), __builtin_bit_cast(us2_t, dot_k), dot_c, false);
                        asm("" : "+v"(goff));        // one v_cndmask below, not an exec-masked block around the lines above
                        goff = __builtin_amdgcn_inverse_ballot_w64(ivm[j]) ? goff : kDropOffset;
                        dv16[j] = __builtin_amdgcn_raw_buffer_load_b16(rs_d2, goff, 0, 0);
                        asm("" : "+v"(dv16[j]));     // buffer_load_ushort zero-extends: no v_and in front of the conversion
                        pix[j] = (int)(pk - pk_bias);              // in-view lanes: both halves at or above their bias, no borrow
                        // A decision can flip only if u is within the guard of a rounding tie (k + 0.5) or of an integer (the image
                        // bounds are integers), i.e. if 2u is within twice the guard of an integer: h = fract(2u) in [0, 1) must
                        // stay clear of both ends, |h - 0.5| < 0.5 - 2 guard.  (fract is exact; a tiny negative 2u gives the
                        // largest double below 1.)  Two issues per axis where rint's result, its difference from u and the
                        // fold |t| - 0.25 were three.  NaN fails the ordered compare.
                        // (The compacted set and the SCALED correspondence set keep the three-issue form -- the rounded value back
                        // from the sum, its exact difference from u, the fold | |t| - 0.25 | < 0.25 - guard: with v_fract the
                        // register allocator spills / drops a wave for them, +1..6 %, tools/ab_k3.py, ab_scannet.py.)
                        if (FRACT_GUARD) {
                            const double wu = __builtin_amdgcn_fract(u[j]) - 0.5;
                            const double wv = __builtin_amdgcn_fract(v[j]) - 0.5;
                            rkc[j] = ballot64(!(__builtin_fmax(__builtin_fabs(wu), __builtin_fabs(wv)) < 0.5 - 2.0 * kGuardPx));
                        } else {
                            const double wu = __builtin_fabs(u[j] - (tu - magic_u)) - 0.25;
                            const double wv = __builtin_fabs(v[j] - (tv - magic_v)) - 0.25;
                            rkc[j] = ballot64(!(__builtin_fmax(__builtin_fabs(wu), __builtin_fabs(wv)) < 0.25 - kGuardPx));
                        }
                        if (CAREFUL) rkc[j] |= ballot64(!(qz[j] > zmin));        // lanes behind the plane are not in ivm
                        // Scheduling barrier between rows (A/B knob, off): left to itself the scheduler interleaves the four rows'
                        // rounding / guard code and keeps all their temporaries live.  Until round 4's second half the compacted set
                        // took one barrier, after the second row (73 VGPRs; 0.381 vs 0.387 ms with none); with the shorter stage 2
                        // the interleaved form is the better one for every set (compact 0.353 vs 0.391 ms with that barrier,
                        // tools/ab_k3.py, one box; a barrier after every row: corr 0.66, minimal 0.40 ms).
                        if (MSPA_STAGE2_ROW_BARRIER == 1 || (MSPA_STAGE2_ROW_BARRIER == 2 && j == 1) ||
                            (MSPA_STAGE2_ROW_BARRIER == 3 && COMPACT && j == 1))
                            __builtin_amdgcn_sched_barrier(0);
                    }
                    // ---- stage 3: depth test, outputs -------------------------------------------------------
                    unsigned long long rbm[RG];
#pragma unroll
                    for (int j = 0; j < RG; ++j) {
                        const bool inview = __builtin_amdgcn_inverse_ballot_w64(ivm[j]);
                        const double sd = qz[j] - (double)dv16[j];              // millimetres; IH:368-371 compares metres
                        vm[j] = ivm[j] & ballot64(sd < 0.0);
                        rbm[j] = ivm[j] & (rkc[j] | ballot64(!(__builtin_fabs(sd) > gz)));
                        if (O::template has<O_PIX>(a.pix_i16)) lds_px[j * 64 + c.lane] = (uint32_t)(inview ? pix[j] : -1);
                        if (COMPACT) compact_row(vm[j], pix[j]);
                    }
                    unsigned long long rb_any = 0;
#pragma unroll
                    for (int j = 0; j < RG; ++j) rb_any |= rbm[j];
                    if (rb_any) {             // wave-uniform, rare: one branch per group, not per row
#pragma unroll
                        for (int j = 0; j < RG; ++j)
                            if (rbm[j]) {
                                writelane64(rbm[j], r0 + j, rb_lo, rb_hi);
                                risky_rows |= 1ull << (r0 + j);
                            }
                    }
                    if (O::template has<O_PIX>(a.pix_i16)) {
                        wave_lds_fence();
                        const u32x4_t q = *reinterpret_cast<const u32x4_t *>(&lds_px[c.lane * 4]);
                        buffer_store_b128_guarded(q, rs_pix, pix_voff, (int)(rowg * Wb * 4u));
                        wave_lds_fence();
                    }
                }
                // ---- common tail: counters (scalar) and the rows' visibility words (zero after an early-out) ----
#pragma unroll
                for (int j = 0; j < RG; ++j) {
                    n_valid += __popcll(vmk[j]);
                    n_vis += __popcll(vm[j]);
                    // (also what the cold loop reads back.  v_mov_b32 from the SGPRs under a one-lane EXEC mask -- a 2-cycle
                    // instruction where v_writelane_b32 takes 4, tools/micro/valu_rates.hip -- measured no different: the four
                    // EXEC writes per group cost what the moves save)
                    writelane64(vm[j], r0 + j, bits_lo, bits_hi);
                }
                // ---- dense payload of the group (byte mask, coloured points): every store is a whole-wave, 4- or 16-byte
                // per lane store.  The previous form -- a byte store per pixel for the mask, three byte loads and a dword
                // store per pixel for the colour, three dword stores per pixel for the point -- issued 37 vector-memory
                // instructions per row group against 14 now, and the texture-address path, not HBM, set the pace. ----
                if (SET & O_VIS_U8) {
                    // lane L owns pixels 4 (L & 15) .. + 3 of row L >> 4: four bits of that row's ballot, spread to four bytes
                    const uint32_t sh = ((uint32_t)c.lane & 15u) * 4u;
                    const uint32_t n0 = (uint32_t)(vm[0] >> sh), n1 = (uint32_t)(vm[1] >> sh), n2 = (uint32_t)(vm[2] >> sh),
                                   n3 = (uint32_t)(vm[3] >> sh);
                    const uint32_t jl = (uint32_t)c.lane >> 4;
                    const uint32_t nib = (jl == 0 ? n0 : jl == 1 ? n1 : jl == 2 ? n2 : n3) & 0xFu;
                    const uint32_t bytes4 = (nib * 0x00204081u) & 0x01010101u;      // bit k -> byte k
                    __builtin_amdgcn_raw_buffer_store_b32(bytes4, rs_vis, vis_voff, (int)(rowg * Wb), 0);
                }
                if (SET & O_RGBA) {
#pragma unroll
                    for (int j = 0; j < RG; ++j) {
                        // the row's 192 colour bytes: one (unaligned) dword per lane at byte 3 L - 1 (lane 0: byte 0), never
                        // past the end of the row
                        const uint32_t w = __builtin_amdgcn_raw_buffer_load_b32(rs_rgb, rgb_voff, (int)((rowg + (uint32_t)j) * Wb * 3u), 0);
                        const uint32_t colr = c.lane == 0 ? (w & 0xFFFFFFu) : (w >> 8);
                        const bool valid = __builtin_amdgcn_inverse_ballot_w64(vmk[j]);
                        lds_px[j * 64 + c.lane] = colr | (valid ? 0xFF000000u : 0u);
                    }
                    wave_lds_fence();
                    const u32x4_t q = *reinterpret_cast<const u32x4_t *>(&lds_px[c.lane * 4]);
                    buffer_store_b128_guarded(q, rs_rgba, pix_voff, (int)(rowg * Wb * 4u));
                    wave_lds_fence();
                }
                if (SET & O_XYZ32) {
                    const uint32_t fnan = 0x7FC00000u;
#pragma unroll
                    for (int j = 0; j < RG; ++j) {
                        const bool valid = __builtin_amdgcn_inverse_ballot_w64(vmk[j]);
                        uint32_t *dst = &lds_px[(j * 64 + c.lane) * 3];
                        dst[0] = valid ? __float_as_uint(fx[j]) : fnan;
                        dst[1] = valid ? __float_as_uint(fy[j]) : fnan;
                        dst[2] = valid ? __float_as_uint(fz[j]) : fnan;
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const u32x4_t q = *reinterpret_cast<const u32x4_t *>(&lds_px[(k * 64 + c.lane) * 4]);
                        buffer_store_b128_guarded(q, rs_xyz, xyz_voff[k], (int)(rowg * Wb * 12u));
                    }
                    wave_lds_fence();
                }
            }
        };
        if (PROLOGUE_PRIO) __builtin_amdgcn_s_setprio(0);
        if (!culled) {
            if (all_front) run_rows(std::false_type{});
            else run_rows(std::true_type{});
        }

        // ---- cold loop: rows with guarded lanes are re-evaluated with the exact chain ---------------
        bool redo = false;                                  // compacted set: a guarded lane changed its visibility (wave-uniform)
#ifdef MSPA_EXPERIMENT_NOCOLD   // timing only (wrong results on rows with guarded lanes)
        risky_rows = 0;
#endif
        if (risky_rows) {
            __builtin_amdgcn_s_waitcnt(0);                 // the fast path's stores are in L2, LDS writes landed
            // compacted set: lane r <- entries of the tile's rows above r (exclusive prefix of the rows' popcounts), so that
            // a guarded lane that stays visible can be patched at its rank
            uint32_t cpref = 0;
            if (COMPACT) {
                const uint32_t cnt = (uint32_t)__popc(bits_lo) + (uint32_t)__popc(bits_hi);
                uint32_t incl = cnt;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
                    if (c.lane >= off) incl += t;
                }
                cpref = incl - cnt;
            }
            while (risky_rows) {                            // wave-uniform
                const int g = __builtin_amdgcn_readfirstlane(__builtin_ctzll(risky_rows));
                risky_rows &= risky_rows - 1ull;
                const unsigned long long rb = readlane64(rb_lo, rb_hi, g);
                const unsigned long long old = readlane64(bits_lo, bits_hi, g);
                const uint32_t row = row0 + (uint32_t)g;
                const uint32_t i = row * Wb + col;
                const bool mine = (rb >> c.lane) & 1ull;
                bool vis = (old >> c.lane) & 1ull;
                int cxi = 0, cyi = 0;
                if (mine) {
                    Pixel p;
                    exact_unproject(m1, mxd, (double)row, (double)sample1(row, col) * 0.001, p.ax, p.ay, p.az);
                    exact_project(m2, p.ax, p.ay, p.az, p.u, p.v, p.qz);
                    p.vis = depth_test(true, p.u, p.v, p.qz, c.depth2, a.dh, a.dw, a.H, a.W, a.sx, a.sy, p.xi, p.yi, &p.inview);
                    vis = p.vis;
                    cxi = p.xi;
                    cyi = p.yi;
                    store_pixel<O, true>(a, c, i, true, true, p);
                }
                const unsigned long long fresh = ballot64(vis);
                n_vis += __popcll(fresh) - __popcll(old);
                writelane64(fresh, g, bits_lo, bits_hi);
                if (COMPACT) {
                    // A guarded lane that stays visible keeps its rank: its entry is rewritten in place with the reference
                    // chain's index (almost always the value the fast path wrote).  One that changes visibility shifts every
                    // later rank of the tile: the segment is rebuilt below.
                    if (fresh != old) {
                        redo = true;
                    } else if (mine && vis) {
                        store_entry((uint32_t)(uint16_t)cxi | ((uint32_t)(uint16_t)cyi << 16),
                                    mbcnt64(old, (uint32_t)__builtin_amdgcn_readlane((int)cpref, g)), 0u);
                    }
                }
            }
        }
        if (COMPACT) {
            if (redo) {
                // A guarded lane changed its visibility: the tile's segment is rewritten from the patched visibility words
                // with the reference chain for every visible pixel -- which yields the very indices the fast path wrote for
                // unguarded lanes, that being the guard band's contract.  Rebuilding the segment of EVERY tile with a
                // guarded row (2.5 % of the tiles: 48 serialized row round trips each) cost the kernel 0.07 of 0.46 ms;
                // a change of visibility needs fast and exact chain to disagree inside the band, which on real frames
                // all but never happens (identity pairs, where every depth test is a tie: all tiles).
                uint32_t base = 0;
                for (int r = 0; r < n_rows; ++r) {                  // wave-uniform
                    const unsigned long long w = readlane64(bits_lo, bits_hi, r);
                    if (w == 0) continue;
                    if ((w >> c.lane) & 1ull) {
                        const uint32_t row = row0 + (uint32_t)r;
                        Pixel p;
                        exact_unproject(m1, mxd, (double)row, (double)sample1(row, col) * 0.001, p.ax, p.ay, p.az);
                        exact_project(m2, p.ax, p.ay, p.az, p.u, p.v, p.qz);
                        depth_test(false, p.u, p.v, p.qz, c.depth2, a.dh, a.dw, a.H, a.W, a.sx, a.sy, p.xi, p.yi);
                        store_entry((uint32_t)(uint16_t)p.xi | ((uint32_t)(uint16_t)p.yi << 16), mbcnt64(w, base), 0u);
                    }
                    base += (uint32_t)__popcll(w);
                }
            }
            if (c.lane == 0) a.tile_counts[pair * (int64_t)a.n_tiles + (int64_t)tile] = n_vis - n_vis_before;
        }
        // the tile's visibility words: lane r stores the word of row row0 + r (8 bytes; rows are W/8 bytes apart)
        if (SCALED && (Wb & 63u)) {
            // rows are not whole words: the row's 64 bits are four 16-bit pieces at bit (row W + 64 stripe), a multiple of 16 --
            // aligned 2-byte stores, lane = row; pieces past the image's last column (ragged stripe) are dropped
            if (O::template has<O_VIS_BITS>(a.vis_bits) && c.lane < n_rows) {
                const uint32_t byte0 = ((row0 + (uint32_t)c.lane) * Wb + stripe * 64u) >> 3;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t piece = (k & 1) ? ((k & 2) ? bits_hi : bits_lo) >> 16 : ((k & 2) ? bits_hi : bits_lo);
                    const int off = (stripe * 64u + 16u * (uint32_t)k) < Wb ? (int)(byte0 + 2u * (uint32_t)k) : kDropOffset;
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)piece, rs_bits, off, 0, 0);
                }
            }
        } else if (O::template has<O_VIS_BITS>(a.vis_bits) && c.lane < n_rows)
            __builtin_amdgcn_raw_buffer_store_b64(u32x2{bits_lo, bits_hi}, rs_bits, (int)((uint32_t)c.lane * wpr * 8u),
                                                  (int)((row0 * wpr + stripe) * 8u), 0);
    }
    if (O::template has<O_COUNTS>(a.counts) && (!COMPACT || a.counts != nullptr)) {   // the compacted set's counters are optional
        // the wave's totals go to bytes 16..23 of its own (finished) depth tile; thread 0 sums the four after the barrier
        int *red = reinterpret_cast<int *>(&lds_w[wave][kPadPx + 8]);
        if (c.lane == 0) {
            red[0] = n_valid;
            red[1] = n_vis;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int sv = 0, ss = 0;
            for (int j = 0; j < kTightBW; ++j) {
                const int *rj = reinterpret_cast<const int *>(&lds_w[j][kPadPx + 8]);
                sv += rj[0];
                ss += rj[1];
            }
            atomicAdd(a.counts + 2 * pair + 0, sv);
            atomicAdd(a.counts + 2 * pair + 1, ss);
        }
    }
}

// --------------------------------------------------------------------------------------------
// fast path, colour grid over a SMALLER depth grid -- ScanNet's own shape: 1296 x 968 colour over 640 x 480 depth
// (extract_posed_images.py:93-97; OPS:272-290 scales colour pixels onto the depth frame, IH:359-366 scales projections)
// --------------------------------------------------------------------------------------------
// Same arithmetic, guard and bookkeeping as pair_fast_tight_kernel (scalar ballot words, visibility words in VGPR lanes,
// 16-byte index stores through an LDS transpose).  What the shape adds, and how it is kept off the per-pixel path:
//   * W = 1296 is 20.25 words of 64 bits: a 64-column stripe's ballot would straddle two words of the bitset in three
//     rows out of four.  The stripe therefore WOBBLES: in row r it starts at column 64 s + off(r), off = 0, 48, 32, 16 for
//     r mod 4 = 0..3, exactly where a word of the row-major bitset begins, so every ballot is one whole word.  The lane's
//     column moves by a wave-uniform amount per row (+48, -16, -16, -16), i.e. the per-row update of the affine terms stays
//     three adds with scalar operands.  The word that straddles the end of a row (3 rows out of 4) and the 21st word of
//     rows with r mod 4 = 0 belong to the LAST stripe's tiles: their lanes beyond column W - 1 are pixels of the next row
//     (per-lane correction of the affine terms, only in those tiles: 1 tile in 20).
//   * the two grid scalings.  OPS:285-290's colour -> depth pixel (round_clip(my * sy), round_clip(mx * sx)) is two small
//     tables per workgroup (byte offsets, built once with the reference's own float64 expression); a row's samples are one
//     or two cache lines.  IH:362-366's projection -> depth pixel is folded into the composed matrix (rows 0, 1 times sx,
//     sy): the kernel works on us = u * sx directly, bounds 0 <= u < W become 0 <= us < dw up to 1e-13 -- inside the guard
//     band, where lanes are re-evaluated with the reference chain anyway -- and the "near an integer" guard of the depth
//     grid covers the image bounds just as in the equal-grid kernel.
template <int W_, int H_, int DW_, int DH_, uint32_t SET, bool STREAM>
__global__ __launch_bounds__(kThreads) void pair_fast_scaled_kernel(const uint16_t *__restrict__ depth,
                                                                    const uint8_t *__restrict__ rgb,
                                                                    const double *__restrict__ mats,
                                                                    const int32_t *__restrict__ pairs, PairArgs a) {
    using O = Outs<SET, false>;
    static_assert(W_ % 16 == 0 && H_ % 4 == 0 && W_ >= 192, "four rows are a whole number of bitset words");
    static_assert((SET & ~(O_VIS_BITS | O_PIX | O_COUNTS)) == 0, "correspondence / minimal output sets");
    constexpr int S = W_ / 64;                       // full stripes; the last one also owns the row-straddling words
    constexpr int kPeriodWords = 4 * W_ / 64;        // 81: four rows of bitset words
    constexpr int kRows = 48;                        // rows per tile
    // r mod 4 = j: first word of the row relative to the period, column where it starts
    constexpr int kC1 = (W_ + 63) / 64, kC2 = (2 * W_ + 63) / 64, kC3 = (3 * W_ + 63) / 64;
    constexpr int kOff1 = kC1 * 64 - W_, kOff2 = kC2 * 64 - 2 * W_, kOff3 = kC3 * 64 - 3 * W_;
    constexpr bool kExtra0 = (kC1 > S);              // rows with r mod 4 == 0 start S + 1 words (1296: yes)
    static_assert(kExtra0 && kC2 - kC1 == S && kC3 - kC2 == S && kPeriodWords - kC3 == S, "word pattern of the shape");
    constexpr int kOff[4] = {0, kOff1, kOff2, kOff3};
    constexpr int kC[4] = {0, kC1, kC2, kC3};

    int64_t pair;
    uint32_t tgroup;
    if (!decode_block(a, pair, tgroup)) return;
    const int f1 = pairs[2 * pair + 0];
    const int f2 = pairs[2 * pair + 1];
    const double *m1 = mats + (int64_t)f1 * (MSPA_FRAME_MATS * 16);
    const double *m2 = mats + (int64_t)f2 * (MSPA_FRAME_MATS * 16);
    constexpr int64_t dpix = (int64_t)DH_ * DW_;
    constexpr uint32_t P = (uint32_t)W_ * H_;
    Ctx c;
    c.depth1 = depth + (int64_t)f1 * dpix;
    c.depth2 = depth + (int64_t)f2 * dpix;
    c.rgb1 = nullptr;
    c.obase = pair * (int64_t)P;
    c.words_per_pair = (P + 63) >> 6;
    c.pair = pair;
    c.lane = threadIdx.x & 63;

    // ---- colour -> depth pixel tables of the workgroup (OPS:285-290), as byte offsets into a depth frame ----
    __shared__ uint16_t lds_dx[W_ + 64];             // columns W .. W+63 repeat 0 .. 63 (lanes that run into the next row)
    __shared__ uint32_t lds_dy[H_ + 4];
    for (int k = threadIdx.x; k < W_ + 64; k += kThreads)
        lds_dx[k] = (uint16_t)(2 * round_clip((double)(k < W_ ? k : k - W_) * a.sx, DW_ - 1));
    for (int k = threadIdx.x; k < H_ + 4; k += kThreads)
        lds_dy[k] = (uint32_t)(round_clip((double)(k < H_ ? k : H_ - 1) * a.sy, DH_ - 1) * (DW_ * 2));
    __shared__ uint32_t lds_px[kThreads / kWave][(kRowGroup + 1) * 64];     // index transpose stage (+ the extra word)
    __shared__ int red[2][kThreads / kWave];
    __syncthreads();

    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t tile = tgroup * (kThreads / kWave) + wave;
    constexpr uint32_t kBands = (H_ + kRows - 1) / kRows;
    const uint32_t band = tile / (uint32_t)S;
    const uint32_t stripe = tile - band * (uint32_t)S;
    const bool tile_ok = tile < kBands * (uint32_t)S;
    const uint32_t row0 = band * (uint32_t)kRows;
    const bool last = stripe == (uint32_t)(S - 1);                        // wave-uniform
    const int n_groups = tile_ok ? (int)(min((uint32_t)kRows, (uint32_t)H_ - row0) / 4u) : 0;

    // composed matrix, millimetre-scaled, rows 0 / 1 additionally scaled onto the depth grid (us = u * sx, vs = v * sy)
    const double *__restrict__ U = m1 + MSPA_MAT_UNPROJ * 16;
    const double *__restrict__ N = m2 + MSPA_MAT_REPROJ * 16;
    double M[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double row[4];
        compose_row(N, U, r, row);
        const double sc = r == 0 ? a.sx : r == 1 ? a.sy : 1.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) M[r][k] = uniform((k < 3 ? row[k] : row[k] * 1000.0) * sc);
    }

    int n_valid = 0, n_vis = 0;
    if (n_groups > 0) {
        const int kRsrcFlags = 0x00020000;
        __amdgpu_buffer_rsrc_t rs_d1 = __builtin_amdgcn_make_buffer_rsrc((void *)c.depth1, 0, (int)(dpix * 2), kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_d2 = __builtin_amdgcn_make_buffer_rsrc((void *)c.depth2, 0, (int)(dpix * 2), kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_pix = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.pix_i16 ? a.pix_i16 + 2 * c.obase : nullptr), 0, O::template has<O_PIX>(a.pix_i16) ? (int)(P * 4) : 0,
            kRsrcFlags);
        __amdgpu_buffer_rsrc_t rs_bits = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.vis_bits ? a.vis_bits + pair * c.words_per_pair : nullptr), 0,
            O::template has<O_VIS_BITS>(a.vis_bits) ? (int)(c.words_per_pair * 8) : 0, kRsrcFlags);
        const uint32_t colL = stripe * 64u + (uint32_t)c.lane;             // the lane's column in rows with r mod 4 == 0
        const uint32_t period0 = row0 >> 2;                                // row0 is a multiple of 4
        // 16-byte index store of a row group: lane L owns pixels 4 (L & 15) .. + 3 of the word of row j = L >> 4
        const uint32_t jL = (uint32_t)c.lane >> 4;
        const uint32_t wordL = (uint32_t)(jL == 0 ? kC[0] : jL == 1 ? kC[1] : jL == 2 ? kC[2] : kC[3]) + stripe;
        const int pix_voff = (int)((wordL * 64u + ((uint32_t)c.lane & 15u) * 4u) * 4u);
        const int hi_x = DW_ - 1, hi_y = DH_ - 1;
        constexpr double DWd = (double)DW_, DHd = (double)DH_;
        // affine terms of row row0 (off = 0): t_k = M[k][0] * col + M[k][1] * row + M[k][2]
        const double mxd = (double)colL, myd0 = (double)row0;
        double t0 = __builtin_fma(M[0][1], myd0, __builtin_fma(M[0][0], mxd, M[0][2]));
        double t1 = __builtin_fma(M[1][1], myd0, __builtin_fma(M[1][0], mxd, M[1][2]));
        double t2 = __builtin_fma(M[2][1], myd0, __builtin_fma(M[2][0], mxd, M[2][2]));
        // wave-uniform per-row steps (SGPR pairs): to the next row of the group, and from row 3 to the next group's row 0
        double D[4][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            D[0][k] = uniform(M[k][1] + (double)(kOff[1] - kOff[0]) * M[k][0]);
            D[1][k] = uniform(M[k][1] + (double)(kOff[2] - kOff[1]) * M[k][0]);
            D[2][k] = uniform(M[k][1] + (double)(kOff[3] - kOff[2]) * M[k][0]);
            D[3][k] = uniform(M[k][1] + (double)(0 - kOff[3]) * M[k][0]);
        }
        double E64[3], Wr[3];                        // 64 columns to the right; one row down and W columns to the left
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            E64[k] = uniform(64.0 * M[k][0]);
            Wr[k] = uniform(M[k][1] - (double)W_ * M[k][0]);
        }
        // ---- tile-level culling (as in the tight kernel; not for the last stripe, whose words wrap into the next row) ----
        // The tile's pixels lie in the colour box [cA, cB] x [rA, rB] (64 columns + the largest wobble), their depth-1 samples
        // in the depth box the two tables map its corners to (both tables are monotone): the box's sample range comes from
        // <= 7 eight-byte loads per lane (16 lanes per depth row, 4 rows per instruction), and if all 8 corners of the
        // frustum {d (mx, my, 1)} violate ONE of the in-view half-spaces by a margin, no pixel of the tile can land in
        // frame 2: its groups then only count their valid samples and write "nothing visible".
        // Guard band and `all_front` as in the tight kernel (bounds taken over the colour grid, which only widens them: the
        // kernel's u, v are in depth-grid units, sx, sy <= 1).
        bool culled = false;
        bool all_front = false;
        const double *__restrict__ bnd1 = m1 + MSPA_MAT_BOUNDS * 16;
        const double *__restrict__ bnd2 = m2 + MSPA_MAT_BOUNDS * 16;
        constexpr double wh_max = (double)(W_ > H_ ? W_ : H_);
        double zmin = 0.0, gz = kGuardZmmFloor, tnear = 0.0;
        if (last || !MSPA_SCALED_TILE_CULL) {             // no pre-pass for the last stripe: the bound over the full sample range
            const Guard gd = guard_from_bounds(bnd1, bnd2, (double)(W_ - 1), (double)(row0 + (uint32_t)(kRows - 1)), 65535.0, wh_max);
            zmin = uniform(gd.zmin);
            gz = uniform(gd.gz);
            tnear = uniform(2.0 * wh_max * gd.zmin);
        }
#if MSPA_SCALED_TILE_CULL
        if (!last) {
            typedef unsigned short us4 __attribute__((ext_vector_type(4)));
            const uint32_t cA = stripe * 64u, cB = cA + 63u + (uint32_t)kOff[1];       // kOff[1] is the largest wobble
            static_assert(kOff[1] >= kOff[2] && kOff[1] >= kOff[3] && (S - 2) * 64 + 63 + kOff[1] < W_, "colour box of a tile");
            const uint32_t rA = row0, rB = row0 + 4u * (uint32_t)n_groups - 1u;
            const uint32_t dxA = lds_dx[cA], dxB = lds_dx[cB];                          // byte offsets within a depth row
            const uint32_t dyA = lds_dy[rA], dyB = lds_dy[rB];                          // byte offsets of the depth rows
            const uint32_t xb0 = dxA & ~7u;
            const uint32_t nxb = (dxB + 2u - xb0 + 7u) >> 3;                            // 8-byte pieces per depth row
            const uint32_t nrow = (dyB - dyA) / (uint32_t)(DW_ * 2) + 1u;
            static_assert((DW_ * 2) % 8 == 0, "8-byte pieces never straddle a depth row");
            const uint32_t xoff = xb0 + min((uint32_t)c.lane & 15u, nxb - 1u) * 8u;
            const uint32_t q = (uint32_t)c.lane >> 4;
            us4 mn = {0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF}, mxv = {0, 0, 0, 0};
            const us4 one = {1, 1, 1, 1};
            const uint32_t nk = __builtin_amdgcn_readfirstlane((nrow + 3u) >> 2);
            for (uint32_t k = 0; k < nk; ++k) {
                const uint32_t r = min(4u * k + q, nrow - 1u);
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs_d1, (int)(dyA + r * (uint32_t)(DW_ * 2) + xoff), 0, 0);
                us4 x;
                __builtin_memcpy(&x, &w, 8);
                mn = __builtin_elementwise_min(mn, (us4)(x - one));                     // 0 (invalid) wraps to 0xFFFF
                mxv = __builtin_elementwise_max(mxv, x);
            }
            int lo = min(min((int)mn.x, (int)mn.y), min((int)mn.z, (int)mn.w));
            int hi = max(max((int)mxv.x, (int)mxv.y), max((int)mxv.z, (int)mxv.w));
            for (int off = 32; off > 0; off >>= 1) {
                lo = min(lo, __shfl_xor(lo, off));
                hi = max(hi, __shfl_xor(hi, off));
            }
            if (hi == 0) {
                culled = true;                                                          // no valid depth sample in the box
            } else {
                const int k = c.lane & 7;
                const double cx = (double)((k & 1) ? cB : cA);
                const double cy = (double)((k & 2) ? rB : rA);
                const double cd = (double)((k & 4) ? hi : lo + 1);
                const double hx = __builtin_fma(__builtin_fma(M[0][1], cy, __builtin_fma(M[0][0], cx, M[0][2])), cd, M[0][3]);
                const double hy = __builtin_fma(__builtin_fma(M[1][1], cy, __builtin_fma(M[1][0], cx, M[1][2])), cd, M[1][3]);
                const double hz = __builtin_fma(__builtin_fma(M[2][1], cy, __builtin_fma(M[2][0], cx, M[2][2])), cd, M[2][3]);
                const bool all_behind = ballot64(hz <= -kCullMarginZ) == ~0ull;        // margins: see the tight kernel
                const bool all_left = ballot64(hx < -kCullMarginXY) == ~0ull;
                const bool all_right = ballot64(hx - DWd * hz > kCullMarginXY) == ~0ull;
                const bool all_above = ballot64(hy < -kCullMarginXY) == ~0ull;
                const bool all_below = ballot64(hy - DHd * hz > kCullMarginXY) == ~0ull;
                culled = all_behind | all_left | all_right | all_above | all_below;
                if (culled) culled = ballot64(cull_margins_hold(bnd1, bnd2, (double)(W_ - 1), (double)(H_ - 1), wh_max)) == ~0ull;
                if (!culled) {
                    const Guard gd = guard_from_bounds(bnd1, bnd2, (double)cB, (double)rB, (double)hi, wh_max);
                    all_front = ballot64(hz > gd.zsafe) == ~0ull;
                    zmin = uniform(gd.zmin);
                    gz = uniform(gd.gz);
                    tnear = uniform(2.0 * wh_max * gd.zmin);
                }
            }
            culled = __builtin_amdgcn_readfirstlane((int)culled) != 0;                  // wave-uniform, and known to be
        }
#endif
        // The tile body once per kind of stripe (compile-time LAST): a run-time flag inside the row loop would keep both
        // variants' masks and corrections live at once (52 spilled SGPRs).
        auto run_tile = [&](auto last_c, auto careful_c) {
            constexpr bool LAST = decltype(last_c)::value;
            constexpr bool CAREFUL = decltype(careful_c)::value;   // the camera-2 plane may cut the tile's frustum (tight kernel)
            unsigned long long risky_chunks = 0;         // wave-uniform: chunk slots with at least one guarded lane
            uint32_t bits_lo = 0, bits_hi = 0, rb_lo = 0, rb_hi = 0;   // lane = chunk slot (5 per group in the last stripe, else 4)
            constexpr int NCH = LAST ? 5 : 4;               // words per row group: the last stripe also owns the 21st word of row 0
            constexpr int slots_per_group = NCH;

            // one chunk = 64 consecutive pixel indices = one word.  (tk) affine terms of the chunk's lanes, (rowc, colc) the
            // lanes' pixel, (wrapped) lanes that belong to the next row.  Returns the visibility word; fills pixv / riskw.
            auto depth1_offset = [&](uint32_t rowu, int col_add, bool may_wrap, bool &wrapped) -> int {
                const uint32_t colw = colL + (uint32_t)col_add;                          // may run past W in the last stripe
                wrapped = may_wrap && (colw >= (uint32_t)W_);
                const uint32_t dyo = wrapped ? lds_dy[rowu + 1] : lds_dy[rowu];
                return (int)(dyo + (uint32_t)lds_dx[colw]);
            };

            // depth-1 samples of a group's chunks (1-2 cache lines each), requested ONE GROUP AHEAD: the loads of group g + 1
            // fly while group g is projected (they were the exposed latency at the top of every iteration)
            bool wrapped[NCH];
            auto load_group = [&](int g, uint32_t (&d)[NCH]) {
                const uint32_t rowg = row0 + 4u * (uint32_t)g;
    #pragma unroll
                for (int e = 0; e < NCH; ++e) {
                    // e = 0..3: rows rowg + e at this stripe; e = 4 (last stripe only): the extra word of row rowg, 64 columns on
                    const int j = e < 4 ? e : 0;
                    const int col_add = kOff[j] + (e == 4 ? 64 : 0);
                    const bool can_wrap = (e == 4) || (kOff[j] + 64 * S > W_);            // compile time: which words straddle a row end
                    const int off = depth1_offset(rowg + (uint32_t)j, col_add, LAST && can_wrap, wrapped[e]);   // wrapped: same every group
                    d[e] = __builtin_amdgcn_raw_buffer_load_b16(rs_d1, off, 0, STREAM ? 2 : 0);
                }
            };
            uint32_t d16n[NCH];
            load_group(0, d16n);
            int g_first = 0;
            if (!LAST && culled) {                           // wave-uniform: count the valid samples, write "nothing visible"
    #pragma unroll 1
                for (int g = 0; g < n_groups; ++g) {
                    uint32_t d16[NCH];
    #pragma unroll
                    for (int e = 0; e < NCH; ++e) d16[e] = d16n[e];
                    if (g + 1 < n_groups) load_group(g + 1, d16n);
    #pragma unroll
                    for (int e = 0; e < NCH; ++e) n_valid += __popcll(ballot64(d16[e] != 0u));
                    if (O::template has<O_PIX>(a.pix_i16)) {
                        const u32x4_t none = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                        buffer_store_b128_guarded(none, rs_pix, pix_voff, (int)((period0 + (uint32_t)g) * (uint32_t)kPeriodWords * 256u));
                    }
                }
                g_first = n_groups;
            }
    #pragma unroll 1
            for (int g = g_first; g < n_groups; ++g) {
                uint32_t d16[NCH];
    #pragma unroll
                for (int e = 0; e < NCH; ++e) {
                    d16[e] = d16n[e];
                    asm("" : "+v"(d16[e]));
                }
                if (g + 1 < n_groups) load_group(g + 1, d16n);
                // ---- stage 1 ----
                double u[NCH], v[NCH], qz[NCH];
                unsigned long long vmk[NCH], ivm[NCH];
                unsigned long long any = 0;
                double t0e = 0, t1e = 0, t2e = 0;
    #pragma unroll
                for (int e = 0; e < NCH; ++e) {
                    double a0 = t0, a1 = t1, a2 = t2;
                    if (e == 4) {                         // 64 columns to the right of the row-0 chunk
                        a0 = t0e; a1 = t1e; a2 = t2e;
                    }
                    const bool can_wrap = (e == 4) || (kOff[e < 4 ? e : 0] + 64 * S > W_);
                    if (LAST && can_wrap) {               // lanes past the end of the row are pixels of the next row
                        const bool w = wrapped[e];
                        a0 = w ? a0 + Wr[0] : a0;
                        a1 = w ? a1 + Wr[1] : a1;
                        a2 = w ? a2 + Wr[2] : a2;
                    }
                    const double dmm = (double)d16[e];
                    const double ix = __builtin_fma(a0, dmm, M[0][3]);
                    const double iy = __builtin_fma(a1, dmm, M[1][3]);
                    const double iz = __builtin_fma(a2, dmm, M[2][3]);
                    if (e < 4) {                          // step to the next row of the group (or to the next group)
                        if (LAST && e == 0) { t0e = t0 + E64[0]; t1e = t1 + E64[1]; t2e = t2 + E64[2]; }
                        t0 += D[e][0];
                        t1 += D[e][1];
                        t2 += D[e][2];
                    }
                    double rz = __builtin_amdgcn_rcp(iz);
                    rz = __builtin_fma(__builtin_fma(-iz, rz, 1.0), rz, rz);
                    u[e] = ix * rz;                       // depth-grid units
                    v[e] = iy * rz;
                    qz[e] = iz;
                    vmk[e] = ballot64(d16[e] != 0u);
                    ivm[e] = vmk[e] & ballot64(u[e] > -kGuardPx) & ballot64(u[e] < DWd + kGuardPx) & ballot64(v[e] > -kGuardPx) &
                             ballot64(v[e] < DHd + kGuardPx);
                    if (CAREFUL)      // near the plane: only within 2 max(W, H) zmin of the optical axis (see the tight kernel)
                        ivm[e] = vmk[e] & ((ivm[e] & ballot64(iz > zmin)) |
                                           (ballot64(!(__builtin_fabs(iz) > zmin)) & ballot64(__builtin_fabs(ix) < tnear) & ballot64(__builtin_fabs(iy) < tnear)));
                    any |= ivm[e];
                }
                unsigned long long vm[NCH] = {};
                const int slot0 = g * slots_per_group;
                const uint32_t wbase = (period0 + (uint32_t)g) * (uint32_t)kPeriodWords;     // word of (rowg, column 0)
                if (any == 0) {
                    if (O::template has<O_PIX>(a.pix_i16)) {
                        const u32x4_t none = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                        buffer_store_b128_guarded(none, rs_pix, pix_voff, (int)(wbase * 256u));
                        if (LAST) __builtin_amdgcn_raw_buffer_store_b32(0xFFFFFFFFu, rs_pix, (int)((uint32_t)c.lane * 4u),
                                                                        (int)((wbase + (uint32_t)S) * 256u), 0);
                    }
                } else {
                    int pix[NCH];
                    uint32_t dv16[NCH];
                    unsigned long long rkc[NCH];
    #pragma unroll
                    for (int e = 0; e < NCH; ++e) {
                        const double ru = __builtin_rint(u[e]), rv = __builtin_rint(v[e]);
                        const int xi = med3_0((int)ru, hi_x);
                        const int yi = med3_0((int)rv, hi_y);
#if !defined(MSPA_EXPERIMENT_GATHER)
                        dv16[e] = __builtin_amdgcn_raw_buffer_load_b16(rs_d2, (int)(__umul24((uint32_t)yi, (uint32_t)(DW_ * 2)) + ((uint32_t)xi << 1)), 0, 0);
#elif MSPA_EXPERIMENT_GATHER == 1   // timing only, as in the tight kernel
                        dv16[e] = __builtin_amdgcn_raw_buffer_load_b16(rs_d2, (int)(__umul24((uint32_t)__builtin_amdgcn_readfirstlane(yi), (uint32_t)(DW_ * 2)) + ((uint32_t)xi << 1)), 0, 0);
#else
                        dv16[e] = (uint32_t)xi + 1000u;
#endif
                        pix[e] = (int)((uint32_t)xi | ((uint32_t)yi << 16));
                        const double wu = __builtin_fabs(u[e] - ru) - 0.25;
                        const double wv = __builtin_fabs(v[e] - rv) - 0.25;
                        rkc[e] = ballot64(!(__builtin_fmax(__builtin_fabs(wu), __builtin_fabs(wv)) < 0.25 - kGuardPx));
                        if (CAREFUL) rkc[e] |= ballot64(!(qz[e] > zmin));
                        if (MSPA_SCALED_ROW_BARRIER) __builtin_amdgcn_sched_barrier(0);   // see the tight kernel: registers vs interleaving
                    }
                    unsigned long long rbm[NCH];
#if MSPA_SCALED_FULL_WAIT
                    __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): ONE wait for the group's gathers instead of one per row
#endif
    #pragma unroll
                    for (int e = 0; e < NCH; ++e) {
                        const bool inview = __builtin_amdgcn_inverse_ballot_w64(ivm[e]);
                        const double sd = qz[e] - (double)dv16[e];
                        vm[e] = ivm[e] & ballot64(sd < 0.0);
                        rbm[e] = ivm[e] & (rkc[e] | ballot64(!(__builtin_fabs(sd) > gz)));
                        if (O::template has<O_PIX>(a.pix_i16)) lds_px[wave][e * 64 + c.lane] = (uint32_t)(inview ? pix[e] : -1);
                    }
                    unsigned long long rb_any = 0;
    #pragma unroll
                    for (int e = 0; e < NCH; ++e) rb_any |= rbm[e];
                    if (rb_any) {
    #pragma unroll
                        for (int e = 0; e < NCH; ++e)
                            if (rbm[e]) {
                                writelane64(rbm[e], slot0 + e, rb_lo, rb_hi);
                                risky_chunks |= 1ull << (slot0 + e);
                            }
                    }
                    if (O::template has<O_PIX>(a.pix_i16)) {
                        wave_lds_fence();
                        const u32x4_t q = *reinterpret_cast<const u32x4_t *>(&lds_px[wave][c.lane * 4]);
                        buffer_store_b128_guarded(q, rs_pix, pix_voff, (int)(wbase * 256u));
                        if (LAST) __builtin_amdgcn_raw_buffer_store_b32(lds_px[wave][4 * 64 + c.lane], rs_pix, (int)((uint32_t)c.lane * 4u),
                                                                        (int)((wbase + (uint32_t)S) * 256u), 0);
                        wave_lds_fence();
                    }
                }
    #pragma unroll
                for (int e = 0; e < NCH; ++e) {
                    n_valid += __popcll(vmk[e]);
                    n_vis += __popcll(vm[e]);
                    writelane64(vm[e], slot0 + e, bits_lo, bits_hi);
                }
            }

            // chunk slot -> (word within the pair, row of its first lane, column of its first lane)
            auto slot_word = [&](int slot, uint32_t &row, uint32_t &col0) -> uint32_t {
                const int g = slot / slots_per_group, e = slot - g * slots_per_group;
                const int j = e < 4 ? e : 0;
                row = row0 + 4u * (uint32_t)g + (uint32_t)j;
                col0 = stripe * 64u + (uint32_t)kOff[j] + (e == 4 ? 64u : 0u);
                return (period0 + (uint32_t)g) * (uint32_t)kPeriodWords + (uint32_t)kC[j] + stripe + (e == 4 ? 1u : 0u);
            };
            // ---- cold loop: chunks with guarded lanes are re-evaluated with the exact chain ----
            if (risky_chunks) {
                __builtin_amdgcn_s_waitcnt(0);
                while (risky_chunks) {
                    const int slot = __builtin_amdgcn_readfirstlane(__builtin_ctzll(risky_chunks));
                    risky_chunks &= risky_chunks - 1ull;
                    const unsigned long long rb = readlane64(rb_lo, rb_hi, slot);
                    const unsigned long long old = readlane64(bits_lo, bits_hi, slot);
                    uint32_t row, col0;
                    const uint32_t w = slot_word(slot, row, col0);
                    uint32_t col = col0 + (uint32_t)c.lane;
                    if (col >= (uint32_t)W_) { col -= (uint32_t)W_; row += 1; }
                    const uint32_t i = w * 64u + (uint32_t)c.lane;                       // == row * W + col
                    const bool mine = (rb >> c.lane) & 1ull;
                    bool vis = (old >> c.lane) & 1ull;
                    if (mine) {
                        const uint32_t dd = c.depth1[round_clip((double)row * a.sy, DH_ - 1) * DW_ + round_clip((double)col * a.sx, DW_ - 1)];
                        Pixel p;
                        exact_unproject(m1, (double)col, (double)row, (double)dd * 0.001, p.ax, p.ay, p.az);
                        exact_project(m2, p.ax, p.ay, p.az, p.u, p.v, p.qz);
                        p.vis = depth_test(true, p.u, p.v, p.qz, c.depth2, DH_, DW_, H_, W_, a.sx, a.sy, p.xi, p.yi, &p.inview);
                        vis = p.vis;
                        store_pixel<O, true>(a, c, i, true, true, p);
                    }
                    const unsigned long long fresh = ballot64(vis);
                    n_vis += __popcll(fresh) - __popcll(old);
                    writelane64(fresh, slot, bits_lo, bits_hi);
                }
            }
            // the tile's visibility words: lane = chunk slot
            if (O::template has<O_VIS_BITS>(a.vis_bits) && c.lane < n_groups * slots_per_group) {
                uint32_t row, col0;
                const uint32_t w = slot_word(c.lane, row, col0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{bits_lo, bits_hi}, rs_bits, (int)(w * 8u), 0, 0);
            }
        };
        if (last) run_tile(std::true_type{}, std::true_type{});
        else if (all_front) run_tile(std::false_type{}, std::false_type{});
        else run_tile(std::false_type{}, std::true_type{});
    }
    if (O::template has<O_COUNTS>(a.counts)) {
        if (c.lane == 0) {
            red[0][wave] = n_valid;
            red[1][wave] = n_vis;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int sv = 0, ss = 0;
            for (int j = 0; j < kThreads / kWave; ++j) {
                sv += red[0][j];
                ss += red[1][j];
            }
            atomicAdd(a.counts + 2 * pair + 0, sv);
            atomicAdd(a.counts + 2 * pair + 1, ss);
        }
    }
}

// --------------------------------------------------------------------------------------------
// compaction of a dense (vis_bits, pix_i16) result into the per-tile segments of mspa_pair_correspondences: the route for
// shapes the fused tight kernel does not take (ragged tiles, colour grid != depth grid, reference-order mode).  One wave
// per 64 x 48 tile, rows in order, rank = entries so far + visible lanes below (same order as the fused kernel).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void compact_corr_kernel(const uint64_t *__restrict__ vis_bits,
                                                                const uint32_t *__restrict__ pix, int64_t n_pairs, int H,
                                                                int W, int n_stripes, int n_tiles, uint32_t *__restrict__ cpix,
                                                                int32_t *__restrict__ tile_counts) {
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
    const int64_t pair = wid / n_tiles;
    if (pair >= n_pairs) return;
    const int tile = (int)(wid - pair * n_tiles);
    const int band = tile / n_stripes, stripe = tile - band * n_stripes;
    const int64_t P = (int64_t)H * W;
    const int64_t wpp = (P + 63) >> 6;
    const int col = stripe * 64 + lane;
    uint32_t *seg = cpix + (pair * n_tiles + tile) * (int64_t)MSPA_CORR_TILE_CAP;
    uint32_t base = 0;
    for (int r = 0; r < MSPA_CORR_TILE_H; ++r) {
        const int row = band * MSPA_CORR_TILE_H + r;
        if (row >= H) break;                                              // wave-uniform
        bool vis = false;
        const int64_t i = (int64_t)row * W + col;
        if (col < W) vis = (vis_bits[pair * wpp + (i >> 6)] >> (i & 63)) & 1ull;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(vis);
        if (vis) seg[mbcnt64(m, base)] = pix[pair * P + i];
        base += (uint32_t)__popcll(m);
    }
    if (lane == 0) tile_counts[pair * n_tiles + tile] = (int32_t)base;
}

}  // namespace mspa

using namespace mspa;

static thread_local int g_last_pair_kernel = MSPA_KERNEL_NONE;

extern "C" int mspa_pair_reproject_last_kernel(void) { return g_last_pair_kernel; }

// whole-tile shapes the tight kernel takes (W % 64 == 0, H % 48 == 0, colour grid == depth grid, 32-bit byte offsets)
static bool tight_shape(int32_t dh, int32_t dw, int32_t H, int32_t W, int rows = kTightRows) {
    return dh == H && dw == W && (W % 64 == 0) && (H % rows == 0) && ((uint64_t)H * (uint64_t)W * 4 < (1ull << 31));
}

// shapes the tight kernel takes in its SCALED form (rectangular tiles on a colour grid at least as fine as the depth grid):
// bitset rows in whole 16-bit pieces, whole 4-row groups, depth rows in whole 8-byte pieces, 32-bit byte offsets
static bool rect_shape(int32_t dh, int32_t dw, int32_t H, int32_t W) {
    return !tight_shape(dh, dw, H, W) && dw <= W && dh <= H && (W % 16 == 0) && (H % 4 == 0) && (dw % 4 == 0) && (dh % 2 == 0) &&
           ((uint64_t)H * (uint64_t)W * 4 < (1ull << 31)) && ((uint64_t)dh * (uint64_t)dw * 2 < (1ull << 31));
}

static int pair_reproject_impl(const uint16_t *depth, const uint8_t *rgb, const double *frame_mats,
                               int32_t n_frames, const int32_t *pairs, int64_t n_pairs, int32_t dh,
                               int32_t dw, int32_t H, int32_t W, uint64_t *out_vis_bits,
                               uint8_t *out_vis_u8, uint8_t *out_valid_u8, int16_t *out_pix_i16,
                               float *out_xyz_f32, uint32_t *out_rgba, double *out_xyz_f64,
                               double *out_uv_f64, double *out_depth_f64, int32_t *out_counts,
                               int16_t *out_cpix, int32_t *out_tile_counts, uint32_t flags, mspa_stream_t stream) {
    if (n_frames <= 0 || n_pairs < 0) return fail(MSPA_EINVAL, "mspa_pair_reproject: bad frame/pair count");
    if (!depth || !frame_mats || (!pairs && n_pairs > 0))
        return fail(MSPA_EINVAL, "mspa_pair_reproject: null input pointer");
    if (dh < 2 || dw < 2 || H < 2 || W < 2 || dh > 32767 || dw > 32767 || H > 32767 || W > 32767)
        return fail(MSPA_EINVAL, "mspa_pair_reproject: image size out of range [2, 32767]");
    const uint64_t P = (uint64_t)H * (uint64_t)W;
    if (P * (uint64_t)W >= (1ull << 32)) return fail(MSPA_EINVAL, "mspa_pair_reproject: H*W*W must be < 2^32");
    if (out_rgba && !rgb) return fail(MSPA_EINVAL, "mspa_pair_reproject: out_rgba needs rgb");
    if (flags & ~(MSPA_PAIR_FAST | MSPA_PAIR_STREAM | MSPA_PAIR_WORD_STRIPES)) return fail(MSPA_EINVAL, "mspa_pair_reproject: unknown flag");
    if (n_pairs == 0) return MSPA_OK;
    hipStream_t s = (hipStream_t)stream;

    PairArgs a;
    a.n_pairs = n_pairs;
    a.dh = dh; a.dw = dw; a.H = H; a.W = W; a.P = (uint32_t)P;
    a.div_magic = (uint32_t)((1ull << 32) / (uint64_t)W) + 1u;
    a.sx = (double)dw / (double)W;
    a.sy = (double)dh / (double)H;
    a.strips = (int)((P + kStrip - 1) / kStrip);
    a.vis_bits = out_vis_bits; a.vis_u8 = out_vis_u8; a.valid_u8 = out_valid_u8; a.pix_i16 = out_pix_i16;
    a.xyz_f32 = out_xyz_f32; a.rgba = out_rgba; a.xyz_f64 = out_xyz_f64; a.uv_f64 = out_uv_f64;
    a.depth_f64 = out_depth_f64; a.counts = out_counts;
    a.cpix = out_cpix; a.tile_counts = out_tile_counts;
    const int n_xcd = xcd_count();
    a.xcd_shift = n_xcd == 8 ? 3u : 0u;
    a.wm1 = (double)(W - 1); a.hm1 = (double)(H - 1); a.wh_max = (double)(W > H ? W : H);
    a.hwi = dw / 2; a.hhi = dh / 2;                       // u, v live on the depth grid (== the colour grid for the whole-tile shapes)
    a.hw = (double)a.hwi; a.hh = (double)a.hhi;
    a.khw = a.hw + kGuardPx; a.khh = a.hh + kGuardPx;

    if (out_counts) {      // 2 us per launch (tools/ab_k3.py: 0.5044 vs 0.5062 ms without / with)
        int rc = check_hip(hipMemsetAsync(out_counts, 0, sizeof(int32_t) * 2 * n_pairs, s), "hipMemsetAsync(counts)");
        if (rc) return rc;
    }
    const bool ident = (dh == H && dw == W);
    // float64 outputs are defined as the reference's own operation order: they force the exact kernel
    // ... and so does a bitset output whose words straddle the 64-column stripes (W % 64 != 0, e.g. ScanNet's
    // 1296-wide colour grid): the stripe-mapped fast kernel would need two atomicOr per wave-row there and
    // measures slower than the exact kernel's word-aligned linear mapping (7.7 vs 6.7 ms per 1 000 pairs).
    // ... and so does a depth grid LARGER than the colour grid (sx or sy > 1; no dataset has one): the guard band is built from
    // colour-grid quantities and its half-guard margin is only argued for sx, sy <= 1.
    const bool fast = (flags & MSPA_PAIR_FAST) && !out_xyz_f64 && !out_uv_f64 && !out_depth_f64 && dw <= W && dh <= H;
    const bool linear = fast && out_vis_bits && (W % 64 != 0);   // bitset on a width that is not a multiple of 64
    uint32_t set = 0;
    set |= out_vis_bits ? O_VIS_BITS : 0; set |= out_vis_u8 ? O_VIS_U8 : 0; set |= out_valid_u8 ? O_VALID_U8 : 0;
    set |= out_pix_i16 ? O_PIX : 0; set |= out_xyz_f32 ? O_XYZ32 : 0; set |= out_rgba ? O_RGBA : 0;
    set |= out_counts ? O_COUNTS : 0; set |= out_cpix ? (O_CPIX | O_COUNTS) : 0;   // compacted set: counters optional at run time
    // the tight kernel's LDS-DMA moves depth in 4-byte units and its 16-byte stores need aligned outputs
    const bool aligned = (((uintptr_t)depth & 3u) == 0) && (((uintptr_t)out_pix_i16 & 15u) == 0) &&
                         (((uintptr_t)out_vis_bits & 7u) == 0) && (((uintptr_t)out_cpix & 15u) == 0) &&
                         (((uintptr_t)out_xyz_f32 & 15u) == 0) && (((uintptr_t)out_rgba & 15u) == 0) &&
                         (((uintptr_t)out_vis_u8 & 3u) == 0);
    const bool tight24 = fast && aligned && tight_shape(dh, dw, H, W, tight_rows_of(set)) &&
                         (set == kSetCorr || set == kSetDense || set == kSetDenseXyz || set == kSetMinimal || set == kSetCompact);
    // The tight kernel's SCALED form (rectangular tiles) takes the correspondence / minimal / compacted sets on every
    // rect_shape, ScanNet's own (1296 x 968 colour over 640 x 480 depth) included: at four waves per workgroup it beats round
    // 2-3's wobbling-stripe kernel there (corr 1.72-1.80 vs 1.90-2.05, minimal 1.55-1.63 vs 1.64-1.69 ms per 1 000 pairs,
    // tools/ab_scannet.py), which stays reachable under MSPA_PAIR_WORD_STRIPES
    const bool scannet = W == 1296 && H == 968 && dw == 640 && dh == 480;
    const bool word_stripes = scannet && (flags & MSPA_PAIR_WORD_STRIPES) && (set == kSetCorr || set == kSetMinimal);
    const bool rect = fast && !tight24 && aligned && rect_shape(dh, dw, H, W) && !word_stripes &&
                      (set == kSetCompact || set == kSetCorr || set == kSetMinimal);
    if (out_cpix && !((tight24 || rect) && out_tile_counts))
        return fail(MSPA_EINVAL, "pair_reproject_impl: the fused compacted set needs the tight kernel and a tile-count table");
    const bool scaled = fast && !tight24 && !rect && aligned && scannet && (set == kSetCorr || set == kSetMinimal);
    if (scaled) {
        a.n_stripes = 1296 / 64;
        a.n_tiles = a.n_wave_tiles = a.n_stripes * ((968 + 47) / 48);
        a.stripe_magic = 0;
        a.strips = (a.n_tiles + (kThreads / kWave) - 1) / (kThreads / kWave);
    } else if (fast) {
        const int tile_rows = (tight24 || rect) ? tight_rows_of(set, rect) : kTileRows;
        a.n_stripes = (W + 63) / 64;
        a.n_tiles = (linear && !rect) ? (int)((P + (int64_t)kTileRows * 64 - 1) / ((int64_t)kTileRows * 64))
                                      : a.n_stripes * ((H + tile_rows - 1) / tile_rows);
        a.stripe_magic = (uint32_t)((1ull << 32) / (uint64_t)a.n_stripes) + 1u;
        const int bw = rect ? tight_bw_of(set, true) : tight24 ? tight_bw_of(set) : (kThreads / kWave);
        const int tpw = rect ? tight_tpw_of(set, true) : 1;                       // tiles a wave walks (vertically adjacent)
        a.n_wave_tiles = (tight24 || rect) ? a.n_stripes * (((H + tile_rows - 1) / tile_rows + tpw - 1) / tpw) : a.n_tiles;
        a.strips = (a.n_wave_tiles + bw - 1) / bw;
    } else {
        a.n_stripes = a.n_tiles = a.n_wave_tiles = 0;
        a.stripe_magic = 0;
    }
    const int64_t groups = (n_pairs + n_xcd - 1) / n_xcd;
    const int64_t blocks = groups * n_xcd * a.strips;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_pair_reproject: too many workgroups; split the batch");
    const dim3 grid((uint32_t)blocks), block(kThreads);
    g_last_pair_kernel = !fast ? MSPA_KERNEL_PAIR_EXACT
                         : scaled ? MSPA_KERNEL_PAIR_FAST_SCALED
                         : rect ? MSPA_KERNEL_PAIR_FAST_RECT
                         : tight24 ? MSPA_KERNEL_PAIR_FAST_TIGHT
                         : linear ? MSPA_KERNEL_PAIR_FAST_LINEAR : MSPA_KERNEL_PAIR_FAST;
    if (!fast) {
        if (ident) hipLaunchKernelGGL(pair_exact_kernel<true>, grid, block, 0, s, depth, rgb, frame_mats, pairs, a);
        else hipLaunchKernelGGL(pair_exact_kernel<false>, grid, block, 0, s, depth, rgb, frame_mats, pairs, a);
    } else if (scaled) {
        const bool st = (flags & MSPA_PAIR_STREAM) != 0;
#define MSPA_LAUNCH_SCALED(SET_, ST_) \
    hipLaunchKernelGGL((pair_fast_scaled_kernel<1296, 968, 640, 480, SET_, ST_>), grid, block, 0, s, depth, rgb, frame_mats, pairs, a)
        if (set == kSetCorr) { if (st) MSPA_LAUNCH_SCALED(kSetCorr, true); else MSPA_LAUNCH_SCALED(kSetCorr, false); }
        else { if (st) MSPA_LAUNCH_SCALED(kSetMinimal, true); else MSPA_LAUNCH_SCALED(kSetMinimal, false); }
#undef MSPA_LAUNCH_SCALED
    } else if (rect) {
#define MSPA_LAUNCH_RECT(SET_) \
    do { \
        if (flags & MSPA_PAIR_STREAM) \
            hipLaunchKernelGGL((pair_fast_tight_kernel<SET_, true, tight_rows_of(SET_, true), tight_rg_of(SET_), true>), grid, dim3(tight_bw_of(SET_, true) * kWave), 0, s, depth, rgb, frame_mats, pairs, a); \
        else \
            hipLaunchKernelGGL((pair_fast_tight_kernel<SET_, false, tight_rows_of(SET_, true), tight_rg_of(SET_), true>), grid, dim3(tight_bw_of(SET_, true) * kWave), 0, s, depth, rgb, frame_mats, pairs, a); \
    } while (0)
        if (set == kSetCorr) MSPA_LAUNCH_RECT(kSetCorr);
        else if (set == kSetCompact) MSPA_LAUNCH_RECT(kSetCompact);
        else MSPA_LAUNCH_RECT(kSetMinimal);
#undef MSPA_LAUNCH_RECT
    } else if (tight24) {
#define MSPA_LAUNCH_TIGHT(SET_) \
    do { \
        if (flags & MSPA_PAIR_STREAM) \
            hipLaunchKernelGGL((pair_fast_tight_kernel<SET_, true, tight_rows_of(SET_), tight_rg_of(SET_)>), grid, dim3(tight_bw_of(SET_) * kWave), 0, s, depth, rgb, frame_mats, pairs, a); \
        else \
            hipLaunchKernelGGL((pair_fast_tight_kernel<SET_, false, tight_rows_of(SET_), tight_rg_of(SET_)>), grid, dim3(tight_bw_of(SET_) * kWave), 0, s, depth, rgb, frame_mats, pairs, a); \
    } while (0)
        if (set == kSetCorr) MSPA_LAUNCH_TIGHT(kSetCorr);
        else if (set == kSetDense) MSPA_LAUNCH_TIGHT(kSetDense);
        else if (set == kSetDenseXyz) MSPA_LAUNCH_TIGHT(kSetDenseXyz);
        else if (set == kSetCompact) MSPA_LAUNCH_TIGHT(kSetCompact);
        else MSPA_LAUNCH_TIGHT(kSetMinimal);
#undef MSPA_LAUNCH_TIGHT
    } else {
#define MSPA_LAUNCH_FAST(ID, TI, SET_, GEN) \
    hipLaunchKernelGGL((pair_fast_kernel<ID, TI, SET_, GEN>), grid, block, 0, s, depth, rgb, frame_mats, pairs, a)
        const bool tight = (W % 64 == 0) && (H % kTileRows == 0);
        if (ident && tight) {           // the BASELINE shape: 640x480 colour == depth
            if (set == kSetCorr) MSPA_LAUNCH_FAST(true, true, kSetCorr, false);
            else if (set == kSetDense) MSPA_LAUNCH_FAST(true, true, kSetDense, false);
            else if (set == kSetMinimal) MSPA_LAUNCH_FAST(true, true, kSetMinimal, false);
            else MSPA_LAUNCH_FAST(true, true, 0u, true);
        } else if (linear) {
#define MSPA_LAUNCH_LINEAR(ID, SET_, GEN) \
    hipLaunchKernelGGL((pair_fast_kernel<ID, false, SET_, GEN, true>), grid, block, 0, s, depth, rgb, frame_mats, pairs, a)
            if (ident) {
                if (set == kSetCorr) MSPA_LAUNCH_LINEAR(true, kSetCorr, false);
                else if (set == kSetMinimal) MSPA_LAUNCH_LINEAR(true, kSetMinimal, false);
                else MSPA_LAUNCH_LINEAR(true, 0u, true);
            } else {
                if (set == kSetCorr) MSPA_LAUNCH_LINEAR(false, kSetCorr, false);
                else if (set == kSetMinimal) MSPA_LAUNCH_LINEAR(false, kSetMinimal, false);
                else MSPA_LAUNCH_LINEAR(false, 0u, true);
            }
#undef MSPA_LAUNCH_LINEAR
        } else if (ident) {
            MSPA_LAUNCH_FAST(true, false, 0u, true);
        } else {
            MSPA_LAUNCH_FAST(false, false, 0u, true);
        }
#undef MSPA_LAUNCH_FAST
    }
    return check_hip(hipGetLastError(), "pair_reproject kernel launch");
}

extern "C" int mspa_pair_reproject(const uint16_t *depth, const uint8_t *rgb, const double *frame_mats,
                                   int32_t n_frames, const int32_t *pairs, int64_t n_pairs, int32_t dh,
                                   int32_t dw, int32_t H, int32_t W, uint64_t *out_vis_bits,
                                   uint8_t *out_vis_u8, uint8_t *out_valid_u8, int16_t *out_pix_i16,
                                   float *out_xyz_f32, uint32_t *out_rgba, double *out_xyz_f64,
                                   double *out_uv_f64, double *out_depth_f64, int32_t *out_counts,
                                   uint32_t flags, mspa_stream_t stream) {
    return pair_reproject_impl(depth, rgb, frame_mats, n_frames, pairs, n_pairs, dh, dw, H, W, out_vis_bits, out_vis_u8,
                               out_valid_u8, out_pix_i16, out_xyz_f32, out_rgba, out_xyz_f64, out_uv_f64, out_depth_f64,
                               out_counts, nullptr, nullptr, flags, stream);
}

static bool corr_args_ok(int32_t H, int32_t W) { return H >= 2 && W >= 2 && H <= 32767 && W <= 32767; }

extern "C" int64_t mspa_corr_tiles(int32_t H, int32_t W) {
    if (!corr_args_ok(H, W)) return -1;
    return (int64_t)((W + MSPA_CORR_TILE_W - 1) / MSPA_CORR_TILE_W) * ((H + MSPA_CORR_TILE_H - 1) / MSPA_CORR_TILE_H);
}

extern "C" int64_t mspa_pair_correspondences_workspace_bytes(int64_t n_pairs, int32_t dh, int32_t dw, int32_t H, int32_t W,
                                                             uint32_t flags) {
    if (n_pairs < 0 || !corr_args_ok(H, W)) return -1;
    if ((flags & MSPA_PAIR_FAST) && (tight_shape(dh, dw, H, W) || rect_shape(dh, dw, H, W))) return 0;   // fused: no dense table
    return n_pairs * (int64_t)H * W * 4;
}

extern "C" int mspa_compact_correspondences(const uint64_t *vis_bits, const int16_t *pix_i16, int64_t n_pairs, int32_t H,
                                            int32_t W, int16_t *out_cpix_i16, int32_t *out_tile_counts,
                                            mspa_stream_t stream) {
    if (n_pairs < 0 || !corr_args_ok(H, W)) return fail(MSPA_EINVAL, "mspa_compact_correspondences: bad pair count / image size");
    if (n_pairs == 0) return MSPA_OK;
    if (!vis_bits || !pix_i16 || !out_cpix_i16 || !out_tile_counts)
        return fail(MSPA_EINVAL, "mspa_compact_correspondences: null pointer");
    if (((uintptr_t)pix_i16 & 3u) || ((uintptr_t)out_cpix_i16 & 3u))
        return fail(MSPA_EINVAL, "mspa_compact_correspondences: pixel-index tables must be 4-byte aligned");
    const int n_stripes = (W + MSPA_CORR_TILE_W - 1) / MSPA_CORR_TILE_W;
    const int n_tiles = n_stripes * ((H + MSPA_CORR_TILE_H - 1) / MSPA_CORR_TILE_H);
    const int64_t waves = n_pairs * n_tiles;
    const int64_t blocks = (waves + (kThreads / kWave) - 1) / (kThreads / kWave);
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_compact_correspondences: too many workgroups; split the batch");
    hipLaunchKernelGGL(compact_corr_kernel, dim3((uint32_t)blocks), dim3(kThreads), 0, (hipStream_t)stream, vis_bits,
                       reinterpret_cast<const uint32_t *>(pix_i16), n_pairs, H, W, n_stripes, n_tiles,
                       reinterpret_cast<uint32_t *>(out_cpix_i16), out_tile_counts);
    return check_hip(hipGetLastError(), "compact_corr_kernel launch");
}

extern "C" int mspa_pair_correspondences(const uint16_t *depth, const double *frame_mats, int32_t n_frames,
                                         const int32_t *pairs, int64_t n_pairs, int32_t dh, int32_t dw, int32_t H, int32_t W,
                                         uint64_t *out_vis_bits, int16_t *out_cpix_i16, int32_t *out_tile_counts,
                                         int32_t *out_counts, void *workspace, int64_t workspace_bytes, uint32_t flags,
                                         mspa_stream_t stream) {
    if (n_pairs == 0) return MSPA_OK;
#ifdef MSPA_EXPERIMENT_ROWS
    return fail(MSPA_EUNSUPPORTED, "mspa_pair_correspondences: timing-only build with another tile height");
#endif
    if (!out_vis_bits || !out_cpix_i16 || !out_tile_counts)
        return fail(MSPA_EINVAL, "mspa_pair_correspondences: vis_bits, cpix and tile_counts are all required");
    if (((uintptr_t)out_cpix_i16 & 15u) || ((uintptr_t)out_vis_bits & 7u))
        return fail(MSPA_EINVAL, "mspa_pair_correspondences: out_cpix_i16 must be 16-byte, out_vis_bits 8-byte aligned");
    const int64_t need = mspa_pair_correspondences_workspace_bytes(n_pairs, dh, dw, H, W, flags);
    if (need < 0) return fail(MSPA_EINVAL, "mspa_pair_correspondences: bad pair count / image size");
    // the fused kernel's LDS-DMA moves depth in 4-byte units: a depth table that is only 2-byte aligned takes the dense route
    const bool depth_aligned = ((uintptr_t)depth & 3u) == 0;
    if (need == 0 && depth_aligned)
        return pair_reproject_impl(depth, nullptr, frame_mats, n_frames, pairs, n_pairs, dh, dw, H, W, out_vis_bits, nullptr,
                                   nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, out_counts, out_cpix_i16,
                                   out_tile_counts, flags, stream);
    const int64_t dense = n_pairs * (int64_t)H * W * 4;
    if (n_pairs > 0 && (!workspace || workspace_bytes < dense || ((uintptr_t)workspace & 15u)))
        return fail(MSPA_EINVAL, need == 0 ? "mspa_pair_correspondences: the fused kernel needs a 4-byte aligned depth table; this one "
                                             "goes through the dense table: pass a 16-byte aligned workspace of n_pairs * H * W * 4 bytes"
                                           : "mspa_pair_correspondences: this shape / mode goes through the dense table: pass a 16-byte "
                                             "aligned workspace of n_pairs * H * W * 4 bytes");
    int rc = pair_reproject_impl(depth, nullptr, frame_mats, n_frames, pairs, n_pairs, dh, dw, H, W, out_vis_bits, nullptr,
                                 nullptr, (int16_t *)workspace, nullptr, nullptr, nullptr, nullptr, nullptr, out_counts,
                                 nullptr, nullptr, flags, stream);
    if (rc) return rc;
    return mspa_compact_correspondences(out_vis_bits, (const int16_t *)workspace, n_pairs, H, W, out_cpix_i16,
                                        out_tile_counts, stream);
}
