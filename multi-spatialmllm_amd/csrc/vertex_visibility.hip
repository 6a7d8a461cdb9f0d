// K1: scene vertices -> a batch of images: projection + bounds + depth-buffer test
// (HOT LOOP 1 of CFR.process_scene / MVI.process_scene; see include/mspa.h).
//
// Mapping.  One lane per vertex; a workgroup (256 threads) keeps its 256 vertices in registers and
// walks kImgPerBlock images, so vertex coordinates are fetched from HBM once per image chunk
// instead of once per image.  Per image the two 3x4 matrices come in through wave-uniform scalar
// loads.  The visibility word of a wave is its 64-lane ballot: bitsets are written coalesced, one
// uint64 per wave per image, in exactly the layout K2 consumes.
#include "mspa_common.h"

namespace mspa {

// Inputs are separate `const T *__restrict__` kernel parameters so the wave-uniform matrix reads
// become scalar loads (see pair_reproject.hip).
struct VertexArgs {
    int64_t n_points, point_stride, comp_stride;
    int n_images;
    int dh, dw, H, W;
    double sx, sy;
    int64_t n_words;
    uint64_t *bits;
    uint8_t *mask;
    double *uv;
    double *depth_out;
    int32_t *count_atomic;
    uint32_t vblocks, igroups;     // vertex blocks, image groups of kImgPerBlock (grid decode)
    uint32_t n_xcd;                // XCDs the hardware deals workgroups over (1 = plain linear decode)
    double dscale;                 // depth_value_scale (IH:76, IH:368): metres per depth unit, 0.001 for ScanNet
};

#ifndef MSPA_VTHREADS
#define MSPA_VTHREADS 256
#endif
constexpr int kVThreads = MSPA_VTHREADS;
static_assert(kVThreads % 64 == 0 && kVThreads <= 256, "the compacted kernel packs the thread index into eight bits");
#ifndef MSPA_VIMG
#define MSPA_VIMG 8
#endif
constexpr int kImgPerBlock = MSPA_VIMG;

template <bool HOMOG>
__global__ __launch_bounds__(kVThreads) void vertex_visibility_kernel(const double *__restrict__ xyz,
                                                                      const double *__restrict__ cam_mats,
                                                                      const uint16_t *__restrict__ depth,
                                                                      VertexArgs a) {
    // XCD-aware decode of the 1-D grid: workgroup b runs on XCD b % n_xcd (8 on an MI355X in SPX mode; the host passes 1 when
    // the device reports anything else, which degrades to a plain linear decode), and each XCD has its own L2.  All vertex blocks of one
    // image group go to the same XCD, so a group's depth frames are gathered through ONE L2 instead of eight (measured before
    // the change: 672 MB fetched per 320-image scene against 197 MB of depth frames).
    const uint32_t xcd = blockIdx.x % a.n_xcd;
    const uint32_t slot = blockIdx.x / a.n_xcd;
    const uint32_t vblock = slot % a.vblocks;
    const uint32_t group = (slot / a.vblocks) * a.n_xcd + xcd;
    if (group >= a.igroups) return;
    const int64_t i = (int64_t)vblock * kVThreads + threadIdx.x;
    const bool live = i < a.n_points;
    const int64_t ic = live ? i : a.n_points - 1;
    const int lane = threadIdx.x & 63;
    const double x = xyz[ic * a.point_stride];
    const double y = xyz[ic * a.point_stride + a.comp_stride];
    const double z = xyz[ic * a.point_stride + 2 * a.comp_stride];
    const double w = HOMOG ? xyz[ic * a.point_stride + 3 * a.comp_stride] : 1.0;     // general [N, 4] input (IH:46-72)
    const int img0 = (int)group * kImgPerBlock;
    const int img1 = min(img0 + kImgPerBlock, a.n_images);
    const int64_t dpix = (int64_t)a.dh * a.dw;

    for (int img = img0; img < img1; ++img) {
        const double *__restrict__ Einv = cam_mats + (int64_t)img * (MSPA_CAM_MATS * 16);
        const double *__restrict__ K = Einv + 16;
        const uint16_t *__restrict__ dimg = depth + (int64_t)img * dpix;
        // IH:57-69
        double qx, qy, qz, ix, iy, iz;
        if (HOMOG) {                                       // all four rows of both products, as the reference's 4x4 @ 4xN
            qx = affine_row_w(Einv + 0, x, y, z, w);
            qy = affine_row_w(Einv + 4, x, y, z, w);
            qz = affine_row_w(Einv + 8, x, y, z, w);
            const double qw = affine_row_w(Einv + 12, x, y, z, w);
            ix = affine_row_w(K + 0, qx, qy, qz, qw);
            iy = affine_row_w(K + 4, qx, qy, qz, qw);
            iz = affine_row_w(K + 8, qx, qy, qz, qw);
        } else {
            qx = affine_row(Einv + 0, x, y, z);
            qy = affine_row(Einv + 4, x, y, z);
            qz = affine_row(Einv + 8, x, y, z);
            ix = affine_row(K + 0, qx, qy, qz);
            iy = affine_row(K + 4, qx, qy, qz);
            iz = affine_row(K + 8, qx, qy, qz);
        }
        const double u = ix / iz, v = iy / iz;
        int xi, yi;
        const bool vis = depth_test(live, u, v, qz, dimg, a.dh, a.dw, a.H, a.W, a.sx, a.sy, xi, yi, nullptr, a.dscale);

        const unsigned long long word = __ballot(vis);
        if (lane == 0) {
            if (a.bits && (i < a.n_points)) a.bits[(int64_t)img * a.n_words + (i >> 6)] = word;
            // Only when no bitset is produced: one atomic per wave and image.  With a bitset the counts are
            // a popcount pass over it (bits_count_kernel): 2048 same-address atomics per image serialise in
            // L2 (~12 ns each) and cost 0.7 ms per 64-image batch, 50x the projection itself.
            if (a.count_atomic) {
                const int c = __popcll(word);
                if (c) atomicAdd(a.count_atomic + img, c);
            }
        }
        if (live) {
            const int64_t o = (int64_t)img * a.n_points + i;
            if (a.mask) a.mask[o] = vis ? 1 : 0;
            if (a.uv) {
                a.uv[2 * o + 0] = u;
                a.uv[2 * o + 1] = v;
            }
            if (a.depth_out) a.depth_out[o] = qz;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Fast form (no float64 outputs requested): the two 3x4 products of IH:57-66 are composed ONCE per image --
// M = K * inv(A E), twelve entries, one per thread of the block, parked in LDS -- and a vertex costs 9 FMAs, a
// reciprocal with one Newton step and two multiplies instead of 24 FP64 operations and two IEEE divisions.  Bit-exactness
// is kept exactly as in K3's fast path: a lane whose decisions (half-to-even rounding of the pixel index, the image bounds,
// the strict depth comparison) sit within a guard band of a decision boundary is re-evaluated with the reference chain.
// The band is a BOUND (round 4; mspa_common.h): two float64 evaluation orders of K inv(A E) p differ by at most
// B_k = c 2^-53 (Nabs[k][:3] . |p| + Nabs[k][3]), Nabs = |K| |inv(A E)|, in the k-th homogeneous coordinate; the image record
// carries the magnitudes (slot MSPA_CAM_BOUNDS), from which every lane takes, with s = |x| + |y| + |z| of its vertex,
//   zmin = za s + zb   (camera depth, mm, at or below which the projection is not trusted: the error of u grows like 1 / depth)
//   gz   = ga s + gb   (depth-test guard, mm).
// Needs a pinhole K (third row 0 0 1 0: the third homogeneous coordinate IS the camera depth); any other image takes
// the reference chain for all its lanes.  The composed matrix is scaled by 1000: u and v are unchanged and the third
// coordinate is the camera depth in millimetres, directly comparable with the raw depth sample.
// ---------------------------------------------------------------------------------------------------------
constexpr double kVGuardPx = 1e-6;              // K1's own pixel guard (K3's is tunable: mspa_common.h MSPA_GUARD_PX)

// Per-image guard coefficients (za, zb, ga, gb) from slot MSPA_CAM_BOUNDS of the image record (host-side:
// mspa_camera_bounds_host; round 4's first form computed the 36 multiply-adds of magnitudes in the kernel, one thread per
// image in front of the block's first barrier: K1 +14 %).
__device__ __forceinline__ void image_guard_coefficients(const double *__restrict__ rec, double wh_max, double *out4) {
    const double *__restrict__ b = rec + MSPA_CAM_BOUNDS * 16;
    out4[0] = (2.0 / kVGuardPx) * __builtin_fma(wh_max + 1.0, b[1], b[0]);
    out4[1] = (2.0 / kVGuardPx) * __builtin_fma(wh_max + 1.0, b[3], b[2]);
    out4[2] = 2.0 * b[1];
    out4[3] = __builtin_fma(2.0, b[3], kGuardZmmFloor);
}
#ifndef MSPA_VBATCH
#define MSPA_VBATCH 8
#endif
constexpr int kVBatch = MSPA_VBATCH;             // images whose depth gathers are in flight together

template <bool IDENT>
__global__ __launch_bounds__(kVThreads) void vertex_visibility_fast_kernel(const double *__restrict__ xyz,
                                                                           const double *__restrict__ cam_mats,
                                                                           const uint16_t *__restrict__ depth,
                                                                           VertexArgs a) {
    const uint32_t xcd = blockIdx.x % a.n_xcd;
    const uint32_t slot = blockIdx.x / a.n_xcd;
    const uint32_t vblock = slot % a.vblocks;
    const uint32_t group = (slot / a.vblocks) * a.n_xcd + xcd;
    if (group >= a.igroups) return;
    const int img0 = (int)group * kImgPerBlock;
    const int img1 = min(img0 + kImgPerBlock, a.n_images);

    __shared__ double lds_m[kImgPerBlock][12];
    __shared__ __attribute__((aligned(16))) double lds_g[kImgPerBlock][4];
    __shared__ int lds_pinhole[kImgPerBlock];
    if (threadIdx.x < kImgPerBlock * 12) {
        const int im = threadIdx.x / 12, e = threadIdx.x % 12, r = e / 4, cidx = e % 4;
        if (img0 + im < img1) {
            const double *__restrict__ Einv = cam_mats + (int64_t)(img0 + im) * (MSPA_CAM_MATS * 16);
            const double *__restrict__ K = Einv + 16;
            double acc = K[4 * r + 0] * Einv[0 + cidx];
            acc = __builtin_fma(K[4 * r + 1], Einv[4 + cidx], acc);
            acc = __builtin_fma(K[4 * r + 2], Einv[8 + cidx], acc);
            if (cidx == 3) acc += K[4 * r + 3];
            lds_m[im][e] = acc * 1000.0;
            if (e == 0) lds_pinhole[im] = (K[8] == 0.0 && K[9] == 0.0 && K[10] == 1.0 && K[11] == 0.0) ? 1 : 0;
        }
    } else if (threadIdx.x >= 128 && threadIdx.x < 128 + kImgPerBlock && img0 + (int)(threadIdx.x - 128) < img1) {
        image_guard_coefficients(cam_mats + (int64_t)(img0 + (int)(threadIdx.x - 128)) * (MSPA_CAM_MATS * 16), (double)max(a.W, a.H),
                                 lds_g[threadIdx.x - 128]);
    }
    const int64_t i = (int64_t)vblock * kVThreads + threadIdx.x;
    const bool live = i < a.n_points;
    const int64_t ic = live ? i : a.n_points - 1;
    const int lane = threadIdx.x & 63;
    const double x = xyz[ic * a.point_stride];
    const double y = xyz[ic * a.point_stride + a.comp_stride];
    const double z = xyz[ic * a.point_stride + 2 * a.comp_stride];
    const double psum = (__builtin_fabs(x) + __builtin_fabs(y)) + __builtin_fabs(z);
    const int64_t dpix = (int64_t)a.dh * a.dw;
    const double Wd = (double)a.W, Hd = (double)a.H;
    const int hi_x = a.dw - 1, hi_y = a.dh - 1;
    const uint32_t dw2 = (uint32_t)a.dw * 2u;
    __syncthreads();

    // Images go through in batches of kVBatch: all projections and depth gathers of a batch are issued before the first
    // sample is used, so a wave pays ONE memory round trip per batch instead of one per image (eight dependent round
    // trips per wave left the kernel latency-bound at 8 waves per SIMD).  Lane predicates travel as ballot words (SGPRs).
    for (int b0 = img0; b0 < img1; b0 += kVBatch) {
        double izs[kVBatch];
        uint32_t dv[kVBatch];
        unsigned long long cand_m[kVBatch], risky_m[kVBatch];
#pragma unroll
        for (int q = 0; q < kVBatch; ++q) {
            const int img = min(b0 + q, img1 - 1);               // a ragged last batch repeats its last image (not stored)
            const int im = img - img0;
            const uint16_t *__restrict__ dimg = depth + (int64_t)img * dpix;
            const double *m = lds_m[im];
            const double ix = __builtin_fma(m[0], x, __builtin_fma(m[1], y, __builtin_fma(m[2], z, m[3])));
            const double iy = __builtin_fma(m[4], x, __builtin_fma(m[5], y, __builtin_fma(m[6], z, m[7])));
            const double iz = __builtin_fma(m[8], x, __builtin_fma(m[9], y, __builtin_fma(m[10], z, m[11])));   // mm
            double rz = __builtin_amdgcn_rcp(iz);
            rz = __builtin_fma(__builtin_fma(-iz, rz, 1.0), rz, rz);
            const double u = ix * rz, v = iy * rz;
            const double us = IDENT ? u : u * a.sx, vs = IDENT ? v : v * a.sy;
            const double ru = __builtin_rint(us), rv = __builtin_rint(vs);
            const double zmin = __builtin_fma(lds_g[im][0], psum, lds_g[im][1]);      // this vertex, this image (see the top)
            // saturating conversion (NaN -> 0), clamp by v_med3; the gather goes through a buffer resource of the frame with a
            // 32-bit byte offset (64-bit address arithmetic cost five more instructions per image in an instruction-bound
            // kernel).  The `cand ? offset : 0` below compiles to an exec-masked load: non-candidates (three lanes in four)
            // never reach the address unit.  Letting them gather at their clamped border pixel was 18 % slower, a plain
            // v_cndmask select (all lanes load, non-candidates sample 0) 25 % slower
            int xi, yi;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(xi) : "v"((int)ru), "s"(hi_x));
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(yi) : "v"((int)rv), "s"(hi_y));
            // candidate = what the reference would accept, widened by the guard (a lane inside the widening is risky); within
            // zmin of the camera plane u and v mean nothing: candidate whatever they say, and risky below
            // ... and only if it is also within 2 max(W, H) zmin of the optical axis: nothing else near the plane can be accepted
            const double tn = 2.0 * __builtin_fmax(Wd, Hd) * zmin;
            const bool nearz = !(__builtin_fabs(iz) > zmin) & (__builtin_fabs(ix) < tn) & (__builtin_fabs(iy) < tn);
            const bool cand = live & (((u > -kVGuardPx) & (u < Wd + kVGuardPx) & (v > -kVGuardPx) & (v < Hd + kVGuardPx) &
                                       (iz > zmin)) | nearz);
            // guard < |t| < 0.5 - guard for both coordinates  <=>  max(||tu| - .25|, ||tv| - .25|) < .25 - guard
            const double wu = __builtin_fabs(us - ru) - 0.25, wv = __builtin_fabs(vs - rv) - 0.25;
            unsigned long long rk = __builtin_amdgcn_ballot_w64(!(__builtin_fmax(__builtin_fabs(wu), __builtin_fabs(wv)) < 0.25 - kVGuardPx)) |
                                    __builtin_amdgcn_ballot_w64(nearz);
            if (!IDENT) {   // the bounds are integers of the COLOUR grid, the rounding ties belong to the depth grid
                const double bu = __builtin_fmin(__builtin_fabs(u), __builtin_fabs(u - Wd));
                const double bv = __builtin_fmin(__builtin_fabs(v), __builtin_fabs(v - Hd));
                rk |= __builtin_amdgcn_ballot_w64(!(__builtin_fmin(bu, bv) > kVGuardPx));
            }
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)dimg, 0, (int)(dpix * 2), 0x00020000);
            dv[q] = __builtin_amdgcn_raw_buffer_load_b16(rs, cand ? (int)(__umul24((uint32_t)yi, dw2) + ((uint32_t)xi << 1)) : 0, 0, 0);
            izs[q] = iz;
            cand_m[q] = __builtin_amdgcn_ballot_w64(cand);
            risky_m[q] = rk;
        }
#pragma unroll
        for (int q = 0; q < kVBatch; ++q) {
            const int img = b0 + q;
            if (img >= img1) break;                              // block-uniform
            const int im = img - img0;
            const double sd = izs[q] - (double)dv[q];
            unsigned long long word = cand_m[q] & __builtin_amdgcn_ballot_w64(sd < 0.0);
            const double gz = __builtin_fma(lds_g[im][2], psum, lds_g[im][3]);
            const unsigned long long rk = cand_m[q] & (risky_m[q] | __builtin_amdgcn_ballot_w64(!(__builtin_fabs(sd) > gz)));
            const bool pin = lds_pinhole[im] != 0;               // block-uniform
            if (rk != 0 || !pin) {                               // rare: the reference chain (IH:57-69, 337-386) for those lanes
                const double *__restrict__ Einv = cam_mats + (int64_t)img * (MSPA_CAM_MATS * 16);
                const double *__restrict__ K = Einv + 16;
                const uint16_t *__restrict__ dimg = depth + (int64_t)img * dpix;
                const bool mine = pin ? (((rk >> lane) & 1ull) != 0) : live;
                bool vis = ((word >> lane) & 1ull) != 0;
                if (mine) {
                    const double qx = affine_row(Einv + 0, x, y, z);
                    const double qy = affine_row(Einv + 4, x, y, z);
                    const double qz = affine_row(Einv + 8, x, y, z);
                    const double jx = affine_row(K + 0, qx, qy, qz);
                    const double jy = affine_row(K + 4, qx, qy, qz);
                    const double jz = affine_row(K + 8, qx, qy, qz);
                    int ex, ey;
                    vis = depth_test(true, jx / jz, jy / jz, qz, dimg, a.dh, a.dw, a.H, a.W, a.sx, a.sy, ex, ey);
                }
                word = __builtin_amdgcn_ballot_w64(vis);
            }
            if (lane == 0) {
                if (a.bits && (i < a.n_points)) a.bits[(int64_t)img * a.n_words + (i >> 6)] = word;
                if (a.count_atomic) {
                    const int c = __popcll(word);
                    if (c) atomicAdd(a.count_atomic + img, c);
                }
            }
            if (live && a.mask) a.mask[(int64_t)img * a.n_points + i] = (uint8_t)((word >> lane) & 1ull);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Compacted form of the fast kernel.  Three (vertex, image) pairs in four fail the frustum test, yet on an unordered vertex
// array every wave holds some candidate for every image, so no wave-level early-out ever fires and the fast kernel spends
// its ~55 VALU issues per (wave, image) on mostly idle lanes -- and it is instruction-bound.  Here the block works in phases:
//   A  every thread decides candidacy of its vertex for the block's images on the homogeneous triple, BEFORE the division
//      (iz > 0:  u > -b  <=>  ix > -b iz;  u < W + b  <=>  ix < (W + b) iz), 17 issues per image, and the candidates
//      (vertex, image) are compacted into an LDS list (ballot + mbcnt ranks, one LDS atomic per wave);
//   B  the threads walk the dense list: projection again for the listed pair (vertex from LDS, matrix row set of its image
//      from LDS), reciprocal, rounding, guards, depth gather (one buffer resource spans the block's frames), depth test;
//      guarded lanes take the reference chain exactly as before; visible pairs OR their bit into an LDS bit table;
//   C  the table leaves as the same coalesced bitset words / byte mask / counts.
// The tie guard of phase B is TWICE the candidate band of phase A, so a lane the band let in from just outside the image
// (|u| < b) is always re-evaluated; lanes with |iz| under the depth guard and images with a non-pinhole K are listed
// unconditionally and re-evaluated.  Same integers as the other two kernels (tests/test_gpu_parity.py: goldens, adversarial
// cameras, ScanNet's two grids, non-pinhole K).
// ---------------------------------------------------------------------------------------------------------
#ifndef MSPA_VCOMPACT
#define MSPA_VCOMPACT 1
#endif
#ifndef MSPA_V_NT
#define MSPA_V_NT 0
#endif
#ifndef MSPA_VCOMPACT_ENTRIES
#define MSPA_VCOMPACT_ENTRIES 1
#endif
constexpr double kVBandPx = kVGuardPx;       // phase A: candidate band around the image, on the homogeneous coordinates
constexpr double kVTiePx = 2.0 * kVGuardPx;  // phase B: tie / bound guard on the divided coordinates

#ifndef MSPA_VCOMPACT_WAVES
#define MSPA_VCOMPACT_WAVES 8
#endif
template <bool IDENT>
__global__ __launch_bounds__(kVThreads)
#if MSPA_VCOMPACT_WAVES
__attribute__((amdgpu_waves_per_eu(MSPA_VCOMPACT_WAVES, MSPA_VCOMPACT_WAVES)))
#endif
void vertex_visibility_compact_kernel(const double *__restrict__ xyz,
                                                                              const double *__restrict__ cam_mats,
                                                                              const uint16_t *__restrict__ depth,
                                                                              VertexArgs a) {
    const uint32_t xcd = blockIdx.x % a.n_xcd;
    const uint32_t slot = blockIdx.x / a.n_xcd;
    const uint32_t vblock = slot % a.vblocks;
    const uint32_t group = (slot / a.vblocks) * a.n_xcd + xcd;
    if (group >= a.igroups) return;
    const int img0 = (int)group * kImgPerBlock;
    const int img1 = min(img0 + kImgPerBlock, a.n_images);
    const int nimg = img1 - img0;
    const int tid = threadIdx.x, lane = tid & 63;

    __shared__ __attribute__((aligned(16))) double lds_m[kImgPerBlock][12];
    __shared__ __attribute__((aligned(16))) double lds_g[4];         // guard coefficients: the LARGEST over the block's images
    __shared__ int lds_pinhole[kImgPerBlock];
    __shared__ __attribute__((aligned(16))) double lds_xyz[kVThreads][3];
    __shared__ uint16_t lds_list[kVThreads * kImgPerBlock];          // tid | image << 8 | (near the camera plane and axis) << 15
    __shared__ double lds_gz[kVThreads];                              // per vertex: depth-test guard (mm)
    static_assert(kImgPerBlock <= 64, "list entry: 6 bits of image index under the flag bit");
    __shared__ uint32_t lds_bits[kImgPerBlock][kVThreads / 32];
    __shared__ uint32_t lds_n;
    if (tid < kImgPerBlock * 12) {
        const int im = tid / 12, e = tid % 12, r = e / 4, cidx = e % 4;
        if (im < nimg) {
            const double *__restrict__ Einv = cam_mats + (int64_t)(img0 + im) * (MSPA_CAM_MATS * 16);
            const double *__restrict__ K = Einv + 16;
            double acc = K[4 * r + 0] * Einv[0 + cidx];
            acc = __builtin_fma(K[4 * r + 1], Einv[4 + cidx], acc);
            acc = __builtin_fma(K[4 * r + 2], Einv[8 + cidx], acc);
            if (cidx == 3) acc += K[4 * r + 3];
            lds_m[im][e] = acc * 1000.0;
            if (e == 0) lds_pinhole[im] = (K[8] == 0.0 && K[9] == 0.0 && K[10] == 1.0 && K[11] == 0.0) ? 1 : 0;
        }
    }
    // Guard coefficients: lane q of the block's LAST wave reads image q's four numbers and folds them into ONE set per block
    // with an LDS atomic maximum on their bit patterns (they are non-negative doubles: the unsigned order of the bits is
    // their order) -- a lane's zmin then costs one FMA per block instead of one per image, at the price of the widest
    // image's band for all eight.  (A shuffle reduction here -- 24 ds_bpermute in front of the block's first barrier --
    // cost K1 8-10 %: tools/ab_k1.py.)  The wave's LDS operations execute in order: the zeros land before the maxima.
    if (tid >= kVThreads - 64) {
        const int q = tid - (kVThreads - 64);
        if (q < 4) lds_g[q] = 0.0;
        if (q < nimg) {
            double g4[4];
            image_guard_coefficients(cam_mats + (int64_t)(img0 + q) * (MSPA_CAM_MATS * 16), (double)max(a.W, a.H), g4);
            // ds_max_u64 spelled out: through atomicMax() LLVM's atomic optimizer turns four same-address LDS atomics into four
            // serial readlane scans over the active lanes (~2 000 cycles in front of the barrier); the LDS unit resolves eight
            // lanes on one address in a few cycles itself
            typedef __attribute__((address_space(3))) double lds_double_t;
            const uint32_t base = (uint32_t)(uintptr_t)(lds_double_t *)&lds_g[0];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                asm volatile("ds_max_u64 %0, %1 offset:%2" ::"v"(base), "v"(__double_as_longlong(g4[k])), "n"(8 * k) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the compiler's scoreboard does not see the asm's LDS operations
    }
    if (tid < kImgPerBlock * (kVThreads / 32)) (&lds_bits[0][0])[tid] = 0u;
    if (tid == 0) lds_n = 0u;
    const int64_t i = (int64_t)vblock * kVThreads + tid;
    const bool live = i < a.n_points;
    const int64_t ic = live ? i : a.n_points - 1;
    // MSPA_V_NT (off): the vertex array is streamed -- every block reads its 256 vertices once -- so non-temporal loads
    // should keep its 3 MB per image group from displacing the eight depth frames the XCD's blocks gather from.  Measured
    // (tools/ab_k1.py, round 3): 0.103 vs 0.092 ms on the shuffled cloud, 0.092 vs 0.077 in Morton order -- slower; so is
    // 16 images per block (0.112 / 0.079).  The 200 MB working set lives in the 256 MB Infinity Cache either way.
#if MSPA_V_NT
    const double x = __builtin_nontemporal_load(&xyz[ic * a.point_stride]);
    const double y = __builtin_nontemporal_load(&xyz[ic * a.point_stride + a.comp_stride]);
    const double z = __builtin_nontemporal_load(&xyz[ic * a.point_stride + 2 * a.comp_stride]);
#else
    const double x = xyz[ic * a.point_stride];
    const double y = xyz[ic * a.point_stride + a.comp_stride];
    const double z = xyz[ic * a.point_stride + 2 * a.comp_stride];
#endif
    lds_xyz[tid][0] = x;
    lds_xyz[tid][1] = y;
    lds_xyz[tid][2] = z;
    const double psum = (__builtin_fabs(x) + __builtin_fabs(y)) + __builtin_fabs(z);
    const unsigned long long live_m = __builtin_amdgcn_ballot_w64(live);
    const double Wd = (double)a.W, Hd = (double)a.H;
    __syncthreads();
    // block-uniform coefficients -> scalar registers; this lane's camera-depth threshold for all of the block's images
    double gc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned long long b = __double_as_longlong(lds_g[k]);
        gc[k] = __longlong_as_double(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)b));
    }
    // this lane's vertex, any image of the block: camera-depth threshold, its "near the optical axis" companion (homogeneous
    // x, y within 2 max(W, H) zmin: nothing else near the camera plane can be accepted by any evaluation order: 0 <= x < W z with
    // z <= zmin + B_2), and the depth-test guard, which phase B reads back per list entry
    const double zmin_l = __builtin_fma(gc[0], psum, gc[1]);
    const double tn_l = 2.0 * (double)max(a.W, a.H) * zmin_l;
    lds_gz[tid] = __builtin_fma(gc[2], psum, gc[3]);

    // ---- phase A: candidates of this wave's 64 vertices for the block's images ----
    unsigned long long cm[kImgPerBlock], nm[kImgPerBlock];
    int total = 0;
#pragma unroll
    for (int q = 0; q < kImgPerBlock; ++q) {
        unsigned long long c = 0;
        if (q < nimg) {                                               // block-uniform
            const double *m = lds_m[q];
            const double ix = __builtin_fma(m[0], x, __builtin_fma(m[1], y, __builtin_fma(m[2], z, m[3])));
            const double iy = __builtin_fma(m[4], x, __builtin_fma(m[5], y, __builtin_fma(m[6], z, m[7])));
            const double iz = __builtin_fma(m[8], x, __builtin_fma(m[9], y, __builtin_fma(m[10], z, m[11])));   // mm
            const double gb = kVBandPx * iz;
            const double zmin = zmin_l;
            unsigned long long near0 = __builtin_amdgcn_ballot_w64(!(iz > zmin));               // NaN lands here too
            // near the plane AND near the axis (rare: a wave-uniform branch).  Listing every vertex near the plane of SOME image --
            // 0.5 % of the list entries: one phase-B wave trip in four ran the reference chain -- had cost K1 9 %.
            if (near0 != 0ull)
                near0 &= __builtin_amdgcn_ballot_w64(__builtin_fabs(ix) < tn_l) & __builtin_amdgcn_ballot_w64(__builtin_fabs(iy) < tn_l);
            nm[q] = near0;
            const unsigned long long inside =
                __builtin_amdgcn_ballot_w64(ix > -gb) & __builtin_amdgcn_ballot_w64(ix < (Wd + kVBandPx) * iz) &
                __builtin_amdgcn_ballot_w64(iy > -gb) & __builtin_amdgcn_ballot_w64(iy < (Hd + kVBandPx) * iz);
            c = lds_pinhole[q] ? (live_m & __builtin_amdgcn_ballot_w64(!(iz <= -zmin)) & (near0 | inside)) : live_m;
        }
        else nm[q] = 0;
        cm[q] = c;
        total += __popcll(c);
    }
    uint32_t run = 0;
    if (total) {                                                      // wave-uniform
        if (lane == 0) run = atomicAdd(&lds_n, (uint32_t)total);
        run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
#pragma unroll
        for (int q = 0; q < kImgPerBlock; ++q) {
            if (cm[q] == 0) continue;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm[q], 0u));
            // lane predicates straight from the scalar masks (inverse ballot: exec mask / v_cndmask operand, no shifts)
            if (__builtin_amdgcn_inverse_ballot_w64(cm[q]))
                lds_list[run + rank] = (uint16_t)((tid | (q << 8)) | (__builtin_amdgcn_inverse_ballot_w64(nm[q]) ? 0x8000 : 0));
            run += (uint32_t)__popcll(cm[q]);
        }
    }
    __syncthreads();

    // ---- phase B: the dense list ----
    const uint32_t n = lds_n;
    const int64_t dpix = (int64_t)a.dh * a.dw;
    const int hi_x = a.dw - 1, hi_y = a.dh - 1;
    const uint32_t dw2 = (uint32_t)a.dw * 2u, dpix2 = (uint32_t)(dpix * 2);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void *)(depth + (int64_t)img0 * dpix), 0, (int)((int64_t)nimg * dpix * 2), 0x00020000);
    constexpr int kEnt = MSPA_VCOMPACT_ENTRIES;                       // list entries per thread and trip: their gathers fly together
    for (uint32_t e0 = 0; e0 < n; e0 += kVThreads * kEnt) {           // block-uniform trip count
        bool active[kEnt], risky[kEnt];
        uint32_t vv[kEnt], qq[kEnt], d[kEnt];
        double pxs[kEnt], pys[kEnt], pzs[kEnt], izs[kEnt], gzs[kEnt];
#pragma unroll
        for (int k = 0; k < kEnt; ++k) {
            const uint32_t e = e0 + (uint32_t)(k * kVThreads + tid);
            active[k] = e < n;
            const uint32_t ent = active[k] ? (uint32_t)lds_list[e] : 0u;
            const uint32_t v = ent & 255u, q = (ent >> 8) & 63u;
            const bool nearp = (ent >> 15) != 0u;
            vv[k] = v;
            qq[k] = q;
            const double px = lds_xyz[v][0], py = lds_xyz[v][1], pz = lds_xyz[v][2];
            pxs[k] = px;
            pys[k] = py;
            pzs[k] = pz;
            const double *m = lds_m[q];
            const double ix = __builtin_fma(m[0], px, __builtin_fma(m[1], py, __builtin_fma(m[2], pz, m[3])));
            const double iy = __builtin_fma(m[4], px, __builtin_fma(m[5], py, __builtin_fma(m[6], pz, m[7])));
            const double iz = __builtin_fma(m[8], px, __builtin_fma(m[9], py, __builtin_fma(m[10], pz, m[11])));
            izs[k] = iz;
            double rz = __builtin_amdgcn_rcp(iz);
            rz = __builtin_fma(__builtin_fma(-iz, rz, 1.0), rz, rz);
            const double u = ix * rz, w = iy * rz;
            const double us = IDENT ? u : u * a.sx, vs = IDENT ? w : w * a.sy;
            const double ru = __builtin_rint(us), rv = __builtin_rint(vs);
            int xi, yi;                                               // saturating conversion; NaN -> 0
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(xi) : "v"((int)ru), "s"(hi_x));
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(yi) : "v"((int)rv), "s"(hi_y));
            const double wu = __builtin_fabs(us - ru) - 0.25, wv = __builtin_fabs(vs - rv) - 0.25;
            gzs[k] = lds_gz[v];
            // entries phase A flagged (within zmin of the camera plane and near the optical axis: the projection itself is not
            // trusted there) take the reference chain; every other entry has a camera depth above its zmin
            bool rk = !(__builtin_fmax(__builtin_fabs(wu), __builtin_fabs(wv)) < 0.25 - kVTiePx) | nearp | (lds_pinhole[q] == 0);
            if (!IDENT) {   // the bounds are integers of the COLOUR grid, the rounding ties belong to the depth grid
                const double bu = __builtin_fmin(__builtin_fabs(u), __builtin_fabs(u - Wd));
                const double bv = __builtin_fmin(__builtin_fabs(w), __builtin_fabs(w - Hd));
                rk |= !(__builtin_fmin(bu, bv) > kVTiePx);
            }
            risky[k] = rk;
            d[k] = 0;
            if (active[k])
                d[k] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)(q * dpix2 + __umul24((uint32_t)yi, dw2) + ((uint32_t)xi << 1)), 0, 0);
        }
#pragma unroll
        for (int k = 0; k < kEnt; ++k) {
            const double sd = izs[k] - (double)d[k];
            bool vis = sd < 0.0;
            const bool rk = active[k] & (risky[k] | !(__builtin_fabs(sd) > gzs[k]));
#ifdef MSPA_EXPERIMENT_NOCOLD   // timing only (wrong results for guarded lanes)
            if (false) {
#else
            if (__builtin_amdgcn_ballot_w64(rk) != 0ull) {            // rare: the reference chain (IH:57-69, 337-386)
#endif
                if (rk) {
                    const int img = img0 + (int)qq[k];
                    const double *__restrict__ Einv = cam_mats + (int64_t)img * (MSPA_CAM_MATS * 16);
                    const double *__restrict__ K = Einv + 16;
                    const double qx = affine_row(Einv + 0, pxs[k], pys[k], pzs[k]);
                    const double qy = affine_row(Einv + 4, pxs[k], pys[k], pzs[k]);
                    const double qz = affine_row(Einv + 8, pxs[k], pys[k], pzs[k]);
                    const double jx = affine_row(K + 0, qx, qy, qz);
                    const double jy = affine_row(K + 4, qx, qy, qz);
                    const double jz = affine_row(K + 8, qx, qy, qz);
                    int ex, ey;
                    vis = depth_test(true, jx / jz, jy / jz, qz, depth + (int64_t)img * dpix, a.dh, a.dw, a.H, a.W, a.sx, a.sy, ex, ey);
                }
            }
            if (active[k] & vis) atomicOr(&lds_bits[qq[k]][vv[k] >> 5], 1u << (vv[k] & 31u));
        }
    }
    __syncthreads();

    // ---- phase C: the block's 256 x nimg bits leave as bitset words / byte mask / counts ----
    constexpr int kWordsPerBlock = kVThreads / 64;
    if (a.bits && tid < nimg * kWordsPerBlock) {
        const int q = tid / kWordsPerBlock, w = tid % kWordsPerBlock;
        const int64_t word = (int64_t)vblock * (kVThreads / 64) + w;
        if (word < a.n_words)
            a.bits[(int64_t)(img0 + q) * a.n_words + word] =
                (uint64_t)lds_bits[q][2 * w] | ((uint64_t)lds_bits[q][2 * w + 1] << 32);
    }
    if (a.mask && live) {
        for (int q = 0; q < nimg; ++q)
            a.mask[(int64_t)(img0 + q) * a.n_points + i] = (uint8_t)((lds_bits[q][tid >> 5] >> (tid & 31)) & 1u);
    }
    if (a.count_atomic && tid < nimg) {
        int c = 0;
        for (int k = 0; k < kVThreads / 32; ++k) c += __popc(lds_bits[tid][k]);
        if (c) atomicAdd(a.count_atomic + img0 + tid, c);
    }
}

// visible vertices per image = popcount of its bitset: one wave per image
__global__ __launch_bounds__(kVThreads) void bits_count_kernel(const uint64_t *__restrict__ bits, int64_t n_words,
                                                               int n_images, int32_t *__restrict__ count) {
    const int img = blockIdx.x * (kVThreads / kWave) + (threadIdx.x >> 6);
    if (img >= n_images) return;
    const int lane = threadIdx.x & 63;
    const uint64_t *__restrict__ row = bits + (int64_t)img * n_words;
    int c = 0;
#pragma unroll 8
    for (int64_t w = lane; w < n_words; w += kWave) c += __popcll(row[w]);     // unrolled: eight loads in flight per lane
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if (lane == 0) count[img] = c;
}

// a3/a4/a5 on already-projected points (the reference exposes them as separate methods, IH:337-386)
__global__ __launch_bounds__(kVThreads) void check_visibility_kernel(const double *__restrict__ uv,
                                                                     const double *__restrict__ pdepth, int64_t n,
                                                                     const uint16_t *__restrict__ dimg, int dh, int dw,
                                                                     int H, int W, double sx, double sy, double dscale,
                                                                     uint8_t *__restrict__ inb_out,
                                                                     uint8_t *__restrict__ byd_out,
                                                                     uint8_t *__restrict__ vis_out) {
    const int64_t i = (int64_t)blockIdx.x * kVThreads + threadIdx.x;
    if (i >= n) return;
    const double u = uv[2 * i], v = uv[2 * i + 1], d = pdepth ? pdepth[i] : 0.0;
    const bool inb = (u >= 0.0) && (u < (double)W) && (v >= 0.0) && (v < (double)H);   // IH:342-343
    bool byd = false;
    if (dimg) {                                                                         // IH:359-371
        const int xi = round_clip(u * sx, dw - 1), yi = round_clip(v * sy, dh - 1);
        const double dv = (double)dimg[yi * dw + xi] * dscale;                          // IH:368
        byd = (d > 0.0) && (d < dv);
    }
    if (inb_out) inb_out[i] = inb ? 1 : 0;
    if (byd_out) byd_out[i] = byd ? 1 : 0;
    if (vis_out) vis_out[i] = (inb && byd) ? 1 : 0;
}

}  // namespace mspa

using namespace mspa;

static bool scale_ok(double s) { return s > 0.0 && s < 1e300; }     // finite, positive (NaN fails both)

extern "C" int mspa_check_visibility_ex(const double *uv, const double *point_depth, int64_t n,
                                        const uint16_t *depth_image, int32_t dh, int32_t dw, int32_t H, int32_t W,
                                        double depth_value_scale, uint8_t *out_in_bounds, uint8_t *out_by_depth,
                                        uint8_t *out_visible, mspa_stream_t stream) {
    if (!scale_ok(depth_value_scale)) return fail(MSPA_EINVAL, "mspa_check_visibility: depth_value_scale must be positive and finite");
    if (n < 0) return fail(MSPA_EINVAL, "mspa_check_visibility: bad count");
    if (n == 0) return MSPA_OK;
    if (!uv) return fail(MSPA_EINVAL, "mspa_check_visibility: null uv");
    if ((out_by_depth || out_visible) && (!depth_image || !point_depth))
        return fail(MSPA_EINVAL, "mspa_check_visibility: depth test needs point_depth and depth_image");
    if (H < 2 || W < 2 || H > 32767 || W > 32767 || (depth_image && (dh < 2 || dw < 2 || dh > 32767 || dw > 32767)))
        return fail(MSPA_EINVAL, "mspa_check_visibility: image size out of range [2, 32767]");
    const int64_t blocks = (n + kVThreads - 1) / kVThreads;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_check_visibility: too many points; split the batch");
    const double sx = depth_image ? (double)dw / (double)W : 1.0, sy = depth_image ? (double)dh / (double)H : 1.0;
    hipLaunchKernelGGL(check_visibility_kernel, dim3((uint32_t)blocks), dim3(kVThreads), 0, (hipStream_t)stream, uv,
                       point_depth, n, depth_image, dh, dw, H, W, sx, sy, depth_value_scale, out_in_bounds, out_by_depth,
                       out_visible);
    return check_hip(hipGetLastError(), "check_visibility_kernel launch");
}

extern "C" int mspa_check_visibility(const double *uv, const double *point_depth, int64_t n,
                                     const uint16_t *depth_image, int32_t dh, int32_t dw, int32_t H, int32_t W,
                                     uint8_t *out_in_bounds, uint8_t *out_by_depth, uint8_t *out_visible,
                                     mspa_stream_t stream) {
    return mspa_check_visibility_ex(uv, point_depth, n, depth_image, dh, dw, H, W, 0.001, out_in_bounds, out_by_depth,
                                    out_visible, stream);
}


extern "C" int mspa_vertex_visibility_ex(const double *xyz, int64_t n_points, int64_t point_stride,
                                         int64_t comp_stride, int32_t homogeneous, const double *cam_mats, int32_t n_images,
                                         const uint16_t *depth, int32_t dh, int32_t dw, int32_t H, int32_t W,
                                         double depth_value_scale, uint64_t *out_bits, uint8_t *out_mask, double *out_uv,
                                         double *out_depth, int32_t *out_count, mspa_stream_t stream) {
    if (!scale_ok(depth_value_scale)) return fail(MSPA_EINVAL, "mspa_vertex_visibility: depth_value_scale must be positive and finite");
    if (n_points < 0 || n_images < 0 || point_stride <= 0 || comp_stride <= 0)
        return fail(MSPA_EINVAL, "mspa_vertex_visibility: bad count or stride");
    if ((!xyz && n_points > 0) || ((!cam_mats || !depth) && n_images > 0))
        return fail(MSPA_EINVAL, "mspa_vertex_visibility: null input pointer");
    if (dh < 2 || dw < 2 || H < 2 || W < 2 || dh > 32767 || dw > 32767 || H > 32767 || W > 32767)
        return fail(MSPA_EINVAL, "mspa_vertex_visibility: image size out of range [2, 32767]");
    hipStream_t s = (hipStream_t)stream;
    const bool count_from_bits = out_count && out_bits;
    if (out_count && n_images > 0 && (!count_from_bits || n_points == 0)) {
        int rc = check_hip(hipMemsetAsync(out_count, 0, sizeof(int32_t) * n_images, s), "hipMemsetAsync(count)");
        if (rc) return rc;
    }
    if (n_points == 0 || n_images == 0) return MSPA_OK;
    VertexArgs a;
    a.n_points = n_points; a.point_stride = point_stride; a.comp_stride = comp_stride;
    a.n_images = n_images; a.dh = dh; a.dw = dw; a.H = H; a.W = W;
    a.sx = (double)dw / (double)W;
    a.sy = (double)dh / (double)H;
    a.n_words = (n_points + 63) / 64;
    a.bits = out_bits; a.mask = out_mask; a.uv = out_uv; a.depth_out = out_depth;
    a.count_atomic = count_from_bits ? nullptr : out_count;
    const int64_t bx = (n_points + kVThreads - 1) / kVThreads;
    const int64_t by = (n_images + kImgPerBlock - 1) / kImgPerBlock;
    const int64_t n_xcd = xcd_count();
    const int64_t blocks = bx * ((by + n_xcd - 1) / n_xcd) * n_xcd;
    if (bx > 0x7fffffffLL || blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_vertex_visibility: batch too large; split it");
    a.vblocks = (uint32_t)bx;
    a.igroups = (uint32_t)by;
    a.n_xcd = (uint32_t)n_xcd;
    a.dscale = depth_value_scale;
    // float64 outputs are DEFINED as the reference's operation order; everything else (bitset, byte mask, counts) takes the
    // composed + guarded kernel, which reproduces the same integers
    const bool compact = MSPA_VCOMPACT && (int64_t)kImgPerBlock * dh * dw * 2 < 0x7fffffffLL && kImgPerBlock * 12 <= kVThreads;
    // ... and so do a general homogeneous coordinate and a depth scale other than the millimetre the composed kernels fold in
    if (homogeneous)
        hipLaunchKernelGGL(vertex_visibility_kernel<true>, dim3((uint32_t)blocks), dim3(kVThreads), 0, s, xyz, cam_mats, depth, a);
    else if (out_uv || out_depth || depth_value_scale != 0.001)
        hipLaunchKernelGGL(vertex_visibility_kernel<false>, dim3((uint32_t)blocks), dim3(kVThreads), 0, s, xyz, cam_mats, depth, a);
    else if (compact && dh == H && dw == W)
        hipLaunchKernelGGL(vertex_visibility_compact_kernel<true>, dim3((uint32_t)blocks), dim3(kVThreads), 0, s, xyz, cam_mats, depth, a);
    else if (compact)
        hipLaunchKernelGGL(vertex_visibility_compact_kernel<false>, dim3((uint32_t)blocks), dim3(kVThreads), 0, s, xyz, cam_mats, depth, a);
    else if (dh == H && dw == W)
        hipLaunchKernelGGL(vertex_visibility_fast_kernel<true>, dim3((uint32_t)blocks), dim3(kVThreads), 0, s, xyz, cam_mats, depth, a);
    else
        hipLaunchKernelGGL(vertex_visibility_fast_kernel<false>, dim3((uint32_t)blocks), dim3(kVThreads), 0, s, xyz, cam_mats, depth, a);
    int rc = check_hip(hipGetLastError(), "vertex_visibility_kernel launch");
    if (rc || !count_from_bits) return rc;
    const int per_block = kVThreads / kWave;
    hipLaunchKernelGGL(bits_count_kernel, dim3((uint32_t)((n_images + per_block - 1) / per_block)), dim3(kVThreads), 0, s,
                       out_bits, a.n_words, n_images, out_count);
    return check_hip(hipGetLastError(), "bits_count_kernel launch");
}

extern "C" int mspa_vertex_visibility(const double *xyz, int64_t n_points, int64_t point_stride,
                                      int64_t comp_stride, const double *cam_mats, int32_t n_images,
                                      const uint16_t *depth, int32_t dh, int32_t dw, int32_t H, int32_t W,
                                      uint64_t *out_bits, uint8_t *out_mask, double *out_uv,
                                      double *out_depth, int32_t *out_count, mspa_stream_t stream) {
    return mspa_vertex_visibility_ex(xyz, n_points, point_stride, comp_stride, 0, cam_mats, n_images, depth, dh, dw, H, W,
                                     0.001, out_bits, out_mask, out_uv, out_depth, out_count, stream);
}
