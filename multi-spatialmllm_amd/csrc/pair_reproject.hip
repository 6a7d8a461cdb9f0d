// K3: frame-pair back-projection -> reprojection -> depth-buffer visibility (see include/mspa.h).
//
// Mapping.  A workgroup of 256 threads (4 waves) owns a strip of kIters*256 consecutive pixels of
// ONE pair; lanes take consecutive pixels so every depth/colour read and every output write is a
// contiguous run per wave.  Blocks of one pair are numbered so that they land on one XCD
// (hardware dispatches block b to XCD b % 8): the frame-2 depth image the pair gathers from
// (600 KB) then stays in that XCD's 4 MB L2.  The five 3x4 matrices of the pair are read through
// wave-uniform addresses (scalar loads into SGPRs, the operand v_fma_f64 takes for free); the
// per-pixel chain lives in VGPRs.  No MFMA: this is point geometry, HBM/FP64-VALU bound.
#include "mspa_common.h"

namespace mspa {

// Inputs are separate `const T *__restrict__` kernel parameters (not struct members) on purpose:
// only then can the compiler prove the wave-uniform matrix/pair reads are never clobbered by the
// kernel's own stores and issue them as scalar loads (s_load_dwordx8 -> SGPRs).
struct PairArgs {
    int64_t n_pairs;
    int dh, dw, H, W;
    uint32_t P;            // H*W
    uint32_t div_magic;    // floor(2^32 / W) + 1 : i / W == umulhi(i, magic) for i*W < 2^32
    double sx, sy;         // dw / W, dh / H  (IH:359-360, OPS:272-273)
    int strips;            // strips per pair
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
};

constexpr int kThreads = 256;
constexpr int kIters = 16;                       // 4096 pixels per workgroup
constexpr int kStrip = kThreads * kIters;

template <bool IDENT>
__global__ __launch_bounds__(kThreads) void pair_reproject_kernel(const uint16_t *__restrict__ depth,
                                                                  const uint8_t *__restrict__ rgb,
                                                                  const double *__restrict__ mats,
                                                                  const int32_t *__restrict__ pairs, PairArgs a) {
    // XCD-aware decode: xcd = b % 8 picks the pair within a group of 8, the rest walks the strips.
    const uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u;
    const uint32_t k = b >> 3;
    const uint32_t strip = k % (uint32_t)a.strips;
    const int64_t pair = (int64_t)(k / (uint32_t)a.strips) * 8 + xcd;
    if (pair >= a.n_pairs) return;

    const int f1 = pairs[2 * pair + 0];
    const int f2 = pairs[2 * pair + 1];
    const double *__restrict__ m1 = mats + (int64_t)f1 * (MSPA_FRAME_MATS * 16);
    const double *__restrict__ m2 = mats + (int64_t)f2 * (MSPA_FRAME_MATS * 16);
    const double *__restrict__ Kinv = m1 + MSPA_MAT_KINV * 16;
    const double *__restrict__ E1 = m1 + MSPA_MAT_E * 16;
    const double *__restrict__ A = m1 + MSPA_MAT_A * 16;
    const double *__restrict__ Einv2 = m2 + MSPA_MAT_EINV_ALIGNED * 16;
    const double *__restrict__ K = m2 + MSPA_MAT_K * 16;

    const int64_t dpix = (int64_t)a.dh * a.dw;
    const uint16_t *__restrict__ depth1 = depth + (int64_t)f1 * dpix;
    const uint16_t *__restrict__ depth2 = depth + (int64_t)f2 * dpix;
    const uint8_t *__restrict__ rgb1 = rgb ? rgb + (int64_t)f1 * a.P * 3 : nullptr;
    const int64_t obase = pair * (int64_t)a.P;
    const int lane = threadIdx.x & 63;

    int n_valid = 0, n_vis = 0;
    const uint32_t i0 = strip * (uint32_t)kStrip + threadIdx.x;
#pragma unroll 2
    for (int it = 0; it < kIters; ++it) {
        const uint32_t i = i0 + (uint32_t)it * kThreads;
        const bool in_img = i < a.P;
        const uint32_t ic = in_img ? i : a.P - 1;
        const uint32_t my = __umulhi(ic, a.div_magic);
        const uint32_t mx = ic - my * (uint32_t)a.W;

        uint32_t d16;
        if (IDENT) {
            d16 = depth1[ic];
        } else {
            const int dy = round_clip((double)my * a.sy, a.dh - 1);   // OPS:285-290
            const int dx = round_clip((double)mx * a.sx, a.dw - 1);
            d16 = depth1[dy * a.dw + dx];
        }
        const double d = (double)d16 * 0.001;                         // OPS:292-294
        const bool valid = in_img && (d > 0.0);                       // OPS:297

        // OPS:303-320: pixel ray -> camera -> world -> aligned
        const double px = (double)mx * d, py = (double)my * d;
        const double cx = affine_row(Kinv + 0, px, py, d);
        const double cy = affine_row(Kinv + 4, px, py, d);
        const double cz = affine_row(Kinv + 8, px, py, d);
        const double wx = affine_row(E1 + 0, cx, cy, cz);
        const double wy = affine_row(E1 + 4, cx, cy, cz);
        const double wz = affine_row(E1 + 8, cx, cy, cz);
        const double ax = affine_row(A + 0, wx, wy, wz);
        const double ay = affine_row(A + 4, wx, wy, wz);
        const double az = affine_row(A + 8, wx, wy, wz);
        // IH:57-69: aligned world -> camera 2 -> image 2
        const double qx = affine_row(Einv2 + 0, ax, ay, az);
        const double qy = affine_row(Einv2 + 4, ax, ay, az);
        const double qz = affine_row(Einv2 + 8, ax, ay, az);
        const double ix = affine_row(K + 0, qx, qy, qz);
        const double iy = affine_row(K + 4, qx, qy, qz);
        const double iz = affine_row(K + 8, qx, qy, qz);
        const double u = ix / iz, v = iy / iz;

        int xi, yi;
        const bool vis = depth_test(valid, u, v, qz, depth2, a.dh, a.dw, a.H, a.W, a.sx, a.sy, xi, yi);
        n_valid += valid ? 1 : 0;
        n_vis += vis ? 1 : 0;

        const unsigned long long vmask = __ballot(vis);
        if (a.vis_bits && lane == 0 && (i - lane) < a.P)
            a.vis_bits[pair * (int64_t)((a.P + 63) >> 6) + ((i - lane) >> 6)] = vmask;
        if (in_img) {
            const int64_t o = obase + i;
            const double nan = __builtin_nan("");
            if (a.vis_u8) a.vis_u8[o] = vis ? 1 : 0;
            if (a.valid_u8) a.valid_u8[o] = valid ? 1 : 0;
            if (a.pix_i16) {
                const uint32_t packed = valid ? ((uint32_t)(uint16_t)xi | ((uint32_t)(uint16_t)yi << 16)) : 0xFFFFFFFFu;
                reinterpret_cast<uint32_t *>(a.pix_i16)[o] = packed;
            }
            if (a.xyz_f32) {
                float *q = a.xyz_f32 + 3 * o;
                const float fn = __builtin_nanf("");
                q[0] = valid ? (float)ax : fn;
                q[1] = valid ? (float)ay : fn;
                q[2] = valid ? (float)az : fn;
            }
            if (a.rgba) {
                uint32_t c = 0;
                if (rgb1) {
                    const uint8_t *s = rgb1 + 3 * (int64_t)i;
                    c = (uint32_t)s[0] | ((uint32_t)s[1] << 8) | ((uint32_t)s[2] << 16);
                }
                a.rgba[o] = c | (valid ? 0xFF000000u : 0u);
            }
            if (a.xyz_f64) {
                double *q = a.xyz_f64 + 3 * o;
                q[0] = valid ? ax : nan;
                q[1] = valid ? ay : nan;
                q[2] = valid ? az : nan;
            }
            if (a.uv_f64) {
                a.uv_f64[2 * o + 0] = valid ? u : nan;
                a.uv_f64[2 * o + 1] = valid ? v : nan;
            }
            if (a.depth_f64) a.depth_f64[o] = valid ? qz : nan;
        }
    }

    if (a.counts) {
        // wave reduce (DPP/bpermute shuffles), then one LDS step and two atomics per workgroup
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
}

}  // namespace mspa

using namespace mspa;

extern "C" int mspa_pair_reproject(const uint16_t *depth, const uint8_t *rgb, const double *frame_mats,
                                   int32_t n_frames, const int32_t *pairs, int64_t n_pairs, int32_t dh,
                                   int32_t dw, int32_t H, int32_t W, uint64_t *out_vis_bits,
                                   uint8_t *out_vis_u8, uint8_t *out_valid_u8, int16_t *out_pix_i16,
                                   float *out_xyz_f32, uint32_t *out_rgba, double *out_xyz_f64,
                                   double *out_uv_f64, double *out_depth_f64, int32_t *out_counts,
                                   uint32_t flags, mspa_stream_t stream) {
    if (!depth || !frame_mats || !pairs) return fail(MSPA_EINVAL, "mspa_pair_reproject: null input pointer");
    if (n_frames <= 0 || n_pairs < 0) return fail(MSPA_EINVAL, "mspa_pair_reproject: bad frame/pair count");
    if (dh < 2 || dw < 2 || H < 2 || W < 2 || dh > 32767 || dw > 32767 || H > 32767 || W > 32767)
        return fail(MSPA_EINVAL, "mspa_pair_reproject: image size out of range [2, 32767]");
    const uint64_t P = (uint64_t)H * (uint64_t)W;
    if (P * (uint64_t)W >= (1ull << 32)) return fail(MSPA_EINVAL, "mspa_pair_reproject: H*W*W must be < 2^32");
    if (out_rgba && !rgb) return fail(MSPA_EINVAL, "mspa_pair_reproject: out_rgba needs rgb");
    if (flags & ~MSPA_PAIR_FAST) return fail(MSPA_EINVAL, "mspa_pair_reproject: unknown flag");
    if (flags & MSPA_PAIR_FAST) return fail(MSPA_EUNSUPPORTED, "mspa_pair_reproject: MSPA_PAIR_FAST not built yet");
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

    if (out_counts) {
        int rc = check_hip(hipMemsetAsync(out_counts, 0, sizeof(int32_t) * 2 * n_pairs, s), "hipMemsetAsync(counts)");
        if (rc) return rc;
    }
    const int64_t groups = (n_pairs + 7) / 8;
    const int64_t blocks = groups * 8 * a.strips;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_pair_reproject: too many workgroups; split the batch");
    const bool ident = (dh == H && dw == W);
    if (ident)
        hipLaunchKernelGGL(pair_reproject_kernel<true>, dim3((uint32_t)blocks), dim3(kThreads), 0, s, depth, rgb, frame_mats, pairs, a);
    else
        hipLaunchKernelGGL(pair_reproject_kernel<false>, dim3((uint32_t)blocks), dim3(kThreads), 0, s, depth, rgb, frame_mats, pairs, a);
    return check_hip(hipGetLastError(), "pair_reproject_kernel launch");
}
