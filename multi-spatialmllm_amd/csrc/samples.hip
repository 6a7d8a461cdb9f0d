// K6: the per-sample primitives of the ScanNet task heads (see include/mspa.h).
//   select_common_kernel  -- the j-th vertex visible in BOTH frames of a pair: element j of
//                            np.intersect1d(points1, points2) (VC_C:303) read straight off K1's bitsets
//                            (with frame1 == frame2 it is element j of one image's visible list, DE_C:190)
//   project_samples_kernel-- get_point_2d_coordinates_in_image (IH:291-305) for a batch of
//                            (vertex, image) samples: projection + visibility in one pass
// Both are one-wave-per-sample / one-lane-per-sample latency-bound kernels on L2-resident data.
#include "mspa_common.h"

namespace mspa {

__global__ __launch_bounds__(256) void select_common_kernel(const uint64_t *__restrict__ bits, int64_t n_words,
                                                            const int32_t *__restrict__ sel, int64_t n,
                                                            int32_t *__restrict__ out) {
    const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n) return;
    const int lane = threadIdx.x & 63;
    const uint64_t *__restrict__ a = bits + (int64_t)sel[3 * s + 0] * n_words;
    const uint64_t *__restrict__ b = bits + (int64_t)sel[3 * s + 1] * n_words;
    const int64_t want = sel[3 * s + 2];
    // each lane owns a contiguous chunk of words; exclusive prefix of the chunk popcounts locates the chunk
    const int64_t chunk = (n_words + 63) / 64;
    const int64_t w0 = lane * chunk, w1 = min(w0 + chunk, n_words);
    int cnt = 0;
    for (int64_t w = w0; w < w1; ++w) cnt += __popcll(a[w] & b[w]);
    int incl = cnt;
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    const int excl = incl - cnt;
    int result = -1;                                     // -1: fewer than want+1 common vertices
    if (want >= excl && want < incl) {
        int64_t rem = want - excl;
        for (int64_t w = w0; w < w1; ++w) {
            uint64_t x = a[w] & b[w];
            const int c = __popcll(x);
            if (rem < c) {
                for (int64_t k = 0; k < rem; ++k) x &= x - 1;   // drop the rem lowest set bits
                result = (int)(w * 64 + __builtin_ctzll(x));
                break;
            }
            rem -= c;
        }
    }
    const unsigned long long owner = __ballot(result >= 0);
    if (owner == 0) {
        if (lane == 0) out[s] = -1;
    } else if (result >= 0) {
        out[s] = result;
    }
}

__global__ __launch_bounds__(256) void project_samples_kernel(const double *__restrict__ xyz, int64_t point_stride,
                                                              int64_t comp_stride, const double *__restrict__ cam_mats,
                                                              const uint16_t *__restrict__ depth, int dh, int dw, int H,
                                                              int W, double sx, double sy, double dscale,
                                                              const int32_t *__restrict__ samples, int64_t n,
                                                              double *__restrict__ uv, double *__restrict__ pdepth,
                                                              uint8_t *__restrict__ vis) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int64_t v = samples[2 * s + 0];
    const int img = samples[2 * s + 1];
    const double x = xyz[v * point_stride], y = xyz[v * point_stride + comp_stride],
                 z = xyz[v * point_stride + 2 * comp_stride];
    const double *Einv = cam_mats + (int64_t)img * (MSPA_CAM_MATS * 16), *K = Einv + 16;
    const double qx = affine_row(Einv + 0, x, y, z);      // IH:57-69, per-lane matrices (samples differ in image)
    const double qy = affine_row(Einv + 4, x, y, z);
    const double qz = affine_row(Einv + 8, x, y, z);
    const double ix = affine_row(K + 0, qx, qy, qz);
    const double iy = affine_row(K + 4, qx, qy, qz);
    const double iz = affine_row(K + 8, qx, qy, qz);
    const double u = ix / iz, w = iy / iz;
    int xi, yi;
    const bool ok = depth_test(true, u, w, qz, depth + (int64_t)img * dh * dw, dh, dw, H, W, sx, sy, xi, yi, nullptr, dscale);
    uv[2 * s] = u;
    uv[2 * s + 1] = w;
    pdepth[s] = qz;
    vis[s] = ok ? 1 : 0;
}

}  // namespace mspa

using namespace mspa;

extern "C" int mspa_select_common_point(const uint64_t *bits, int32_t n_images, int64_t n_words,
                                        const int32_t *selections, int64_t n, int32_t *out_vertex,
                                        mspa_stream_t stream) {
    if (n_images <= 0 || n_words <= 0 || n < 0) return fail(MSPA_EINVAL, "mspa_select_common_point: bad size");
    if (n == 0) return MSPA_OK;
    if (!bits || !selections || !out_vertex) return fail(MSPA_EINVAL, "mspa_select_common_point: null pointer");
    if (n_words > (1LL << 25)) return fail(MSPA_EINVAL, "mspa_select_common_point: bitset too long");
    const int64_t blocks = (n + 3) / 4;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_select_common_point: too many samples; split the batch");
    hipLaunchKernelGGL(select_common_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, bits, n_words,
                       selections, n, out_vertex);
    return check_hip(hipGetLastError(), "select_common_kernel launch");
}

extern "C" int mspa_project_samples_ex(const double *xyz, int64_t n_points, int64_t point_stride, int64_t comp_stride,
                                       const double *cam_mats, int32_t n_images, const uint16_t *depth, int32_t dh,
                                       int32_t dw, int32_t H, int32_t W, double depth_value_scale, const int32_t *samples,
                                       int64_t n, double *out_uv, double *out_depth, uint8_t *out_visible,
                                       mspa_stream_t stream) {
    if (!(depth_value_scale > 0.0 && depth_value_scale < 1e300))
        return fail(MSPA_EINVAL, "mspa_project_samples: depth_value_scale must be positive and finite");
    if (n_points <= 0 || n_images <= 0 || n < 0 || point_stride <= 0 || comp_stride <= 0)
        return fail(MSPA_EINVAL, "mspa_project_samples: bad size");
    if (n == 0) return MSPA_OK;
    if (!xyz || !cam_mats || !depth || !samples || !out_uv || !out_depth || !out_visible)
        return fail(MSPA_EINVAL, "mspa_project_samples: null pointer");
    if (dh < 2 || dw < 2 || H < 2 || W < 2 || dh > 32767 || dw > 32767 || H > 32767 || W > 32767)
        return fail(MSPA_EINVAL, "mspa_project_samples: image size out of range [2, 32767]");
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_project_samples: too many samples; split the batch");
    hipLaunchKernelGGL(project_samples_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, xyz,
                       point_stride, comp_stride, cam_mats, depth, dh, dw, H, W, (double)dw / (double)W,
                       (double)dh / (double)H, depth_value_scale, samples, n, out_uv, out_depth, out_visible);
    return check_hip(hipGetLastError(), "project_samples_kernel launch");
}

extern "C" int mspa_project_samples(const double *xyz, int64_t n_points, int64_t point_stride, int64_t comp_stride,
                                    const double *cam_mats, int32_t n_images, const uint16_t *depth, int32_t dh,
                                    int32_t dw, int32_t H, int32_t W, const int32_t *samples, int64_t n,
                                    double *out_uv, double *out_depth, uint8_t *out_visible, mspa_stream_t stream) {
    return mspa_project_samples_ex(xyz, n_points, point_stride, comp_stride, cam_mats, n_images, depth, dh, dw, H, W, 0.001,
                                   samples, n, out_uv, out_depth, out_visible, stream);
}
