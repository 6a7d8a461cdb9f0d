// K4: per-pair camera relations (CFR:176-189 distance / yaw / pitch columns, CME:185-190 relative pose)
// K5: TAPVid-3D track geometry (OM_C:293-315 projection, OM_C:446-454 camera->world,
//     OM_C:324-356 / OM_C:484-498 displacements).  See include/mspa.h.
//
// Both are tiny, latency-bound element-wise kernels: one lane per pair / per track sample, float64
// throughout, same FMA-chain order as NumPy (4-term rows; np.linalg.norm of one vector = ddot with
// FMA, np.linalg.norm(axis=1) = plain sum of squares).  They exist to keep the pair tables and track
// records on the device next to K1-K3's outputs, not because they are hot.
#include "mspa_common.h"

namespace mspa {

__device__ __forceinline__ double norm3_dot(double x, double y, double z) {   // np.linalg.norm(v): sqrt(ddot)
    return __builtin_sqrt(__builtin_fma(z, z, __builtin_fma(y, y, x * x)));
}
__device__ __forceinline__ double norm3_sum(double x, double y, double z) {   // np.linalg.norm(a, axis=1)
    return __builtin_sqrt((x * x + y * y) + z * z);
}

__global__ __launch_bounds__(256) void pair_pose_kernel(const double *__restrict__ E, const double *__restrict__ Einv,
                                                        const double *__restrict__ yaw,
                                                        const double *__restrict__ pitch,
                                                        const int32_t *__restrict__ pairs, int64_t n_pairs,
                                                        double *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n_pairs) return;
    const int i = pairs[2 * p], j = pairs[2 * p + 1];
    const double *Ei = E + (int64_t)i * 16, *Ej = E + (int64_t)j * 16, *Ii = Einv + (int64_t)i * 16;
    double *o = out + p * 6;
    // CFR:183  ||t_j - t_i||
    o[0] = norm3_dot(Ej[3] - Ei[3], Ej[7] - Ei[7], Ej[11] - Ei[11]);
    o[1] = yaw[j] - yaw[i];        // CFR:181 (raw difference, wrapped only later at CME:168-172)
    o[2] = pitch[j] - pitch[i];    // CFR:182
    // CME:185-189  translation column of inv(E_i) @ E_j: full 4-term rows, E_j[3][3] == 1
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double acc = Ii[4 * r] * Ej[3];
        acc = __builtin_fma(Ii[4 * r + 1], Ej[7], acc);
        acc = __builtin_fma(Ii[4 * r + 2], Ej[11], acc);
        acc = __builtin_fma(Ii[4 * r + 3], Ej[15], acc);
        o[3 + r] = acc;
    }
}

// extract_yaw_pitch (CFR:86-100) per frame: z = E[:3, 2]; yaw = degrees(atan2(z_y, z_x)); pitch = degrees(asin(z_z / ||z||)).
// np.degrees(x) = x * (180 / pi) with the double constant; np.linalg.norm of a 3-vector = sqrt of the pairwise-summed squares.
// The device libm (ocml) is not glibc: results agree to a few ulp, not bit for bit -- these are float64 quantities (1e-5 bar);
// callers that need the reference's exact bits (the parquet writers) keep the NumPy path (engine.extract_yaw_pitch_host).
__global__ __launch_bounds__(256) void yaw_pitch_kernel(const double *__restrict__ E, int n, double *__restrict__ yaw,
                                                        double *__restrict__ pitch) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= n) return;
    const double *M = E + (int64_t)f * 16;
    const double zx = M[2], zy = M[6], zz = M[10];
    const double kDeg = 57.29577951308232;                       // 180 / pi as NumPy's degrees() multiplies
    yaw[f] = atan2(zy, zx) * kDeg;
    pitch[f] = asin(zz / norm3_sum(zx, zy, zz)) * kDeg;
}

__global__ __launch_bounds__(256) void track_world_kernel(const double *__restrict__ tracks, const double *__restrict__ c2w,
                                                          int T, int P, double fx, double fy, double cx, double cy,
                                                          double Wd, double Hd, double *__restrict__ world,
                                                          double *__restrict__ uvn, uint8_t *__restrict__ ok) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (int64_t)T * P) return;
    const int t = (int)(k / P);
    const double x = tracks[3 * k], y = tracks[3 * k + 1], z = tracks[3 * k + 2];
    const double *M = c2w + (int64_t)t * 16;
    if (world) {   // OM_C:446-454: einsum('nij,nkj->nki') over homogeneous points
#pragma unroll
        for (int r = 0; r < 3; ++r) world[3 * k + r] = affine_row(M + 4 * r, x, y, z);
    }
    // OM_C:293-315
    const double u = (fx * x / (z + 1e-8)) + cx;
    const double v = (fy * y / (z + 1e-8)) + cy;
    const double un = u / Wd, vn = v / Hd;
    if (uvn) {
        uvn[2 * k] = un;
        uvn[2 * k + 1] = vn;
    }
    if (ok) ok[k] = (0.0 <= un && un < 1.0 && 0.0 <= vn && vn < 1.0 && z > 0.0) ? 1 : 0;
}

__global__ __launch_bounds__(256) void track_disp_kernel(const double *__restrict__ world, const double *__restrict__ w2c,
                                                         const double *__restrict__ c2w, const int32_t *__restrict__ trip,
                                                         int64_t n, int P, double thr_obj, double thr_cam,
                                                         double *__restrict__ out, uint8_t *__restrict__ flags) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int f1 = trip[3 * k], f2 = trip[3 * k + 1], p = trip[3 * k + 2];
    const double *a = world + ((int64_t)f1 * P + p) * 3, *b = world + ((int64_t)f2 * P + p) * 3;
    double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    double *o = out + 5 * k;
    o[4] = norm3_sum(dx, dy, dz);                  // OM_C:491 (axis=1 form, used for the pair binning)
    double dist = norm3_dot(dx, dy, dz);           // OM_C:332
    const bool moving = !(dist < thr_obj);         // OM_C:334-339
    if (!moving) { dist = 0.0; dx = dy = dz = 0.0; }
    const double *C1 = c2w + (int64_t)f1 * 16, *C2 = c2w + (int64_t)f2 * 16;
    const double cd = norm3_dot(C2[3] - C1[3], C2[7] - C1[7], C2[11] - C1[11]);   // OM_C:346
    const double *Wm = w2c + (int64_t)f1 * 16;     // OM_C:354-356, homogeneous w = 0
    o[0] = dist;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double acc = Wm[4 * r] * dx;
        acc = __builtin_fma(Wm[4 * r + 1], dy, acc);
        acc = __builtin_fma(Wm[4 * r + 2], dz, acc);
        acc = __builtin_fma(Wm[4 * r + 3], 0.0, acc);
        o[1 + r] = acc;
    }
    flags[2 * k] = moving ? 1 : 0;
    flags[2 * k + 1] = (cd < thr_cam) ? 0 : 1;     // OM_C:347-350
}

// K5c: pair mining of the object-movement head (OM_C:484-498): for a selected track point and the frames it is
// visible in, the world-space distance between every two of those frames, in the reference's order (i < j,
// row-major).  np.linalg.norm(points2 - points1, axis=1): sqrt((dx*dx + dy*dy) + dz*dz).
// grid = (ceil(n_max / 16), ceil(n_max / 16), selected points), 16 x 16 threads = a tile of the (i, j) square.
__global__ __launch_bounds__(256) void track_pair_distances_kernel(const double *__restrict__ world, int P,
                                                                   const int32_t *__restrict__ points,
                                                                   const int32_t *__restrict__ frame_offsets,
                                                                   const int32_t *__restrict__ frames,
                                                                   const int64_t *__restrict__ out_offsets,
                                                                   double *__restrict__ out) {
    const int s = blockIdx.z;
    const int beg = frame_offsets[s];
    const int n = frame_offsets[s + 1] - beg;
    const int i = blockIdx.y * 16 + (threadIdx.x >> 4);
    const int j = blockIdx.x * 16 + (threadIdx.x & 15);
    if (i >= n || j >= n || j <= i) return;
    const int p = points[s];
    const double *a = world + ((int64_t)frames[beg + i] * P + p) * 3;
    const double *b = world + ((int64_t)frames[beg + j] * P + p) * 3;
    const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    const int64_t k = out_offsets[s] + (int64_t)i * n - (int64_t)i * (i + 1) / 2 + (j - i - 1);
    out[k] = __builtin_sqrt((dx * dx + dy * dy) + dz * dz);
}

// K7: accumulated pairwise-distance change of the tracks (rigid_body_segmentation, OM_C:49-92): for every
// point pair (i, j): sum over t >= 1 of |d_t - d_{t-1}| where it exceeds the smoothing threshold, d_t the
// Euclidean distance of the two points at frame t (scipy pdist: sqrt((dx*dx + dy*dy) + dz*dz)).  One lane
// per pair walks the frames in order, so the float64 sum has the reference's order.
__global__ __launch_bounds__(256) void rigidity_loss_kernel(const double *__restrict__ tracks, int T, int P,
                                                            double smoothing, double *__restrict__ loss) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (int64_t)P * P) return;
    const int i = (int)(k / P), j = (int)(k % P);
    double acc = 0.0;
    if (i != j) {
        double prev = 0.0;
        for (int t = 0; t < T; ++t) {
            const double *a = tracks + ((int64_t)t * P + i) * 3, *b = tracks + ((int64_t)t * P + j) * 3;
            const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
            const double d = __builtin_sqrt((dx * dx + dy * dy) + dz * dz);
            if (t > 0) {
                const double ch = __builtin_fabs(d - prev);
                acc += (ch > smoothing) ? ch : 0.0;        // OM_C:46-47
            }
            prev = d;
        }
    }
    loss[k] = acc;
}

}  // namespace mspa

using namespace mspa;

extern "C" int mspa_track_pair_distances(const double *world, int32_t T, int32_t P, const int32_t *points,
                                         const int32_t *frame_offsets, const int32_t *frames, int32_t n_selected,
                                         int32_t n_frames_max, const int64_t *out_offsets, double *out,
                                         mspa_stream_t stream) {
    if (T < 0 || P < 0 || n_selected < 0 || n_frames_max < 0) return fail(MSPA_EINVAL, "mspa_track_pair_distances: bad size");
    if (n_selected == 0 || n_frames_max < 2) return MSPA_OK;
    if (!world || !points || !frame_offsets || !frames || !out_offsets || !out)
        return fail(MSPA_EINVAL, "mspa_track_pair_distances: null pointer");
    if (n_selected > 65535) return fail(MSPA_EINVAL, "mspa_track_pair_distances: too many selected points for one launch");
    const uint32_t tiles = ((uint32_t)n_frames_max + 15u) / 16u;
    hipLaunchKernelGGL(track_pair_distances_kernel, dim3(tiles, tiles, (uint32_t)n_selected), dim3(256), 0,
                       (hipStream_t)stream, world, P, points, frame_offsets, frames, out_offsets, out);
    return check_hip(hipGetLastError(), "track_pair_distances_kernel launch");
}

extern "C" int mspa_track_rigidity_loss(const double *tracks_xyz, int32_t T, int32_t P, double smoothing_factor,
                                        double *out_loss, mspa_stream_t stream) {
    if (T < 0 || P < 0) return fail(MSPA_EINVAL, "mspa_track_rigidity_loss: bad size");
    if (P == 0) return MSPA_OK;
    if (!tracks_xyz || !out_loss) return fail(MSPA_EINVAL, "mspa_track_rigidity_loss: null pointer");
    const int64_t blocks = ((int64_t)P * P + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_track_rigidity_loss: too many points");
    hipLaunchKernelGGL(rigidity_loss_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, tracks_xyz, T, P,
                       smoothing_factor, out_loss);
    return check_hip(hipGetLastError(), "rigidity_loss_kernel launch");
}


extern "C" int mspa_pair_pose(const double *E_aligned, const double *Einv_aligned, const double *yaw,
                              const double *pitch, int32_t n_frames, const int32_t *pairs, int64_t n_pairs,
                              double *out, mspa_stream_t stream) {
    if (n_frames <= 0 || n_pairs < 0) return fail(MSPA_EINVAL, "mspa_pair_pose: bad size");
    if (n_pairs == 0) return MSPA_OK;
    if (!E_aligned || !Einv_aligned || !yaw || !pitch || !pairs || !out)
        return fail(MSPA_EINVAL, "mspa_pair_pose: null pointer");
    const int64_t blocks = (n_pairs + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_pair_pose: too many pairs; split the batch");
    hipLaunchKernelGGL(pair_pose_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, E_aligned,
                       Einv_aligned, yaw, pitch, pairs, n_pairs, out);
    return check_hip(hipGetLastError(), "pair_pose_kernel launch");
}

extern "C" int mspa_extract_yaw_pitch(const double *E_aligned, int32_t n_frames, double *out_yaw, double *out_pitch,
                                      mspa_stream_t stream) {
    if (n_frames < 0) return fail(MSPA_EINVAL, "mspa_extract_yaw_pitch: bad count");
    if (n_frames == 0) return MSPA_OK;
    if (!E_aligned || !out_yaw || !out_pitch) return fail(MSPA_EINVAL, "mspa_extract_yaw_pitch: null pointer");
    hipLaunchKernelGGL(yaw_pitch_kernel, dim3((uint32_t)((n_frames + 255) / 256)), dim3(256), 0, (hipStream_t)stream, E_aligned,
                       (int)n_frames, out_yaw, out_pitch);
    return check_hip(hipGetLastError(), "yaw_pitch_kernel launch");
}

extern "C" int mspa_track_to_world(const double *tracks_xyz, const double *c2w, int32_t T, int32_t P,
                                   const double *fx_fy_cx_cy_host, int32_t H, int32_t W, double *out_world,
                                   double *out_uvn, uint8_t *out_ok, mspa_stream_t stream) {
    if (T < 0 || P < 0 || H <= 0 || W <= 0) return fail(MSPA_EINVAL, "mspa_track_to_world: bad size");
    if ((int64_t)T * P == 0) return MSPA_OK;
    if (!tracks_xyz || !fx_fy_cx_cy_host || (out_world && !c2w))
        return fail(MSPA_EINVAL, "mspa_track_to_world: null pointer");
    const int64_t blocks = ((int64_t)T * P + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_track_to_world: too many samples; split the batch");
    hipLaunchKernelGGL(track_world_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, tracks_xyz,
                       c2w ? c2w : tracks_xyz, T, P, fx_fy_cx_cy_host[0], fx_fy_cx_cy_host[1], fx_fy_cx_cy_host[2],
                       fx_fy_cx_cy_host[3], (double)W, (double)H, out_world, out_uvn, out_ok);
    return check_hip(hipGetLastError(), "track_world_kernel launch");
}

extern "C" int mspa_track_displacement(const double *world, const double *w2c, const double *c2w, int32_t T,
                                       int32_t P, const int32_t *triples, int64_t n, double obj_threshold,
                                       double cam_threshold, double *out, uint8_t *out_flags,
                                       mspa_stream_t stream) {
    if (T <= 0 || P <= 0 || n < 0) return fail(MSPA_EINVAL, "mspa_track_displacement: bad size");
    if (n == 0) return MSPA_OK;
    if (!world || !w2c || !c2w || !triples || !out || !out_flags)
        return fail(MSPA_EINVAL, "mspa_track_displacement: null pointer");
    const int64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_track_displacement: too many samples; split the batch");
    hipLaunchKernelGGL(track_disp_kernel, dim3((uint32_t)blocks), dim3(256), 0, (hipStream_t)stream, world, w2c, c2w,
                       triples, n, P, obj_threshold, cam_threshold, out, out_flags);
    return check_hip(hipGetLastError(), "track_disp_kernel launch");
}
