// K8: per (object, image) axis-aligned extent of the object's vertices the image sees -- what the coverage
// search of object_perception reads off its boolean union masks (single_object_coverage_finder.py
// compute_coverage, COV:56-65: max(coords) - min(coords) over mask & object, per axis).  The extent of a
// union of images is the min / max of the per-image extents, so the search itself never needs the masks.
//
// Block = (object, chunk of 64 images); lane <-> image, the four waves of the block split the object's
// vertex list.  Per vertex the index and its coordinates are wave-uniform (scalar loads), every lane tests
// its image's bit in K1's bitset (the bitsets of a scene sit in L2) and keeps min / max / count in
// registers; the four waves meet in LDS.  Pure selection: the result is bit-exact for any input.
#include "mspa_common.h"

namespace mspa {

struct ExtentArgs {
    const uint64_t *bits;      // [F, n_words]
    int64_t n_words;
    int32_t F;
    const double *xyz;         // [V, 3]
    const int32_t *offsets;    // [O + 1]
    const int32_t *vertices;   // [offsets[O]]
    double *lo;                // [O, F, 3]
    double *hi;                // [O, F, 3]
    int32_t *count;            // [O, F]
};

constexpr int kEWaves = 4;

__global__ __launch_bounds__(kEWaves *kWave) void object_extents_kernel(ExtentArgs a) {
    const int o = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f = blockIdx.y * kWave + lane;
    const int fc = f < a.F ? f : a.F - 1;
    const uint64_t *__restrict__ row = a.bits + (int64_t)fc * a.n_words;
    const int beg = a.offsets[o], end = a.offsets[o + 1];
    const double inf = __builtin_inf();
    double lx = inf, ly = inf, lz = inf, hx = -inf, hy = -inf, hz = -inf;
    int cnt = 0;
    constexpr int kU = 4;
    for (int i = beg + w * kU; i < end; i += kEWaves * kU) {
        int v[kU];
        uint64_t word[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            v[u] = (i + u < end) ? a.vertices[i + u] : -1;
            word[u] = v[u] >= 0 ? row[v[u] >> 6] : 0;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (v[u] < 0) continue;
            const bool seen = (word[u] >> (v[u] & 63)) & 1;
            const double x = a.xyz[3 * (int64_t)v[u] + 0], y = a.xyz[3 * (int64_t)v[u] + 1], z = a.xyz[3 * (int64_t)v[u] + 2];
            if (seen) {
                lx = x < lx ? x : lx; hx = x > hx ? x : hx;
                ly = y < ly ? y : ly; hy = y > hy ? y : hy;
                lz = z < lz ? z : lz; hz = z > hz ? z : hz;
                ++cnt;
            }
        }
    }
    __shared__ double s_ext[kEWaves][6][kWave];
    __shared__ int s_cnt[kEWaves][kWave];
    s_ext[w][0][lane] = lx; s_ext[w][1][lane] = ly; s_ext[w][2][lane] = lz;
    s_ext[w][3][lane] = hx; s_ext[w][4][lane] = hy; s_ext[w][5][lane] = hz;
    s_cnt[w][lane] = cnt;
    __syncthreads();
    if (w == 0 && f < a.F) {
        for (int k = 1; k < kEWaves; ++k) {
            double t;
            t = s_ext[k][0][lane]; lx = t < lx ? t : lx;
            t = s_ext[k][1][lane]; ly = t < ly ? t : ly;
            t = s_ext[k][2][lane]; lz = t < lz ? t : lz;
            t = s_ext[k][3][lane]; hx = t > hx ? t : hx;
            t = s_ext[k][4][lane]; hy = t > hy ? t : hy;
            t = s_ext[k][5][lane]; hz = t > hz ? t : hz;
            cnt += s_cnt[k][lane];
        }
        const int64_t r = (int64_t)o * a.F + f;
        a.lo[3 * r + 0] = lx; a.lo[3 * r + 1] = ly; a.lo[3 * r + 2] = lz;
        a.hi[3 * r + 0] = hx; a.hi[3 * r + 1] = hy; a.hi[3 * r + 2] = hz;
        a.count[r] = cnt;
    }
}

}  // namespace mspa

using namespace mspa;

extern "C" int mspa_object_extents(const uint64_t *vis_bits, int32_t n_images, int64_t n_words, const double *xyz,
                                   int64_t n_vertices, const int32_t *obj_offsets, const int32_t *obj_vertices,
                                   int32_t n_objects, double *out_lo, double *out_hi, int32_t *out_count,
                                   mspa_stream_t stream) {
    if (n_images < 0 || n_objects < 0 || n_words < 0 || n_vertices < 0)
        return fail(MSPA_EINVAL, "mspa_object_extents: negative size");
    if (n_images == 0 || n_objects == 0) return MSPA_OK;
    if (!vis_bits || !xyz || !obj_offsets || !obj_vertices || !out_lo || !out_hi || !out_count)
        return fail(MSPA_EINVAL, "mspa_object_extents: null pointer");
    if (n_words * 64 < n_vertices) return fail(MSPA_EINVAL, "mspa_object_extents: bitset rows shorter than the vertex count");
    const int64_t chunks = ((int64_t)n_images + kWave - 1) / kWave;
    if (chunks > 65535) return fail(MSPA_EINVAL, "mspa_object_extents: too many images");
    ExtentArgs a{vis_bits, n_words, n_images, xyz, obj_offsets, obj_vertices, out_lo, out_hi, out_count};
    hipLaunchKernelGGL(object_extents_kernel, dim3((uint32_t)n_objects, (uint32_t)chunks), dim3(kEWaves * kWave), 0,
                       (hipStream_t)stream, a);
    return check_hip(hipGetLastError(), "object_extents_kernel launch");
}
