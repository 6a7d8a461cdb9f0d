// K8: per (object, image) axis-aligned extent of the object's vertices the image sees -- what the coverage
// search of object_perception reads off its boolean union masks (single_object_coverage_finder.py
// compute_coverage, COV:56-65: max(coords) - min(coords) over mask & object, per axis).  The extent of a
// union of images is the min / max of the per-image extents, so the search itself never needs the masks.
//
// The objects arrive as one concatenated vertex list (CSR).  A wave owns 64 consecutive entries of that list
// and 64 images (lane <-> image): per entry the vertex id and its coordinates are wave-uniform (scalar loads),
// every lane tests its image's bit in K1's bitset (a scene's bitsets sit in L2) and keeps min / max / count in
// registers -- no cross-lane reduction anywhere.  When the walk crosses an object boundary, and at the end,
// lanes that saw something fold their registers into the [object, image] slot with the native float64
// atomic min / max of gfx950.  ceil(M/64) x ceil(F/64) waves: ~2.7k for a ScanNet scene's furniture.
// Pure selection: the result is bit-exact whatever the order of the atomics.
#include "mspa_common.h"

namespace mspa {

constexpr int kEWaves = 4;                 // waves per block, 64 list entries each
constexpr int kESeg = kEWaves * kWave;

__global__ __launch_bounds__(256) void extents_init_kernel(double *__restrict__ lo, double *__restrict__ hi,
                                                           int32_t *__restrict__ count, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const double inf = __builtin_inf();
    lo[3 * k + 0] = inf; lo[3 * k + 1] = inf; lo[3 * k + 2] = inf;
    hi[3 * k + 0] = -inf; hi[3 * k + 1] = -inf; hi[3 * k + 2] = -inf;
    count[k] = 0;
}

__global__ __launch_bounds__(kESeg) void object_extents_kernel(
    const uint64_t *__restrict__ bits, int64_t n_words, int32_t F, const double *__restrict__ xyz,
    const int32_t *__restrict__ offsets, const int32_t *__restrict__ vertices, int32_t n_objects,
    double *__restrict__ out_lo, double *__restrict__ out_hi, int32_t *__restrict__ out_count) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f = blockIdx.y * kWave + lane;
    const bool live = f < F;
    const uint64_t *__restrict__ row = bits + (int64_t)(live ? f : F - 1) * n_words;
    const int total = offsets[n_objects];
    const int beg = (blockIdx.x * kEWaves + w) * kWave;
    if (beg >= total) return;
    const int end = beg + kWave < total ? beg + kWave : total;
    // object of the first entry: last o with offsets[o] <= beg
    int a = 0, b = n_objects;
    while (b - a > 1) {
        const int m = (a + b) >> 1;
        if (offsets[m] <= beg) a = m; else b = m;
    }
    int o = a;
    int o_end = offsets[o + 1];
    const double inf = __builtin_inf();
    double lx = inf, ly = inf, lz = inf, hx = -inf, hy = -inf, hz = -inf;
    int cnt = 0;
    auto flush = [&]() {
        if (cnt > 0 && live) {
            const int64_t r = (int64_t)o * F + f;
            unsafeAtomicMin(out_lo + 3 * r + 0, lx); unsafeAtomicMin(out_lo + 3 * r + 1, ly); unsafeAtomicMin(out_lo + 3 * r + 2, lz);
            unsafeAtomicMax(out_hi + 3 * r + 0, hx); unsafeAtomicMax(out_hi + 3 * r + 1, hy); unsafeAtomicMax(out_hi + 3 * r + 2, hz);
            atomicAdd(out_count + r, cnt);
        }
        lx = ly = lz = inf; hx = hy = hz = -inf; cnt = 0;
    };
    constexpr int kU = 8;
    for (int i = beg; i < end; i += kU) {
        int v[kU];
        uint64_t word[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            v[u] = (i + u < end) ? vertices[i + u] : -1;
            word[u] = v[u] >= 0 ? row[v[u] >> 6] : 0;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (v[u] < 0) break;
            if (i + u >= o_end) {                       // wave-uniform: the list moved on to the next object
                flush();
                do { ++o; o_end = offsets[o + 1]; } while (i + u >= o_end);
            }
            const bool seen = (word[u] >> (v[u] & 63)) & 1;
            const double x = xyz[3 * (int64_t)v[u] + 0], y = xyz[3 * (int64_t)v[u] + 1], z = xyz[3 * (int64_t)v[u] + 2];
            if (seen) {
                lx = x < lx ? x : lx; hx = x > hx ? x : hx;
                ly = y < ly ? y : ly; hy = y > hy ? y : hy;
                lz = z < lz ? z : lz; hz = z > hz ? z : hz;
                ++cnt;
            }
        }
    }
    flush();
}

}  // namespace mspa

using namespace mspa;

extern "C" int mspa_object_extents(const uint64_t *vis_bits, int32_t n_images, int64_t n_words, const double *xyz,
                                   int64_t n_vertices, const int32_t *obj_offsets, const int32_t *obj_vertices,
                                   int64_t n_list_entries, int32_t n_objects, double *out_lo, double *out_hi, int32_t *out_count,
                                   mspa_stream_t stream) {
    if (n_images < 0 || n_objects < 0 || n_words < 0 || n_vertices < 0)
        return fail(MSPA_EINVAL, "mspa_object_extents: negative size");
    if (n_images == 0 || n_objects == 0) return MSPA_OK;
    if (!vis_bits || !xyz || !obj_offsets || (!obj_vertices && n_list_entries > 0) || !out_lo || !out_hi || !out_count)
        return fail(MSPA_EINVAL, "mspa_object_extents: null pointer");
    if (n_words * 64 < n_vertices) return fail(MSPA_EINVAL, "mspa_object_extents: bitset rows shorter than the vertex count");
    const int64_t chunks = ((int64_t)n_images + kWave - 1) / kWave;
    if (chunks > 65535) return fail(MSPA_EINVAL, "mspa_object_extents: too many images");
    if (n_list_entries < 0) return fail(MSPA_EINVAL, "mspa_object_extents: negative size");
    const int64_t slots = (int64_t)n_objects * n_images;
    hipLaunchKernelGGL(extents_init_kernel, dim3((uint32_t)((slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       out_lo, out_hi, out_count, slots);
    if (n_list_entries == 0) return check_hip(hipGetLastError(), "extents_init_kernel launch");
    const int64_t segs = (n_list_entries + kESeg - 1) / kESeg;
    hipLaunchKernelGGL(object_extents_kernel, dim3((uint32_t)segs, (uint32_t)chunks), dim3(kESeg), 0, (hipStream_t)stream,
                       vis_bits, n_words, n_images, xyz, obj_offsets, obj_vertices, n_objects, out_lo, out_hi, out_count);
    return check_hip(hipGetLastError(), "object_extents_kernel launch");
}
