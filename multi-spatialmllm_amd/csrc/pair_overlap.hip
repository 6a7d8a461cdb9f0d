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

}  // namespace mspa

using namespace mspa;

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
