// K10: the JSON text of the visibility index's lists, written ON the MI355X.
//
// What it replaces: `json.dumps(points)` / `json.dumps(images)` per key of make_visibility_info.py:38-73 (the parquet stores each list as
// its JSON text), i.e. 81 MB of text per 320-frame scene.  Round 5 wrote that text with a sequential host loop straight into arrow's
// string buffers (mspa_format_int_lists_host / mspa_format_token_lists_host: ~40 ms of one core per scene, half of what the index
// sweep's encoder threads spend per scene); here every list ITEM is formatted by its own lane and lands at its final byte position.
//
// Two launches around two prefix sums (the caller's plumbing, as for K9's popcount -> prefix sum -> expand):
//   mspa_format_list_costs_device   cost[e] = bytes of item e's text + 2 (its ", " or its share of the brackets); a token id out of
//                                   range raises a flag
//   mspa_format_lists_device        item e of list r = offsets[r] + k goes to
//                                       row_start[r] + 1 + (T[e] - T[offsets[r]]),   row_start[r] = 2 r + T[offsets[r]] - 2 NE[r]
//                                   (T: exclusive prefix sum of cost, nnz + 1 entries; NE[r]: non-empty lists in front of r, n + 1
//                                   entries); the list's first item also writes '[', its last one ']', the others ", "; a second
//                                   small launch writes "[]" for empty lists and arrow's int32 string offsets.
// Bytes are written where they belong, 5 - 10 per lane and contiguous over the wave (a wave's items are neighbours in the text):
// HBM-bound byte work -- 4 B read per item + its ~8 B of text written; nothing is reshaped.
// Bit-exact with the host formatters and json.dumps (tests/test_gpu_format_lists.py).
#include "mspa_common.h"

namespace mspa {
namespace fmt {

__device__ __forceinline__ int digits_u32(uint32_t v) {
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7
         : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}

__global__ __launch_bounds__(256) void list_costs_kernel(const int32_t *__restrict__ values, int64_t nnz,
                                                         const int32_t *__restrict__ token_offsets, int32_t n_tokens,
                                                         int32_t *__restrict__ cost, int32_t *__restrict__ bad) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    const int32_t v = values[e];
    int32_t c;
    if (token_offsets != nullptr) {
        if (v < 0 || v >= n_tokens) {
            atomicOr(bad, 1);
            c = 0;
        } else {
            c = token_offsets[v + 1] - token_offsets[v];
        }
    } else {
        const uint32_t mag = v < 0 ? (uint32_t)0 - (uint32_t)v : (uint32_t)v;
        c = digits_u32(mag) + (v < 0 ? 1 : 0);
    }
    cost[e] = c + 2;
}

__global__ __launch_bounds__(256) void lists_write_kernel(const int64_t *__restrict__ offsets, const int32_t *__restrict__ values,
                                                          int64_t n_lists, int64_t nnz, const int64_t *__restrict__ T,
                                                          const int64_t *__restrict__ NE, const char *__restrict__ tokens,
                                                          const int32_t *__restrict__ token_offsets, char *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= nnz) return;
    // the list this item belongs to: the last r with offsets[r] <= e (empty lists in front of it share its offset: upper bound)
    int64_t lo = 0, hi = n_lists;                                // invariant: offsets[lo] <= e < offsets[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= e) lo = mid;
        else hi = mid;
    }
    const int64_t r = lo, o = offsets[r];
    const int64_t row_start = 2 * r + T[o] - 2 * NE[r];
    const int64_t te = T[e];
    char *p = out + row_start + 1 + (te - T[o]);
    const int32_t len = (int32_t)(T[e + 1] - te) - 2;
    if (e == o) out[row_start] = '[';
    const int32_t v = values[e];
    if (token_offsets != nullptr) {
        const char *t = tokens + token_offsets[v];
        for (int32_t i = 0; i < len; ++i) p[i] = t[i];
    } else {
        uint32_t mag = v < 0 ? (uint32_t)0 - (uint32_t)v : (uint32_t)v;
        for (int32_t i = len - 1; i >= (v < 0 ? 1 : 0); --i) {
            p[i] = (char)('0' + mag % 10u);
            mag /= 10u;
        }
        if (v < 0) p[0] = '-';
    }
    if (e + 1 == offsets[r + 1]) {
        p[len] = ']';
    } else {
        p[len] = ',';
        p[len + 1] = ' ';
    }
}

// arrow's string offsets (list r's text starts at row_start[r]; entry n_lists = the total) and the text of empty lists
__global__ __launch_bounds__(256) void lists_rows_kernel(const int64_t *__restrict__ offsets, int64_t n_lists, const int64_t *__restrict__ T,
                                                         const int64_t *__restrict__ NE, char *__restrict__ out,
                                                         int32_t *__restrict__ out_text_offsets) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r > n_lists) return;
    const int64_t o = offsets[r];
    const int64_t row_start = 2 * r + T[o] - 2 * NE[r];
    out_text_offsets[r] = (int32_t)row_start;
    if (r < n_lists && offsets[r + 1] == o) {
        out[row_start] = '[';
        out[row_start + 1] = ']';
    }
}

}  // namespace fmt
}  // namespace mspa

using namespace mspa;

extern "C" int mspa_format_list_costs_device(const int32_t *values_dev, int64_t nnz, const int32_t *token_offsets_dev, int32_t n_tokens,
                                             int32_t *out_cost_dev, int32_t *bad_flag_dev, void *stream) {
    if (nnz < 0 || n_tokens < 0) return fail(MSPA_EINVAL, "mspa_format_list_costs_device: bad size");
    if (nnz == 0) return MSPA_OK;
    if (!values_dev || !out_cost_dev || !bad_flag_dev) return fail(MSPA_EINVAL, "mspa_format_list_costs_device: null pointer");
    if (nnz > (int64_t)0x7fffffff * 256) return fail(MSPA_EINVAL, "mspa_format_list_costs_device: too many items");
    hipLaunchKernelGGL(fmt::list_costs_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, (hipStream_t)stream, values_dev, nnz,
                       token_offsets_dev, n_tokens, out_cost_dev, bad_flag_dev);
    return check_hip(hipGetLastError(), "mspa_format_list_costs_device");
}

extern "C" int mspa_format_lists_device(const int64_t *offsets_dev, const int32_t *values_dev, int64_t n_lists, int64_t nnz,
                                        const int64_t *cost_prefix_dev, const int64_t *nonempty_prefix_dev, const char *tokens_dev,
                                        const int32_t *token_offsets_dev, char *out_text_dev, int64_t text_bytes,
                                        int32_t *out_text_offsets_dev, void *stream) {
    if (n_lists < 0 || nnz < 0 || text_bytes < 0) return fail(MSPA_EINVAL, "mspa_format_lists_device: bad size");
    if (text_bytes > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_format_lists_device: more than 2 GiB of text");
    if (!offsets_dev || !cost_prefix_dev || !nonempty_prefix_dev || !out_text_offsets_dev || (text_bytes > 0 && !out_text_dev) ||
        (nnz > 0 && !values_dev) || ((tokens_dev == nullptr) != (token_offsets_dev == nullptr)))
        return fail(MSPA_EINVAL, "mspa_format_lists_device: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (nnz > 0)
        hipLaunchKernelGGL(fmt::lists_write_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, offsets_dev, values_dev, n_lists, nnz,
                           cost_prefix_dev, nonempty_prefix_dev, tokens_dev, token_offsets_dev, out_text_dev);
    hipLaunchKernelGGL(fmt::lists_rows_kernel, dim3((unsigned)((n_lists + 1 + 255) / 256)), dim3(256), 0, st, offsets_dev, n_lists,
                       cost_prefix_dev, nonempty_prefix_dev, out_text_dev, out_text_offsets_dev);
    return check_hip(hipGetLastError(), "mspa_format_lists_device");
}
