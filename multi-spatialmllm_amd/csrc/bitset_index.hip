// K9: visibility bitsets -> index lists, on the device.
//
// MVI.process_scene (make_visibility_info.py:103-118) turns every image's visibility mask into
// `np.where(mask)[0].tolist()` and, per vertex, collects the sorted ids of the images that see it.  With K1's bitsets
// ([n_images, ceil(N/64)] uint64) both are compactions of a bit matrix -- the second one of its transpose -- and
// neither needs a Python list: the results are CSR tables (row offsets + int32 indices) that go to arrow / parquet as
// columns (mspa/visindex.py).
//
//   mspa_bits_popcount   per-word popcounts (their exclusive prefix sum -- one cumsum of the host layer -- positions
//                        every set bit in the output)
//   mspa_bits_expand     a wave walks 64 words; for each word lane L owns bit L and writes its index at
//                        base(word) + popcount(word & lanes below L): consecutive positions, coalesced stores
//   mspa_bits_transpose  64 x 64 bit tiles through ballots: lane r holds row r's word, ballot(bit b of my word) IS row b of
//                        the transposed tile (lane b keeps it)
#include "mspa_common.h"
#include "inflate_fast.h"
#include "host_pool.h"

#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <zlib.h>

namespace mspa {

constexpr int kBThreads = 256;

__global__ __launch_bounds__(kBThreads) void bits_popcount_kernel(const uint64_t *__restrict__ bits, int64_t n, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBThreads + threadIdx.x;
    if (i < n) out[i] = __popcll(bits[i]);
}

// one wave per 64 consecutive words of the flat [n_rows * n_words] table
__global__ __launch_bounds__(kBThreads) void bits_expand_kernel(const uint64_t *__restrict__ bits, int64_t n_total, int64_t n_words,
                                                                const int64_t *__restrict__ word_offsets,
                                                                int32_t *__restrict__ out) {
    const int64_t w0 = ((int64_t)blockIdx.x * (kBThreads / kWave) + (threadIdx.x >> 6)) * kWave;
    if (w0 >= n_total) return;
    const int lane = threadIdx.x & 63;
    const int64_t wi = w0 + lane;
    const bool live = wi < n_total;
    const uint64_t mine = live ? bits[wi] : 0ull;
    const int64_t base = live ? word_offsets[wi] : 0;
    const uint32_t col = live ? (uint32_t)(wi % n_words) : 0u;                  // word index within its row
    const uint32_t mlo = (uint32_t)mine, mhi = (uint32_t)(mine >> 32);
    const uint32_t blo = (uint32_t)base, bhi = (uint32_t)((uint64_t)base >> 32);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;        // lanes (= bits) below this one
    const int n_here = (int)min((int64_t)kWave, n_total - w0);
    for (int k = 0; k < n_here; ++k) {                                            // wave-uniform trip count
        const unsigned long long word = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mlo, k) |
                                        ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)mhi, k) << 32);
        if (word == 0) continue;                                                  // uniform
        const int64_t b = (int64_t)((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)blo, k) |
                                    ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)bhi, k) << 32));
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)col, k);
        if ((word >> lane) & 1ull) out[b + __popcll(word & below)] = (int32_t)(c * 64u + (uint32_t)lane);
    }
}

// out[(64 wq + b), rq] bit r  =  in[(64 rq + r), wq] bit b
__global__ __launch_bounds__(kBThreads) void bits_transpose_kernel(const uint64_t *__restrict__ in, int n_rows, int64_t n_words,
                                                                   int64_t row_blocks, uint64_t *__restrict__ out) {
    const int64_t tile = (int64_t)blockIdx.x * (kBThreads / kWave) + (threadIdx.x >> 6);
    if (tile >= n_words * row_blocks) return;
    const int64_t wq = tile / row_blocks, rq = tile - wq * row_blocks;
    const int lane = threadIdx.x & 63;
    const int64_t r = rq * 64 + lane;
    const uint64_t mine = r < n_rows ? in[r * n_words + wq] : 0ull;
    uint64_t mineT = 0;
#pragma unroll 8
    for (int b = 0; b < 64; ++b) {
        const unsigned long long t = __builtin_amdgcn_ballot_w64(((mine >> b) & 1ull) != 0);   // row b of the transposed tile
        mineT = (lane == b) ? t : mineT;
    }
    out[(wq * 64 + lane) * row_blocks + rq] = mineT;
}

}  // namespace mspa

using namespace mspa;

extern "C" int mspa_bits_popcount(const uint64_t *bits, int64_t n_words_total, int32_t *out_counts, mspa_stream_t stream) {
    if (n_words_total < 0) return fail(MSPA_EINVAL, "mspa_bits_popcount: bad size");
    if (n_words_total == 0) return MSPA_OK;
    if (!bits || !out_counts) return fail(MSPA_EINVAL, "mspa_bits_popcount: null pointer");
    const int64_t blocks = (n_words_total + kBThreads - 1) / kBThreads;
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_bits_popcount: table too large; split it");
    hipLaunchKernelGGL(bits_popcount_kernel, dim3((uint32_t)blocks), dim3(kBThreads), 0, (hipStream_t)stream, bits, n_words_total,
                       out_counts);
    return check_hip(hipGetLastError(), "bits_popcount_kernel launch");
}

extern "C" int mspa_bits_expand(const uint64_t *bits, int64_t n_rows, int64_t n_words, const int64_t *word_offsets,
                                int32_t *out_indices, mspa_stream_t stream) {
    if (n_rows < 0 || n_words < 0) return fail(MSPA_EINVAL, "mspa_bits_expand: bad size");
    if (n_rows == 0 || n_words == 0) return MSPA_OK;
    if (!bits || !word_offsets || !out_indices) return fail(MSPA_EINVAL, "mspa_bits_expand: null pointer");
    if (n_words > (1LL << 25)) return fail(MSPA_EINVAL, "mspa_bits_expand: rows longer than 2^31 bits");
    const int64_t total = n_rows * n_words;
    const int64_t waves = (total + kWave - 1) / kWave;
    const int64_t blocks = (waves + (kBThreads / kWave) - 1) / (kBThreads / kWave);
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_bits_expand: table too large; split it");
    hipLaunchKernelGGL(bits_expand_kernel, dim3((uint32_t)blocks), dim3(kBThreads), 0, (hipStream_t)stream, bits, total, n_words,
                       word_offsets, out_indices);
    return check_hip(hipGetLastError(), "bits_expand_kernel launch");
}

extern "C" int mspa_bits_transpose(const uint64_t *bits, int32_t n_rows, int64_t n_words, uint64_t *out, mspa_stream_t stream) {
    if (n_rows < 0 || n_words < 0) return fail(MSPA_EINVAL, "mspa_bits_transpose: bad size");
    if (n_rows == 0 || n_words == 0) return MSPA_OK;
    if (!bits || !out) return fail(MSPA_EINVAL, "mspa_bits_transpose: null pointer");
    const int64_t row_blocks = ((int64_t)n_rows + 63) / 64;
    const int64_t tiles = n_words * row_blocks;
    const int64_t blocks = (tiles + (kBThreads / kWave) - 1) / (kBThreads / kWave);
    if (blocks > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_bits_transpose: table too large; split it");
    hipLaunchKernelGGL(bits_transpose_kernel, dim3((uint32_t)blocks), dim3(kBThreads), 0, (hipStream_t)stream, bits, (int)n_rows,
                       n_words, row_blocks, out);
    return check_hip(hipGetLastError(), "bits_transpose_kernel launch");
}

// ---------------------------------------------------------------------------------------------------------
// Host-side text formatting of the index columns (HOST pointers; names end in _host).  The visibility parquet stores every
// list as the JSON text json.dumps gives it (make_visibility_info.py:38-73): 1.2 M numbers and 131 k keys per 64-frame
// scene.  Formatting them with Python objects or generic string kernels costs 0.15-0.5 s per scene -- three orders of
// magnitude more than the kernels that produce the lists -- so the text is written here, straight into arrow's
// (offsets, data) layout.
// ---------------------------------------------------------------------------------------------------------
namespace {

inline char *put_uint(char *p, uint64_t v) {
    char tmp[20];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

inline char *put_int(char *p, int64_t v) {
    if (v < 0) {
        *p++ = '-';
        return put_uint(p, (uint64_t)(-(v + 1)) + 1u);
    }
    return put_uint(p, (uint64_t)v);
}

}  // namespace

extern "C" int64_t mspa_format_int_lists_host(const int64_t *offsets_host, const int32_t *values_host, int64_t n_lists,
                                              char *out_text_host, int64_t capacity, int32_t *out_text_offsets_host) {
    if (n_lists < 0 || capacity < 0 || (n_lists > 0 && (!offsets_host || !out_text_host || !out_text_offsets_host)))
        return fail(MSPA_EINVAL, "mspa_format_int_lists_host: bad argument");
    char *p = out_text_host, *const end = out_text_host + capacity;
    if (out_text_offsets_host) out_text_offsets_host[0] = 0;
    for (int64_t k = 0; k < n_lists; ++k) {
        const int64_t a = offsets_host[k], b = offsets_host[k + 1];
        if (b < a || (b > a && !values_host)) return fail(MSPA_EINVAL, "mspa_format_int_lists_host: offsets not ascending");
        if (end - p < 2 + (b - a) * 13) return fail(MSPA_EINVAL, "mspa_format_int_lists_host: output buffer too small");
        *p++ = '[';
        for (int64_t e = a; e < b; ++e) {
            if (e > a) { *p++ = ','; *p++ = ' '; }
            p = put_int(p, values_host[e]);
        }
        *p++ = ']';
        if (p - out_text_host > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_format_int_lists_host: more than 2 GiB of text");
        out_text_offsets_host[k + 1] = (int32_t)(p - out_text_host);
    }
    return (int64_t)(p - out_text_host);
}

extern "C" int64_t mspa_format_token_lists_host(const int64_t *offsets_host, const int32_t *token_ids_host, int64_t n_lists,
                                                const char *tokens_host, const int32_t *token_offsets_host, int32_t n_tokens,
                                                char *out_text_host, int64_t capacity, int32_t *out_text_offsets_host) {
    if (n_lists < 0 || capacity < 0 || n_tokens < 0 ||
        (n_lists > 0 && (!offsets_host || !out_text_host || !out_text_offsets_host)) || (n_tokens > 0 && (!tokens_host || !token_offsets_host)))
        return fail(MSPA_EINVAL, "mspa_format_token_lists_host: bad argument");
    int32_t longest = 0;
    for (int32_t t = 0; t < n_tokens; ++t) {
        const int32_t len = token_offsets_host[t + 1] - token_offsets_host[t];
        if (len < 0) return fail(MSPA_EINVAL, "mspa_format_token_lists_host: token offsets not ascending");
        longest = len > longest ? len : longest;
    }
    char *p = out_text_host, *const end = out_text_host + capacity;
    if (out_text_offsets_host) out_text_offsets_host[0] = 0;
    for (int64_t k = 0; k < n_lists; ++k) {
        const int64_t a = offsets_host[k], b = offsets_host[k + 1];
        if (b < a || (b > a && !token_ids_host)) return fail(MSPA_EINVAL, "mspa_format_token_lists_host: offsets not ascending");
        if (end - p < 2 + (b - a) * (int64_t)(longest + 2)) return fail(MSPA_EINVAL, "mspa_format_token_lists_host: output buffer too small");
        *p++ = '[';
        for (int64_t e = a; e < b; ++e) {
            const int32_t t = token_ids_host[e];
            if (t < 0 || t >= n_tokens) return fail(MSPA_EINVAL, "mspa_format_token_lists_host: token id out of range");
            if (e > a) { *p++ = ','; *p++ = ' '; }
            const int32_t len = token_offsets_host[t + 1] - token_offsets_host[t];
            memcpy(p, tokens_host + token_offsets_host[t], (size_t)len);
            p += len;
        }
        *p++ = ']';
        if (p - out_text_host > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_format_token_lists_host: more than 2 GiB of text");
        out_text_offsets_host[k + 1] = (int32_t)(p - out_text_host);
    }
    return (int64_t)(p - out_text_host);
}

extern "C" int64_t mspa_format_int_keys_host(const char *prefix_host, int64_t first, int64_t n, char *out_text_host,
                                             int64_t capacity, int32_t *out_text_offsets_host) {
    if (n < 0 || capacity < 0 || !prefix_host || (n > 0 && (!out_text_host || !out_text_offsets_host)))
        return fail(MSPA_EINVAL, "mspa_format_int_keys_host: bad argument");
    const size_t plen = strlen(prefix_host);
    char *p = out_text_host, *const end = out_text_host + capacity;
    if (out_text_offsets_host) out_text_offsets_host[0] = 0;
    for (int64_t k = 0; k < n; ++k) {
        if (end - p < (int64_t)plen + 21) return fail(MSPA_EINVAL, "mspa_format_int_keys_host: output buffer too small");
        memcpy(p, prefix_host, plen);
        p = put_int(p + plen, first + k);
        if (p - out_text_host > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_format_int_keys_host: more than 2 GiB of text");
        out_text_offsets_host[k + 1] = (int32_t)(p - out_text_host);
    }
    return (int64_t)(p - out_text_host);
}

// Host-side staging: n equally sized blocks (a scene's depth frames as the image reader left them, one allocation each)
// gathered into one contiguous destination -- the pinned buffer the H2D copy reads -- by up to n_threads copy threads.
// A single thread writes pinned memory at ~20 GB/s, below what the PCIe link takes; the caller (mspa/upload.py) holds no
// interpreter lock while this runs.
extern "C" int mspa_gather_blocks_host(const void *const *src_blocks_host, int64_t n_blocks, int64_t block_bytes,
                                       void *dst_host, int32_t n_threads) {
    if (n_blocks < 0 || block_bytes < 0 || (n_blocks > 0 && block_bytes > 0 && (!src_blocks_host || !dst_host)))
        return fail(MSPA_EINVAL, "mspa_gather_blocks_host: bad argument");
    for (int64_t k = 0; k < n_blocks; ++k)
        if (block_bytes > 0 && !src_blocks_host[k]) return fail(MSPA_EINVAL, "mspa_gather_blocks_host: null block");
    if (n_blocks == 0 || block_bytes == 0) return MSPA_OK;
    const int64_t nt = n_threads < 1 ? 1 : (n_threads > n_blocks ? n_blocks : (int64_t)n_threads);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_blocks) return;
            memcpy((char *)dst_host + k * block_bytes, src_blocks_host[k], (size_t)block_bytes);
        }
    };
    HostPool::get().parallel((int)nt, work);                // persistent worker threads (host_pool.h); the caller copies too
    return MSPA_OK;
}

// Host-side ingest of zlib-compressed frames (the depth payloads of a ScanNet .sens stream, SENS:49-57 `zlib_ushort`): block k is
// inflated straight into dst + k * block_bytes by up to n_threads threads (blocks are handed out one at a time: payload
// sizes differ).  A block that does not inflate to exactly block_bytes fails the call and names the block.
extern "C" int mspa_inflate_blocks_host(const void *const *src_blocks_host, const int64_t *src_bytes_host, int64_t n_blocks,
                                        int64_t block_bytes, void *dst_host, int32_t n_threads) {
    if (n_blocks < 0 || block_bytes <= 0 || (n_blocks > 0 && (!src_blocks_host || !src_bytes_host || !dst_host)))
        return fail(MSPA_EINVAL, "mspa_inflate_blocks_host: bad argument");
    if (block_bytes > 0x7fffffffLL) return fail(MSPA_EINVAL, "mspa_inflate_blocks_host: block larger than 2 GiB");
    for (int64_t k = 0; k < n_blocks; ++k)
        if (!src_blocks_host[k] || src_bytes_host[k] <= 0 || src_bytes_host[k] > 0x7fffffffLL)
            return fail(MSPA_EINVAL, "mspa_inflate_blocks_host: null or empty block");
    if (n_blocks == 0) return MSPA_OK;
    const int64_t nt = n_threads < 1 ? 1 : (n_threads > n_blocks ? n_blocks : (int64_t)n_threads);
    std::atomic<int64_t> next{0}, bad{-1};
    auto work = [&]() {
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_blocks) return;
            // first through the table-driven decoder (inflate_fast.h: exact size + Adler-32, or it declines), then zlib itself
            if (inflate_zlib((const uint8_t *)src_blocks_host[k], (size_t)src_bytes_host[k], (uint8_t *)dst_host + k * block_bytes,
                             (size_t)block_bytes))
                continue;
            uLongf got = (uLongf)block_bytes;
            const int rc = uncompress((Bytef *)dst_host + k * block_bytes, &got, (const Bytef *)src_blocks_host[k],
                                      (uLong)src_bytes_host[k]);
            if (rc != Z_OK || (int64_t)got != block_bytes) {
                int64_t none = -1;
                bad.compare_exchange_strong(none, k);
            }
        }
    };
    HostPool::get().parallel((int)nt, work);
    if (bad.load() >= 0) {
        char msg[128];
        snprintf(msg, sizeof msg, "mspa_inflate_blocks_host: block %lld is not a zlib stream of the expected size", (long long)bad.load());
        return fail(MSPA_EINVAL, msg);
    }
    return MSPA_OK;
}
