// Depth-frame decode ON the MI355X: DEFLATE (RFC 1951) / zlib (RFC 1950) streams inflated by one wave each, Adler-32 checked,
// PNG scanline filters undone -- compressed frames cross PCIe (half the bytes) and land as the [F, h, w] uint16 block K1 / K3 read.
//
// What it replaces: the per-frame `cv2.imread(depth_png, -1)` of SceneInfoHandler.get_depth_image (info_handler.py:149-155) and
// the per-frame `zlib.decompress` of the .sens reader (extract_posed_images.py:49-57), i.e. the stage that bounds every from-disk
// sweep: 16 CPUs' worth of container quota inflate ~10 k frames/s (profiles/r06_ingest_scaling.txt) while the kernels downstream
// take 3 M images/s.  DEFLATE is serial inside a stream, so the parallelism is ACROSS streams: a scene has 320 of them, the
// loader keeps several scenes in flight, the chip has room for 4 096 such waves (16 per CU: 104 registers and 9 KB of LDS each):
// 86 k frames/s (75 k with 3 584).
//
// mspa::dinf::inflate_kernel -- one 64-lane workgroup (one wave) per stream (v6):
//   * the POSITION in the stream is wave-uniform and lives in two SGPRs; the stream's bytes sit in two VGPRs as a sliding window of
//     128 words (lane l: word base + l, and word base + 64 + l on its way from HBM), re-aligned every ~28 batches with two
//     `ds_bpermute_b32`.  Consuming bits is an addition: there is no bit buffer to shift or refill.
//   * symbols in batches of 64 bit offsets: EVERY LANE DECODES THE WHOLE SYMBOL that would start at the bit offset equal to its
//     lane number -- three `ds_bpermute_b32` cut its windows out of the window register, an LDS gather reads the 11-bit
//     literal/length table (16-bit entries), a length's extra bits come from the window, a second gather reads the 8-bit distance
//     table behind them -- and packs total bits, copy routine, match length and distance into one word.  The serial part DEFLATE
//     forces is following the chain from offset 0: one `v_readlane_b32`, a compare, a bit set and an add per literal.  A run's
//     literals are stored by their own lanes in one instruction (exec = the run's lane mask, rank = `v_mbcnt` of exec).
//   * the whole batch loop -- that decode, the chain, the bounds checks, the two hot copy routines (one `ds_read_u8` /
//     `ds_write_b8` pair ring to ring; `global_load_ubyte` from the flushed bytes for what lies beyond the ring), the batch's
//     bookkeeping -- is ONE hand-written statement (block_symbols): the frame's time is its instruction count (8.6 M per 640 x 480
//     frame; the CU issues ~1.1 per cycle with 14 such waves, profiles/r06_inflate_v6_pmc.md), and compiled from C the same loop took
//     three times the scalar instructions.  C handles what is rare: the general copy (overlapping or longer than 64 bytes), a
//     flush, a window re-alignment, a code longer than the table index, the block headers.
//   * tables: built per dynamic block by all 64 lanes (ballot ranks -> canonical codes -> replicated entries).  Codes longer than
//     the table index (< 1 % of the symbols of a noisy depth frame with 11 bits; 5 % with 10), end of block and invalid patterns
//     take a scalar one-symbol path: the lanes test one code length each against the canonical first codes, the chain then goes
//     on in the same batch.
//   * output: a 2 KB ring in LDS takes every byte; whole 256-byte lines leave for HBM as one coalesced dword store per lane.
//     A match whose distance fits the ring (<= 1 790: every filter-row distance of a 640-pixel image, 1 281) is copied LDS to
//     LDS by the lanes; a farther one loads the flushed bytes from HBM behind `s_waitcnt vmcnt(8)` (the lines it needs left >= 12
//     stores ago) into a0 -- an accumulation register only the hand-written statements touch, because the load is still in
//     flight while compiled code runs -- and its ring write is deferred until the next match or flush needs it.
//   * a stream is ACCEPTED only if it ends exactly at the expected size, stays inside its input, and (second kernel,
//     adler32_kernel) the Adler-32 of the output equals the stream's trailer -- the contract of csrc/inflate_fast.h.  Anything
//     else is reported per block and the caller decodes that frame on the host.
// mspa::dinf::png_unfilter_kernel -- one wave per image.  Images whose rows use only None / Sub / Up (what adaptive writers choose
//   for depth maps) go row by row with the lanes along the row: coalesced dword loads, bytewise SWAR adds, Sub as a wave-wide
//   prefix sum.  An image with an Average or Paeth row (smooth depth: practically every row is Paeth) takes the skewed pipeline:
//   64 rows at a time, lane y works on pixel pair i - y at step i, so the left, upper and upper-left neighbours are a register,
//   the lane above's previous output via DPP, and the one before that; the row's raw bytes arrive through prefetched aligned
//   dwords.  Both write host-order uint16 pixels.
#include "mspa_common.h"

namespace mspa {
namespace dinf {

constexpr int kLitBits = 11, kDistBits = 8;
#ifndef MSPA_INFLATE_RING
// 2 KB, not 4: a wave's LDS is 9 KB instead of 11, so that the compute unit holds the 16 waves its registers allow instead of 14
// (4 096 streams: 85.8 k frames/s against 75.6 k at 3 584; per wave the two sizes decode at the same speed: tools/ab_ring.sh,
// profiles/r06_sweep_timeline.md).  Distances of 1 791 .. 3 838 take the HBM path with it; the row above (1 281) stays in the ring.
#define MSPA_INFLATE_RING 2048
#endif
constexpr int kRing = MSPA_INFLATE_RING, kRingMask = kRing - 1;
static_assert(kRing >= 2048 && (kRing & kRingMask) == 0, "the ring is a power of two that holds a row-to-row distance of a 640-pixel image");
constexpr int kRingNear = kRing - 258;          // a match at most this far back never reads a ring slot it is overwriting
// A far match (beyond kRingNear, at most 64 bytes) reads bytes that left the ring as whole 256-byte lines.  When it is looked at,
// fewer than 256 + 64 bytes are unflushed (the check behind every match and every batch, then one batch's literals), so the last
// line it reads was stored at least kFarLinesAgo line stores ago; stores and loads of one wave complete in order, so with at most
// kFarLinesAgo - 1 memory operations outstanding that store has landed.
constexpr int kFarLinesAgo = (kRingNear - 64 - 320) / 256 - 1;
constexpr int kFarVmcnt = kFarLinesAgo - 1 < 8 ? kFarLinesAgo - 1 : 8;
static_assert(kFarVmcnt >= 1, "ring too small for the deferred far-match load");
constexpr int kLitSyms = 288, kDistSyms = 32;
constexpr int32_t kPendingHard = 0x40000000;   // status of an image between png_unfilter_kernel and png_unfilter_hard_kernel

// entry: bits 0..7 code length, 8..10 kind, 11..15 extra-bit count, 16..31 payload
//   kind 0 literal (payload = byte) / code-length symbol; 1 length or distance (payload = base); 2 end of block;
//   3 a code longer than the table index (canonical walk); 4 invalid
__device__ __forceinline__ uint32_t pack(uint32_t len, uint32_t kind, uint32_t extra, uint32_t payload) {
    return len | (kind << 8) | (extra << 11) | (payload << 16);
}
constexpr uint32_t kInvalid = 1u | (4u << 8);
// The literal/length table holds 2 048 entries (11-bit index: the noisy depth frames' Huffman trees give ~5 % of the symbols
// an 11-bit code and < 1 % a longer one; every code beyond the index costs a trip through the scalar one-symbol path and a
// batch of its own) in 16 bits each, so that it still fits 4 KB: bits 0..3 code length, 4..7 kind (0 literal, 1..6 a length with
// kind - 1 extra bits, 7 end of block, 8 longer than the index, 9 invalid), 8..15 the literal's byte / the length's base - 3.
__device__ __forceinline__ uint32_t narrow(uint32_t e) {
    const uint32_t kind = (e >> 8) & 7u, extra = (e >> 11) & 31u, payload = e >> 16;
    const uint32_t k4 = kind == 0u ? 0u : kind == 1u ? 1u + extra : kind + 5u;          // 2, 3, 4 -> 7, 8, 9
    const uint32_t pay = kind == 1u ? payload - 3u : payload & 0xFFu;
    return (e & 15u) | (k4 << 4) | (pay << 8);
}
__device__ __forceinline__ uint32_t widen(uint32_t n) {
    const uint32_t len = n & 15u, k4 = (n >> 4) & 15u, pay = (n >> 8) & 0xFFu;
    if (k4 == 0u) return pack(len, 0, 0, pay);
    if (k4 < 7u) return pack(len, 1, k4 - 1u, pay + 3u);
    return pack(len, k4 - 5u, 0, 0);
}

__constant__ uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct __align__(16) WaveLds {
    uint16_t lit[1 << kLitBits];        // 4 096 B (narrow entries)
    uint32_t dist[1 << kDistBits];      // 1 024 B
    uint8_t ring[kRing];                // 2 048 B (MSPA_INFLATE_RING)
    uint32_t pre[128];                  //   512 B  code-length code, 7-bit index
    uint16_t sorted[kLitSyms + kDistSyms];   // 640 B  symbols by (code length, symbol), literal/length then distance alphabet
    uint8_t lens[kLitSyms + kDistSyms];      // 320 B
    // per alphabet (0 literal/length, 1 distance, 2 code-length code) and code length: first canonical code, symbol count,
    // offset of the length's first symbol in `sorted`
    uint32_t first[3][16], count[3][16], offs[3][16];
};

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The decode state.  The POSITION in the stream is wave-uniform and lives in SGPRs (`base`, `rb`); the stream's bytes sit in two
// VGPRs as a sliding window of 128 words: lane l holds word base + l (W) and word base + 64 + l (Wn, on its way from HBM since
// the previous re-alignment).  Consuming bits is `rb += k` -- there is no bit buffer to shift or refill.  The vector side cuts
// every lane's 32-bit window out of W with three `ds_bpermute_b32` (no LDS memory involved); the scalar side (block headers,
// code-length codes, the rare one-symbol path) reads two words with `v_readlane_b32`.  Once `rb` has moved 56 words into the
// window (every ~28 batches) the window is re-aligned: W = the words from position `rb >> 5` on (two bpermutes over W / Wn and a
// select), the next Wn is requested -- its latency hides behind the batches that follow.
struct Reader {
    const uint32_t *src;         // 8-byte aligned start of the stream
    uint32_t n_words;            // 32-bit words the stream's bytes span (rounded up)
    uint32_t n_words_ok;         // words that may be loaded (inside the caller's buffer)
    uint32_t W, Wn;              // per lane: word base + lane, word base + 64 + lane of the stream (zeros beyond its end)
    uint32_t base;               // word index of W's lane 0 (wave-uniform)
    uint32_t rb;                 // the next bit of the stream, relative to word `base` (wave-uniform; < 64 words)
    int lane_;

    static constexpr uint32_t kRealignAt = 56u * 32u;     // leaves 8 words: more than a batch (111 bits) + the windows behind it

    __device__ __forceinline__ uint32_t load_word(uint32_t i) const {
        uint32_t v = 0u;
        if (i < n_words_ok) v = src[i];
        return v;
    }
    __device__ __forceinline__ void seek(uint32_t byte_off, int lane) {
        base = byte_off >> 2;
        rb = (byte_off & 3u) * 8u;
        W = load_word(base + (uint32_t)lane);
        Wn = load_word(base + 64u + (uint32_t)lane);
    }
    __device__ __forceinline__ void realign() {
        const uint32_t s = rb >> 5;                              // words used up (wave-uniform, < 64)
        const uint32_t i = (uint32_t)lane_ + s;
        const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((i & 63u) << 2), (int)W);
        const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((i & 63u) << 2), (int)Wn);
        W = i < 64u ? a : b;
        base += s;
        rb &= 31u;
        Wn = load_word(base + 64u + (uint32_t)lane_);
    }
    // afterwards the next 8 words (256 bits) are in W: a batch of the symbol walk, or any header field sequence between two calls
    __device__ __forceinline__ void refill() {
        if (rb >= kRealignAt) realign();
    }
    __device__ __forceinline__ uint32_t peek(int k) const {  // 0 <= k <= 16 (any k < 32 works)
        const int j = (int)(rb >> 5);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)W, j);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)W, j + 1);
        const uint64_t v = ((uint64_t)hi << 32) | lo;
        return (uint32_t)(v >> (rb & 31u)) & ((1u << k) - 1u);
    }
    __device__ __forceinline__ void drop(int k) { rb += (uint32_t)k; }
    __device__ __forceinline__ uint32_t take(int k) {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    __device__ __forceinline__ void to_byte_boundary() { rb = (rb + 7u) & ~7u; }
    __device__ __forceinline__ uint32_t word_index() const { return base + (rb >> 5); }
    // bytes of the stream consumed so far (a partly consumed byte counts)
    __device__ __forceinline__ int64_t consumed_bytes() const { return ((int64_t)base * 32 + (int64_t)rb + 7) >> 3; }
};

__device__ __forceinline__ uint32_t brev(uint32_t v, int len) { return __builtin_bitreverse32(v) >> (32 - len); }

// Canonical Huffman tables of one alphabet from its code lengths, by all 64 lanes.  lens[0 .. n_syms) in LDS; `table` has
// 1 << bits entries; `sorted` receives the symbols ordered by (length, symbol).  `kind`: 0 literal/length alphabet, 1 distance,
// 2 code-length code.  Returns false (wave-uniform) for an over-subscribed set.
__device__ __attribute__((noinline)) bool build_table(WaveLds &L, int alpha, const uint8_t *lens, int n_syms, void *table_, int bits, uint16_t *sorted,
                            int lane) {
    // 1. symbols per code length (ballots over chunks of 64 symbols)
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (int base = 0; base < n_syms; base += 64) {
        const int s = base + lane;
        const int len = s < n_syms ? (int)lens[s] : 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) cnt[l] += (uint32_t)__builtin_popcountll(__ballot(len == l));
    }
    // 2. first canonical code and offset into `sorted` per length; Kraft check
    uint32_t first[16], offs[16];
    uint32_t code = 0, off = 0;
    int left = 1;
    bool ok = true;
    first[0] = offs[0] = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        left = (left << 1) - (int)cnt[l];
        ok = ok && left >= 0;
        code = (code + cnt[l - 1]) << 1;
        first[l] = code;
        offs[l] = off;
        off += cnt[l];
    }
    if (!ok) return false;
    if (lane < 16) {
        uint32_t f = 0, c = 0, o = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l)
            if (lane == l) { f = first[l]; c = cnt[l]; o = offs[l]; }
        L.first[alpha][lane] = f;
        L.count[alpha][lane] = c;
        L.offs[alpha][lane] = o;
    }
    uint32_t *const table = (uint32_t *)table_;                  // wide entries (distance and code-length alphabets)
    uint16_t *const table16 = (uint16_t *)table_;                // narrow entries (the literal/length alphabet)
    const bool is_narrow = alpha == 0;
    for (int i = lane; i < (1 << bits); i += 64) {
        if (is_narrow) table16[i] = (uint16_t)narrow(kInvalid);
        else table[i] = kInvalid;
    }
    wave_lds_fence();
    // 3. every symbol: its rank among the symbols of its length (in symbol order) -> canonical code -> entries
    uint32_t seen[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) seen[l] = 0;
    for (int base = 0; base < n_syms; base += 64) {
        const int s = base + lane;
        const int len = s < n_syms ? (int)lens[s] : 0;
        uint32_t rank = 0, fcode = 0, so = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) {
            const uint64_t b = __ballot(len == l);
            if (len == l) {
                rank = seen[l] + (uint32_t)__builtin_popcountll(b & ((1ull << lane) - 1ull));
                fcode = first[l];
                so = offs[l];
            }
            seen[l] += (uint32_t)__builtin_popcountll(b);
        }
        if (len > 0) {
            sorted[so + rank] = (uint16_t)s;
            const uint32_t rev = brev(fcode + rank, len);
            if (len <= bits) {
                uint32_t e;
                if (alpha == 0) {
                    if (s < 256) e = pack((uint32_t)len, 0, 0, (uint32_t)s);
                    else if (s == 256) e = pack((uint32_t)len, 2, 0, 0);
                    else if (s > 285) e = pack((uint32_t)len, 4, 0, 0);
                    else e = pack((uint32_t)len, 1, kLenExtra[s - 257], kLenBase[s - 257]);
                } else if (alpha == 1) {
                    e = s > 29 ? pack((uint32_t)len, 4, 0, 0) : pack((uint32_t)len, 1, kDistExtra[s], kDistBase[s]);
                } else {
                    e = pack((uint32_t)len, 0, 0, (uint32_t)s);
                }
                if (is_narrow) {
                    const uint16_t n = (uint16_t)narrow(e);
                    for (uint32_t i = rev; i < (1u << bits); i += 1u << len) table16[i] = n;
                } else {
                    for (uint32_t i = rev; i < (1u << bits); i += 1u << len) table[i] = e;
                }
            } else {
                const uint32_t e = pack((uint32_t)bits, 3, 0, 0);      // a longer code starts with these bits
                if (is_narrow) table16[rev & ((1u << bits) - 1u)] = (uint16_t)narrow(e);
                else table[rev & ((1u << bits) - 1u)] = e;
            }
        }
    }
    wave_lds_fence();
    return true;
}

// A code longer than the table index, by the lanes: lane l tests length l (first-code test on the bit-reversed prefix: one read
// of first / count / offs per lane instead of a scalar walk with three dependent LDS round trips per length), the shortest
// length that matches wins.  Returns symbol | length << 16, length 0 when no code matches (an incomplete set's unused pattern).
// Wave-uniform result; every lane must call it.
__device__ __forceinline__ uint32_t long_code(const WaveLds &L, int alpha, int bits, const uint16_t *sorted, uint32_t peek16, int lane) {
    const int l = lane & 15;
    const uint32_t code = brev(peek16 & ((1u << l) - 1u), l ? l : 1);
    const uint32_t f = L.first[alpha][l], c = L.count[alpha][l], o = L.offs[alpha][l];
    const bool hit = lane < 16 && l > bits && code - f < c;      // unsigned: code >= f and code < f + c
    const uint64_t m = __ballot(hit);
    if (m == 0) return 0;
    const int len = __builtin_ctzll(m);
    const uint32_t at = (uint32_t)__builtin_amdgcn_readlane((int)(o + (code - f)), len);
    return uni((uint32_t)sorted[at]) | ((uint32_t)len << 16);
}

// Whole 256-byte lines leave the ring: one coalesced dword store per lane and line.  Returns the new `flushed`.
__device__ __forceinline__ uint32_t flush_lines(WaveLds &L, uint8_t *dst, uint32_t pos, uint32_t flushed, int lane) {
    const uint32_t lines = (pos - flushed) >> 8;
    for (uint32_t n = 0; n < lines; ++n) {
        const uint32_t v = *(const uint32_t *)&L.ring[(flushed + 4u * (uint32_t)lane) & kRingMask];
        *(uint32_t *)(dst + flushed + 4u * (uint32_t)lane) = v;
        flushed += 256u;
    }
    return flushed;
}

// A match copied by the lanes: ring to ring when the source is within kRingNear, else (a long match far back) from the flushed
// bytes in HBM behind a workgroup-scope fence.  `pos` is the output position the match starts at.
__device__ __forceinline__ void copy_match(WaveLds &L, const uint8_t *dst, uint32_t pos, uint32_t length, uint32_t dist, int lane) {
    if (dist <= (uint32_t)kRingNear) {
        if (length <= 64u && dist >= length) {                   // the common case: one step, source and destination apart
            uint8_t b = 0;
            if ((uint32_t)lane < length) b = L.ring[(pos - dist + (uint32_t)lane) & kRingMask];
            wave_lds_fence();
            if ((uint32_t)lane < length) L.ring[(pos + (uint32_t)lane) & kRingMask] = b;
        } else if (dist >= 64u) {                                // 64 bytes per step; a step's sources were written by earlier steps
            for (uint32_t done = 0; done < length; done += 64) {
                const uint32_t i = done + (uint32_t)lane;
                uint8_t b = 0;
                if (i < length) b = L.ring[(pos - dist + i) & kRingMask];
                wave_lds_fence();
                if (i < length) L.ring[(pos + i) & kRingMask] = b;
                wave_lds_fence();
            }
        } else {                                                 // overlapping (run-like) copy: byte i repeats byte i mod dist
            const float inv = 1.0f / (float)dist;
            for (uint32_t done = 0; done < length; done += 64) {
                const uint32_t i = done + (uint32_t)lane;
                uint32_t q = (uint32_t)((float)i * inv);
                int rem = (int)i - (int)(q * dist);
                rem = rem < 0 ? rem + (int)dist : (rem >= (int)dist ? rem - (int)dist : rem);
                uint8_t b = 0;
                if (i < length) b = L.ring[(pos - dist + (uint32_t)rem) & kRingMask];
                wave_lds_fence();
                if (i < length) L.ring[(pos + i) & kRingMask] = b;
                wave_lds_fence();
            }
        }
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t done = 0; done < length; done += 64) {
            const uint32_t i = done + (uint32_t)lane;
            if (i < length) L.ring[(pos + i) & kRingMask] = dst[pos - dist + i];
        }
    }
    wave_lds_fence();
}

// The bytes of a far match that are still on their way from HBM go to the ring (before anything reads or flushes them).  They
// live in a0, an accumulation register only the hand-written statements here and in block_symbols touch: the load that fills it
// is issued by hand and is still in flight while C code runs, so it must not sit in a register the compiler may copy or reuse.
__device__ __forceinline__ void settle_far(WaveLds &L, uint32_t far_pos, uint32_t &far_len, int lane) {
    if (far_len) {
        uint32_t v;
        asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, a0" : "=v"(v) : : "memory", "a0");
        if ((uint32_t)lane < far_len) L.ring[(far_pos + (uint32_t)lane) & kRingMask] = (uint8_t)v;
        far_len = 0;
        wave_lds_fence();
    }
}

// The symbols of one block, tables in LDS.  true: the block's end-of-block symbol was reached; false: the stream is bad.
//
// In batches of 64 bit offsets.  Every lane DECODES THE WHOLE SYMBOL that would start at the bit offset equal to its lane
// number -- its 32-bit window of the stream (three `ds_bpermute_b32` over the window register W), the literal/length entry (an LDS
// gather), for a length its extra bits, the distance entry at the offset behind them (a second gather) and its extra bits -- and
// packs what the scalar side needs into one word P, INCLUDING the decisions that depend on the symbol alone (which copy routine
// a match takes).  All of that is vector work on 64 hypotheses at once.  The serial part, the thing DEFLATE forces, is following
// the chain from offset 0: for a literal one `v_readlane_b32`, a compare, a bit set and an add.  The literals of a run are then
// stored by their own lanes in ONE instruction.
//   P: a literal: its code length (1 .. 15), + 32 if the next symbol starts beyond lane 63 (the batch's last one);
//      16: not decodable here (a code longer than the table index, end of block, an invalid pattern): the scalar one-symbol path;
//      a match (>= 65 536): total bits (6) | routine (2: 0 one-step ring copy, 1 from HBM, 2 the general copy) << 6 |
//      (length - 3) << 8 | distance << 16.
// Zeros follow the stream's end: what decodes from them runs into the output bound or an invalid block header.
//
// THE WHOLE BATCH LOOP IS ONE HAND-WRITTEN STATEMENT.  The frame's time is its instruction count (profiles/r06_inflate_v6_pmc.md):
// as C the compiler carried the decode state through register copies at every merge and materialised each decision as a 64-bit
// lane mask -- ~90 scalar-side instructions per match, ~80 per batch; written out they are ~30 and ~12.  The statement leaves for
// what is rare, says why, and is re-entered where it left:
//   why 0  the symbol at `off` is for the scalar path (then enter 2)       why 1  bad stream
//   why 2  a match for the general copy routine, in p (then enter 2)       why 3  >= 256 bytes ready to leave the ring, mid-batch (enter 3)
//   why 5  the window register must be re-aligned (enter 0)                why 6  >= 256 bytes ready, at a batch's end (enter 0)
//   enter 0 a new batch, 1 on with the chain, 2 behind a symbol handled outside (lines ready? batch used up?), 3 behind a flush
// exec is all ones on entry (one wave, uniform control flow) and on every exit.  v90 .. v101 are its scratch registers -- on purpose ABOVE what the compiler needs: the kernel then counts 102 registers, four waves per SIMD, and a compute unit's 16 waves sit 4 + 4 + 4 + 4.  With v64 .. v75 it counts 87, five waves fit a SIMD, the dispatcher packs 5 + 5 + 5 + 1 and 4 096 streams take 52.0 ms instead of 45.8 (a SIMD issues for four of these waves without slowing them, not for five).  Hazards (gfx9
// rules): no VALU-written SGPR is used as a lane select or by VMEM; a0 (the far match's bytes, see settle_far) is waited for with
// vmcnt(0) before it is stored.
__device__ __forceinline__ bool block_symbols(WaveLds &L, Reader &r, uint8_t *__restrict__ dst, uint32_t out_n, int lane, uint32_t &pos,
                                              uint32_t &flushed, uint32_t &far_pos, uint32_t &far_len) {
    uint32_t E = 0, P = 0, off = 0, enter = 0;
    for (;;) {
        uint32_t p, why, len_s, dist_s, t0;
        uint64_t run, m0;
        if (enter == 0u) r.refill();                              // (the statement asks for it with why 5 as well)
#define MSPA_LIT_STEP                                   \
    "v_readlane_b32 %[p], %[P], %[off]\n\t"             \
    "s_cmp_lt_u32 %[p], 16\n\t"                         \
    "s_cbranch_scc0 2f\n\t"                             \
    "s_bitset1_b64 %[run], %[off]\n\t"                  \
    "s_add_u32 %[off], %[off], %[p]\n\t"
        asm volatile(
            "s_mov_b64 %[run], 0\n\t"
            "s_cmp_eq_u32 %[enter], 0\n\t"
            "s_cbranch_scc1 20f\n\t"
            "s_cmp_eq_u32 %[enter], 1\n\t"
            "s_cbranch_scc1 1f\n\t"
            "s_cmp_eq_u32 %[enter], 2\n\t"
            "s_cbranch_scc1 7f\n\t"
            "s_branch 8f\n"
            // ---- a new batch: every lane decodes the symbol at its bit offset --------------------------------------------------
            "20:\n\t"
            "s_mov_b32 %[why], 5\n\t"
            "s_cmp_ge_u32 %[rb], 0x700\n\t"                       // Reader::kRealignAt
            "s_cbranch_scc1 9f\n\t"
            "v_add_u32 v90, %[rb], %[lane]\n\t"                   // b: this lane's bit offset inside the window
            "v_lshrrev_b32 v91, 3, v90\n\t"
            "v_and_b32 v91, 0x1fc, v91\n\t"
            "ds_bpermute_b32 v92, v91, %[W]\n\t"                  // the three words its windows can touch
            "ds_bpermute_b32 v93, v91, %[W] offset:4\n\t"
            "ds_bpermute_b32 v94, v91, %[W] offset:8\n\t"
            "v_and_b32 v97, 31, v90\n\t"                          // sh
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbit_b32 v95, v93, v92, v90\n\t"               // win: stream bits [b, b + 32)
            "v_and_b32 v91, 0x7ff, v95\n\t"
            "v_lshlrev_b32 v91, 1, v91\n\t"
            "ds_read_u16 v96, v91 offset:%[lit]\n\t"              // the narrow literal/length entry
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_and_b32 v98, 15, v96\n\t"                          // clen
            "v_bfe_u32 v99, v96, 4, 4\n\t"                        // kind
            "v_lshrrev_b32 %[E], 8, v96\n\t"                      // a literal's byte / a length's base - 3
            "v_add_u32 v100, -1, v99\n\t"
            "v_cmp_gt_u32 vcc, 6, v100\n\t"                       // a length (kind 1 .. 6)
            "v_cndmask_b32 v100, 0, v100, vcc\n\t"                // lx: its extra bits
            "s_mov_b64 %[m0], vcc\n\t"
            "v_lshrrev_b32 v101, v98, v95\n\t"
            "v_bfe_u32 v101, v101, 0, v100\n\t"
            "v_add_u32 v101, %[E], v101\n\t"                      // length - 3
            "v_add3_u32 v97, v97, v98, v100\n\t"                  // o2: where the distance code starts (<= 31 + 20)
            "v_cmp_gt_u32 vcc, 32, v97\n\t"
            "v_cndmask_b32 v92, v93, v92, vcc\n\t"
            "v_cndmask_b32 v93, v94, v93, vcc\n\t"
            "v_alignbit_b32 v92, v93, v92, v97\n\t"               // win2
            "v_and_b32 v93, 0xff, v92\n\t"
            "v_lshlrev_b32 v93, 2, v93\n\t"
            "ds_read_b32 v93, v93 offset:%[dtab]\n\t"             // the distance entry
            "v_add_u32 v94, %[lane], v98\n\t"                     // a literal's word meanwhile
            "v_cmp_lt_u32 vcc, 63, v94\n\t"
            "v_cndmask_b32_e64 v94, 0, 32, vcc\n\t"
            "v_or_b32 v94, v94, v98\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_bfe_u32 v95, v93, 11, 5\n\t"                       // dx
            "v_and_b32 v96, 0xff, v93\n\t"                        // dlen
            "v_lshrrev_b32 v92, v96, v92\n\t"
            "v_bfe_u32 v92, v92, 0, v95\n\t"
            "v_lshrrev_b32 v97, 16, v93\n\t"
            "v_add_u32 v92, v97, v92\n\t"                         // distance
            "v_add3_u32 v96, v96, v95, v98\n\t"
            "v_add_u32 v96, v96, v100\n\t"                        // total bits
            "v_bfe_u32 v95, v93, 8, 3\n\t"
            "v_cmp_eq_u32 vcc, 1, v95\n\t"                        // a distance entry
            "s_and_b64 %[m0], %[m0], vcc\n\t"                     // a whole match
            "v_add_u32 v95, 3, v101\n\t"                          // length
            "v_cmp_lt_u32 vcc, v92, v95\n\t"                      // overlapping: the general copy
            "v_cndmask_b32_e64 v97, 0, 1, vcc\n\t"
            "v_lshlrev_b32 v97, 7, v97\n\t"
            "v_cmp_lt_u32 vcc, %[near], v92\n\t"                  // beyond the ring: from HBM
            "v_cndmask_b32_e64 v97, v97, 64, vcc\n\t"
            "v_mov_b32 v95, 0x80\n\t"
            "v_cmp_lt_u32 vcc, 61, v101\n\t"                      // longer than one step: the general copy
            "v_cndmask_b32 v97, v97, v95, vcc\n\t"
            "v_lshl_or_b32 v96, v101, 8, v96\n\t"
            "v_lshl_or_b32 v96, v92, 16, v96\n\t"
            "v_or_b32 v96, v96, v97\n\t"
            "v_cndmask_b32_e64 %[P], 16, v96, %[m0]\n\t"
            "v_cmp_eq_u32 vcc, 0, v99\n\t"
            "v_cndmask_b32 %[P], %[P], v94, vcc\n\t"
            "s_mov_b32 %[off], 0\n"
            // ---- the chain ---------------------------------------------------------------------------------------------------
            "1:\n\t" MSPA_LIT_STEP MSPA_LIT_STEP MSPA_LIT_STEP MSPA_LIT_STEP MSPA_LIT_STEP MSPA_LIT_STEP MSPA_LIT_STEP MSPA_LIT_STEP
            "s_branch 1b\n"
            "2:\n\t"                                              // p >= 16
            "s_cmp_gt_u32 %[p], 47\n\t"
            "s_cbranch_scc1 3f\n\t"                               // a match
            "s_cmp_eq_u32 %[p], 16\n\t"
            "s_cbranch_scc1 3f\n\t"
            "s_bitset1_b64 %[run], %[off]\n\t"                    // the batch's last literal
            "s_and_b32 %[t0], %[p], 15\n\t"
            "s_add_u32 %[off], %[off], %[t0]\n"
            "3:\n\t"
            "s_cmp_eq_u64 %[run], 0\n\t"
            "s_cbranch_scc1 4f\n\t"
            "s_mov_b64 exec, %[run]\n\t"                          // the run's literals, each by its own lane
            "v_mbcnt_lo_u32_b32 v90, exec_lo, 0\n\t"
            "v_mbcnt_hi_u32_b32 v90, exec_hi, v90\n\t"
            "v_add_u32 v90, %[pos], v90\n\t"
            "v_and_b32 v90, %[rmask], v90\n\t"
            "ds_write_b8 v90, %[E] offset:%[ring]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_bcnt1_i32_b64 %[t0], %[run]\n\t"
            "s_add_u32 %[pos], %[pos], %[t0]\n\t"
            "s_mov_b64 %[run], 0\n"
            "4:\n\t"
            "s_cmp_gt_u32 %[p], 0xffff\n\t"
            "s_cbranch_scc1 10f\n\t"                              // a match
            "s_mov_b32 %[why], 0\n\t"
            "s_cmp_eq_u32 %[p], 16\n\t"
            "s_cbranch_scc1 9f\n\t"                               // the scalar path's symbol: why 0
            "s_branch 11f\n"                                      // the last literal: the batch is used up
            "10:\n\t"
            "s_bfe_u32 %[len], %[p], 0x80008\n\t"
            "s_add_u32 %[len], %[len], 3\n\t"
            "s_lshr_b32 %[dist], %[p], 16\n\t"
            "s_mov_b32 %[why], 1\n\t"
            "s_cmp_gt_u32 %[dist], %[pos]\n\t"
            "s_cbranch_scc1 9f\n\t"                               // reaches in front of the output: bad
            "s_add_u32 %[t0], %[pos], %[len]\n\t"
            "s_cmp_gt_u32 %[t0], %[out_n]\n\t"
            "s_cbranch_scc1 9f\n\t"                               // runs past the expected size: bad
            "s_cmp_eq_u32 %[far_len], 0\n\t"
            "s_cbranch_scc1 5f\n\t"
            "v_cmp_gt_u32 vcc, %[far_len], %[lane]\n\t"           // the previous far match's bytes into the ring
            "s_mov_b64 exec, vcc\n\t"
            "v_add_u32 v90, %[far_pos], %[lane]\n\t"
            "v_and_b32 v90, %[rmask], v90\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "ds_write_b8 v90, a0 offset:%[ring]\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_mov_b32 %[far_len], 0\n"
            "5:\n\t"
            "s_mov_b32 %[why], 2\n\t"
            "s_bitcmp1_b32 %[p], 7\n\t"
            "s_cbranch_scc1 9f\n\t"                               // the general copy: why 2
            "v_cmp_gt_u32 vcc, %[len], %[lane]\n\t"
            "s_mov_b64 exec, vcc\n\t"                             // lanes < length
            "v_add_u32 v90, %[pos], %[lane]\n\t"
            "s_bitcmp1_b32 %[p], 6\n\t"
            "s_cbranch_scc1 6f\n\t"
            "v_subrev_u32 v91, %[dist], v90\n\t"                  // ring to ring, source and destination apart
            "v_and_b32 v91, %[rmask], v91\n\t"
            "ds_read_u8 v91, v91 offset:%[ring]\n\t"
            "v_and_b32 v90, %[rmask], v90\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_write_b8 v90, v91 offset:%[ring]\n\t"
            "s_branch 12f\n"
            "6:\n\t"
            "s_waitcnt vmcnt(%[farcnt])\n\t"                      // from HBM: the lines it reads left > kFarVmcnt stores ago
            "v_subrev_u32 v90, %[dist], v90\n\t"
            "global_load_ubyte a0, v90, %[dst]\n\t"
            "s_mov_b32 %[far_pos], %[pos]\n\t"
            "s_mov_b32 %[far_len], %[len]\n"
            "12:\n\t"
            "s_mov_b64 exec, -1\n\t"
            "s_add_u32 %[pos], %[pos], %[len]\n\t"
            "s_and_b32 %[t0], %[p], 63\n\t"
            "s_add_u32 %[off], %[off], %[t0]\n"
            "7:\n\t"                                              // behind a match (or a symbol handled outside)
            "s_mov_b32 %[why], 3\n\t"
            "s_sub_u32 %[t0], %[pos], %[flushed]\n\t"
            "s_cmp_ge_u32 %[t0], 0x100\n\t"
            "s_cbranch_scc1 9f\n"                                 // lines are ready to leave the ring: why 3
            "8:\n\t"
            "s_cmp_gt_u32 %[off], 63\n\t"
            "s_cbranch_scc0 1b\n"                                 // on with the chain
            "11:\n\t"                                             // ---- the batch is used up ----
            "s_add_u32 %[rb], %[rb], %[off]\n\t"
            "s_mov_b32 %[why], 1\n\t"
            "s_cmp_gt_u32 %[pos], %[out_n]\n\t"
            "s_cbranch_scc1 9f\n\t"                               // more output than expected: bad (before any of it leaves the ring)
            "s_mov_b32 %[why], 6\n\t"
            "s_sub_u32 %[t0], %[pos], %[flushed]\n\t"
            "s_cmp_ge_u32 %[t0], 0x100\n\t"
            "s_cbranch_scc0 20b\n"                                // the next batch
            "9:\n\t"
            : [rb] "+s"(r.rb), [off] "+s"(off), [pos] "+s"(pos), [far_pos] "+s"(far_pos), [far_len] "+s"(far_len),
              [p] "=&s"(p), [why] "=&s"(why), [len] "=&s"(len_s), [dist] "=&s"(dist_s), [t0] "=&s"(t0), [run] "=&s"(run), [m0] "=&s"(m0),
              [P] "+v"(P), [E] "+v"(E)          // (the vector operands last: behind one, the compiler takes scalar results for divergent)
            : [enter] "s"(enter), [W] "v"(r.W), [lane] "v"(lane), [flushed] "s"(flushed), [out_n] "s"(out_n), [dst] "s"(dst),
              [near] "s"((uint32_t)kRingNear), [ring] "n"(offsetof(WaveLds, ring)), [lit] "n"(offsetof(WaveLds, lit)),
              [dtab] "n"(offsetof(WaveLds, dist)), [rmask] "n"(kRingMask), [farcnt] "n"(kFarVmcnt)
            : "scc", "vcc", "memory", "a0", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101");
#undef MSPA_LIT_STEP
        // ---- what is rare: in C ----
        if (why == 5u) {                                         // (refill() at the top re-aligns)
            enter = 0;
            continue;
        }
        if (why == 3u || why == 6u) {
            settle_far(L, far_pos, far_len, lane);
            wave_lds_fence();
            flushed = uni(flush_lines(L, dst, pos, flushed, lane));
            enter = why == 3u ? 3u : 0u;
            continue;
        }
        if (why == 2u) {                                         // the general copy (overlapping or longer than a step)
            copy_match(L, dst, pos, len_s, dist_s, lane);
            pos += len_s;
            off += uni(p) & 63u;                                 // (uni: the compiler takes this one scalar result for divergent)
            enter = 2;
            continue;
        }
        if (why != 0u) return false;
        // ---- one symbol on the scalar path: a code longer than the table index, end of block, or an invalid pattern -------------------
        // It starts `off` bits into the batch.  The bits in front of it stay undropped: P is a function of the bit offset alone, so
        // behind this symbol the chain goes on in the SAME batch.
        if (pos > out_n) return false;
        const uint32_t rb0 = r.rb;
        r.rb += off;                                             // (inside the window: refill() left 8 words, a batch uses < 4)
        uint32_t e = widen(uni((uint32_t)L.lit[r.peek(kLitBits)]));
        uint32_t kind = (e >> 8) & 7u;
        if (kind == 3u) {
            const uint32_t lc = long_code(L, 0, kLitBits, L.sorted, r.peek(16), lane);
            const uint32_t sym = lc & 0xFFFFu;
            const uint32_t len = lc >> 16;
            if (len == 0u) return false;
            if (sym < 256u) e = pack(len, 0, 0, sym);
            else if (sym == 256u) e = pack(len, 2, 0, 0);
            else if (sym > 285u) e = pack(len, 4, 0, 0);
            else e = pack(len, 1, kLenExtra[sym - 257u], kLenBase[sym - 257u]);
            kind = (e >> 8) & 7u;
        }
        r.drop((int)(e & 0xFFu));
        if (kind == 0u) {                                        // (a far match's bytes may stay on their way: they lie in front of pos)
            if (pos >= out_n) return false;
            if (lane == 0) L.ring[pos & kRingMask] = (uint8_t)(e >> 16);
            ++pos;
        } else if (kind == 2u) {
            settle_far(L, far_pos, far_len, lane);
            wave_lds_fence();
            if (pos - flushed >= 256u) flushed = uni(flush_lines(L, dst, pos, flushed, lane));
            return true;
        } else if (kind == 1u) {
            const uint32_t length = (e >> 16) + r.take((int)((e >> 11) & 31u));
            uint32_t d = uni(L.dist[r.peek(kDistBits)]);
            if (((d >> 8) & 7u) == 3u) {
                const uint32_t lc = long_code(L, 1, kDistBits, L.sorted + kLitSyms, r.peek(16), lane);
                const uint32_t sym = lc & 0xFFFFu;
                if ((lc >> 16) == 0u || sym > 29u) return false;
                d = pack(lc >> 16, 1, kDistExtra[sym], kDistBase[sym]);
            }
            if (((d >> 8) & 7u) != 1u) return false;
            r.drop((int)(d & 0xFFu));
            const uint32_t dist = (d >> 16) + r.take((int)((d >> 11) & 31u));
            if (dist > pos || pos + length > out_n) return false;
            settle_far(L, far_pos, far_len, lane);
            copy_match(L, dst, pos, length, dist, lane);
            pos += length;
        } else {
            return false;
        }
        wave_lds_fence();
        off = uni(r.rb - rb0);
        r.rb = rb0;
        enter = 2;
    }
}

// status codes of a block (int32): 0 accepted; 1 not a valid / supported stream or wrong size; (2 set by the Adler pass: checksum)
__global__ __launch_bounds__(64) void inflate_kernel(const uint8_t *__restrict__ src_base, const int64_t *__restrict__ src_offsets,
                                                     const int64_t *__restrict__ src_bytes, int64_t src_capacity, uint8_t *__restrict__ dst_base, int64_t dst_pitch,
                                                     int64_t block_bytes, int32_t *__restrict__ status,
                                                     uint32_t *__restrict__ adler_expected) {
    __shared__ WaveLds L;
    const int lane = (int)threadIdx.x;
    const int64_t k = blockIdx.x;
    const int64_t s0 = src_offsets[k], nb = src_bytes[k];
    uint8_t *const dst = dst_base + k * dst_pitch;
    // a stream must lie inside the buffer, start on an 8-byte unit and be shorter than 2 GiB
    bool good = nb >= 6 && nb < (1ll << 31) && s0 >= 0 && (s0 & 7) == 0 && s0 + nb <= src_capacity;
    // block_symbols addresses the tables and the ring by their offsets inside WaveLds: the kernel's only LDS object, at address 0
    good = good && (uint32_t)(uintptr_t)&L == 0u;
    const uint32_t n_src = good ? (uint32_t)nb : 0u;
    // zlib header: CM = 8, window <= 32 K, FCHECK, no preset dictionary
    if (good) {
        const uint32_t cmf = src_base[s0], flg = src_base[s0 + 1];
        good = (cmf & 0x0Fu) == 8u && (cmf >> 4) <= 7u && ((cmf << 8) | flg) % 31u == 0u && !(flg & 0x20u);
    }
    Reader r;
    r.lane_ = lane;
    r.src = (const uint32_t *)(src_base + s0);
    r.n_words = (n_src + 3u) >> 2;
    {   // words that lie inside the buffer (the stream's last word may reach up to 3 bytes past its end)
        const int64_t room = good ? src_capacity - s0 : 0;
        r.n_words_ok = (uint32_t)((int64_t)r.n_words * 4 <= room ? (int64_t)r.n_words : room >> 2);
    }
    r.base = r.rb = r.W = r.Wn = 0u;
    uint32_t pos = 0, flushed = 0;
    uint32_t far_pos = 0, far_len = 0;                  // a far match whose bytes are still on their way from HBM (in a0, per lane: byte `lane`)
    const uint32_t out_n = (uint32_t)block_bytes;
    if (good) r.seek(2, lane);
    bool last = false;
    while (good && !last) {
        if (r.word_index() > r.n_words + 6u) { good = false; break; }   // ran past the stream's end (zeros decode to nothing useful)
        r.refill();
        last = r.take(1) != 0;
        const uint32_t type = r.take(2);
        if (type == 3) { good = false; break; }
        if (type == 0) {
            // ---- stored block: LEN, NLEN at the next byte boundary, then LEN raw bytes ------------------------------------------
            r.to_byte_boundary();
            r.refill();
            const uint32_t len = r.take(16);
            r.refill();
            const uint32_t nlen = r.take(16);
            const int64_t at = r.consumed_bytes();
            if ((len ^ nlen) != 0xFFFFu || at + (int64_t)len > (int64_t)n_src || (uint64_t)pos + len > out_n) { good = false; break; }
            const uint8_t *q = src_base + s0 + at;
            for (uint32_t done = 0; done < len; done += 64) {
                const uint32_t i = done + (uint32_t)lane;
                if (i < len) L.ring[(pos + i) & kRingMask] = q[i];
                const uint32_t step = len - done < 64u ? len - done : 64u;
                wave_lds_fence();
                // flush as we go: a stored block can be longer than the ring
                const uint32_t npos = pos + done + step;
                while (npos - flushed >= 256u) {
                    const uint32_t v = *(const uint32_t *)&L.ring[(flushed + 4u * (uint32_t)lane) & kRingMask];
                    *(uint32_t *)(dst + flushed + 4u * (uint32_t)lane) = v;
                    flushed += 256u;
                }
                wave_lds_fence();
            }
            pos += len;
            r.seek((uint32_t)(at + len), lane);
            continue;
        }
        // ---- code lengths of the block's two alphabets into L.lens[0 .. 288) and L.lens[288 .. 320) -----------------------------
        if (type == 1) {
            for (int i = lane; i < kLitSyms + kDistSyms; i += 64)
                L.lens[i] = (uint8_t)(i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5);
            wave_lds_fence();
        } else {
            r.refill();
            const int n_lit = (int)r.take(5) + 257, n_dist = (int)r.take(5) + 1, n_pre = (int)r.take(4) + 4;
            if (n_lit > 286 || n_dist > 30) { good = false; break; }
            // the code-length code's own lengths: 3 bits each, in the fixed permuted order
            uint32_t mine = 0;                                   // lane s (< 19) ends up with the length of code-length symbol s
            for (int i = 0; i < n_pre; ++i) {
                r.refill();
                const uint32_t v = r.take(3);
                if (lane == (int)kPreOrder[i]) mine = v;
            }
            if (lane < 19) L.lens[lane] = (uint8_t)mine;
            wave_lds_fence();
            if (!uni(build_table(L, 2, L.lens, 19, L.pre, 7, L.sorted, lane))) { good = false; break; }
            // the run-length coded lengths, decoded into VGPR-free LDS bytes: lens2 lives behind the two alphabets' final place,
            // so decode into a scratch area of the ring's far side?  No: the ring holds live output.  Decode straight into
            // L.lens (both alphabets back to back), then move the distance lengths to offset 288.
            int i = 0;
            const int total = n_lit + n_dist;
            uint32_t prev = 0;
            bool bad = false;
            // the 19 lengths at L.lens[0 .. 19) have served their purpose (the table is built)
            while (i < total) {
                r.refill();
                const uint32_t e = uni(L.pre[r.peek(7)]);
                if (((e >> 8) & 7u) != 0u) { bad = true; break; }
                r.drop((int)(e & 0xFFu));
                const uint32_t sym = e >> 16;
                if (sym < 16) {
                    if (lane == 0) L.lens[i] = (uint8_t)sym;
                    prev = sym;
                    ++i;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) { bad = true; break; }
                    val = prev;
                    rep = 3 + r.take(2);
                } else if (sym == 17) {
                    rep = 3 + r.take(3);
                } else {
                    rep = 11 + r.take(7);
                }
                if (i + (int)rep > total) { bad = true; break; }
                for (uint32_t j = (uint32_t)lane; j < rep; j += 64) L.lens[i + (int)j] = (uint8_t)val;
                prev = val;
                i += (int)rep;
            }
            if (bad) { good = false; break; }
            wave_lds_fence();
            // distance lengths to their fixed place, the unused tails zeroed
            uint8_t dl = 0;
            if (lane < kDistSyms && lane < n_dist) dl = L.lens[n_lit + lane];
            wave_lds_fence();
            for (int j = n_lit + lane; j < kLitSyms; j += 64) L.lens[j] = 0;
            if (lane < kDistSyms) L.lens[kLitSyms + lane] = dl;
            wave_lds_fence();
            if (uni((uint32_t)L.lens[256]) == 0u) { good = false; break; }       // no end-of-block code
        }
        if (!uni(build_table(L, 0, L.lens, kLitSyms, L.lit, kLitBits, L.sorted, lane))) { good = false; break; }
        if (!uni(build_table(L, 1, L.lens + kLitSyms, kDistSyms, L.dist, kDistBits, L.sorted + kLitSyms, lane))) { good = false; break; }

        // ---- the block's symbols ---------------------------------------------------------------------------------------------
        if (!block_symbols(L, r, dst, out_n, lane, pos, flushed, far_pos, far_len)) { good = false; break; }
    }
    settle_far(L, far_pos, far_len, lane);
    // the tail of the ring, byte by byte
    wave_lds_fence();
    if (good) {
        for (uint32_t i = flushed + (uint32_t)lane; i < pos; i += 64) dst[i] = L.ring[i & kRingMask];
    }
    const int64_t used = r.consumed_bytes();
    good = good && pos == out_n && used >= 2 && used + 4 <= (int64_t)n_src;
    if (lane == 0) {
        uint32_t want = 0;
        if (good) {
            const uint8_t *t = src_base + s0 + used;
            want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | (uint32_t)t[3];
        }
        status[k] = good ? 0 : 1;
        adler_expected[k] = want;
    }
}

// Adler-32 of every accepted block against its stream's trailer: a = 1 + sum d_j, b = N + sum (N - j) d_j (mod 65521).
// One 256-thread workgroup per block, coalesced dword reads; 64-bit sums (block_bytes <= 2^26 keeps sum j d_j below 2^63).
__global__ __launch_bounds__(256) void adler32_kernel(const uint8_t *__restrict__ dst_base, int64_t dst_pitch, int64_t block_bytes,
                                                      int32_t *__restrict__ status, const uint32_t *__restrict__ adler_expected) {
    const int64_t k = blockIdx.x;
    if (status[k] != 0) return;
    const uint8_t *d = dst_base + k * dst_pitch;
    const int64_t n_words = block_bytes >> 2;
    uint64_t A = 0, W = 0;
    for (int64_t w = threadIdx.x; w < n_words; w += 256) {
        const uint32_t v = ((const uint32_t *)d)[w];
        const uint64_t b0 = v & 0xFFu, b1 = (v >> 8) & 0xFFu, b2 = (v >> 16) & 0xFFu, b3 = v >> 24;
        const uint64_t j = (uint64_t)w * 4;
        A += b0 + b1 + b2 + b3;
        W += j * (b0 + b1 + b2 + b3) + b1 + 2 * b2 + 3 * b3;
    }
    for (int64_t j = n_words * 4 + threadIdx.x; j < block_bytes; j += 256) {
        A += d[j];
        W += (uint64_t)j * d[j];
    }
    __shared__ uint64_t sA[256], sW[256];
    sA[threadIdx.x] = A;
    sW[threadIdx.x] = W;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sA[threadIdx.x] += sA[threadIdx.x + s];
            sW[threadIdx.x] += sW[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint64_t N = (uint64_t)block_bytes, M = 65521u;
        const uint64_t a = (1 + sA[0]) % M;
        // b = N + N * A - W, each term reduced first (N * A < 2^26 * 2^34 fits; W <= N * A)
        const uint64_t b = (N % M + ((N % M) * (sA[0] % M)) % M + M - sW[0] % M) % M;
        const uint32_t got = (uint32_t)((b << 16) | a);
        if (got != adler_expected[k]) status[k] = 2;
    }
}

// ---- PNG scanline filters (PNG spec 9.2; bytes per pixel = 2) -----------------------------------------------------------
__device__ __forceinline__ int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// value of lane - 1 (lane 0 receives `fill`)
__device__ __forceinline__ int from_lane_above(int v, int fill, int lane) {
    const int got = __shfl_up(v, 1, 64);
    return lane == 0 ? fill : got;
}

// bytewise a + b (mod 256 per byte) on four bytes at once
__device__ __forceinline__ uint32_t add_bytes(uint32_t a, uint32_t b) {
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);
}

// Images whose rows use only None / Sub / Up (what an adaptive writer picks for depth maps almost always): row by row with the
// lanes ALONG the row.  Lane l owns 4 nd consecutive bytes (2 nd pixels) of every row: coalesced dword loads re-aligned by the
// row's byte phase, bytewise SWAR arithmetic, Up against the previous output row held in registers, Sub as a prefix sum -- in the
// lane, then an exclusive scan of the lanes' totals over the wave -- and coalesced dword stores of the byte-swapped samples.
// nd <= 8 (w <= 1 024), w even.
__device__ void unfilter_rows(const uint8_t *__restrict__ raw, int64_t raw_pitch, int64_t stride, int32_t h, int32_t w,
                              uint16_t *__restrict__ out, int lane, int nd) {
    uint32_t prior[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) prior[d] = 0u;
    const int64_t last = (raw_pitch - 4) & ~(int64_t)3;          // no load past the image's own block
    uint32_t v[9], vn[9];
    auto load_row = [&](int y, uint32_t *dstv) {
        const int64_t a0 = (int64_t)y * stride + 1;
        const int64_t base = (a0 & ~(int64_t)3) + 4 * (int64_t)nd * lane;
#pragma unroll
        for (int d = 0; d < 9; ++d) {
            if (d <= nd) {
                int64_t o = base + 4 * d;
                o = o < last ? o : last;
                dstv[d] = *(const uint32_t *)(raw + o);
            }
        }
    };
    load_row(0, v);
    for (int y = 0; y < h; ++y) {
        if (y + 1 < h) load_row(y + 1, vn);                      // the next row is on its way while this one is worked on
        const int64_t a0 = (int64_t)y * stride + 1;
        const uint32_t ft = raw[a0 - 1];                         // wave-uniform
        const uint32_t sh = (uint32_t)(a0 & 3);
        uint32_t x[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) x[d] = d < nd ? __builtin_amdgcn_alignbyte(v[d + 1], v[d], sh) : 0u;
        if (ft == 2u) {
#pragma unroll
            for (int d = 0; d < 8; ++d) x[d] = add_bytes(x[d], prior[d]);
        } else if (ft == 1u) {
            uint32_t acc = 0;                                    // two byte channels in the low half
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                if (d < nd) {
                    acc = add_bytes(acc, x[d] & 0xFFFFu) & 0xFFFFu;
                    const uint32_t lo = acc;
                    acc = add_bytes(acc, x[d] >> 16) & 0xFFFFu;
                    x[d] = lo | (acc << 16);
                }
            }
            uint32_t incl = acc;                                 // inclusive scan of the lanes' totals, bytewise
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, sft, 64);
                if (lane >= sft) incl = add_bytes(incl, t) & 0xFFFFu;
            }
            uint32_t excl = (uint32_t)__shfl_up((int)incl, 1, 64);
            excl = lane == 0 ? 0u : excl;
            const uint32_t both = excl | (excl << 16);
#pragma unroll
            for (int d = 0; d < 8; ++d) x[d] = add_bytes(x[d], both);
        }
        uint32_t *orow = (uint32_t *)(out + (int64_t)y * w);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            prior[d] = x[d];
            if (d < nd && 2 * (nd * lane + d) < w)
                orow[nd * lane + d] = ((x[d] & 0x00FF00FFu) << 8) | ((x[d] >> 8) & 0x00FF00FFu);     // big-endian samples -> host order
        }
#pragma unroll
        for (int d = 0; d < 9; ++d) v[d] = vn[d];
    }
}

// One 16-bit pixel = two byte channels, worked on TOGETHER as two 16-bit halves of a register (v_pk_* instructions): raw +
// predictor(left a, up b, upper-left c) by filter type (PNG 9.2), each half 0 .. 255.  Paeth's three distances are
// |b - c|, |a - c|, |(b - c) + (a - c)|; "x <= y" per half is the sign of y - x spread over the half by an arithmetic shift, the
// choice a bit-field insert.  Every filter's predictor is computed and selected (the lanes of a band hold rows of different
// types); `all_paeth` (wave-uniform) skips the selection.
typedef short pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t as_u32(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ pk16 as_pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
__device__ __forceinline__ pk16 pk_abs(pk16 v) { return __builtin_elementwise_max(v, -v); }
__device__ __forceinline__ uint32_t unfilter_pixel(uint32_t raw, uint32_t a, uint32_t b, uint32_t c, int ft, bool all_paeth) {
    const pk16 A = as_pk(a), B = as_pk(b), C = as_pk(c);
    const pk16 t = B - C, u = A - C;
    const pk16 pa = pk_abs(t), pb = pk_abs(u), pc = pk_abs(t + u);
    const uint32_t a_gt_b = as_u32((pb - pa) >> 15), a_gt_c = as_u32((pc - pa) >> 15), b_gt_c = as_u32((pc - pb) >> 15);   // all ones where pa > pb, ...
    const uint32_t bc = (c & b_gt_c) | (b & ~b_gt_c);            // pb <= pc ? b : c
    const uint32_t not_a = a_gt_b | a_gt_c;
    const uint32_t pae = (bc & not_a) | (a & ~not_a);            // pa <= pb && pa <= pc ? a : bc
    uint32_t pred = pae;
    if (!all_paeth) {
        const uint32_t avg = as_u32(as_pk(as_u32(A + B)) >> 1);  // halves are 0 .. 510: the arithmetic shift is the logical one
        pred = ft == 4 ? pae : ft == 3 ? avg : ft == 2 ? b : ft == 1 ? a : 0u;
    }
    return as_u32(as_pk(raw) + as_pk(pred)) & 0x00FF00FFu;
}

constexpr int kUpRowPairs = 2048;                                // widest image of the pair pipeline: 4 096 pixels
// up_row: the row above the band (lane 0's upper neighbour), staged in LDS once per band: read from memory step by step it cost
// the whole wave one L2 round trip per step
__device__ void unfilter_skewed_pairs(const uint8_t *__restrict__ raw, int64_t raw_pitch, int64_t stride, int32_t h, int32_t w,
                                      uint16_t *__restrict__ out, int lane, uint32_t *up_row) {
    const int wp = w >> 1;                                       // pixel pairs per row
    const int64_t last = (raw_pitch - 4) & ~(int64_t)3;          // no load past the image's own block
    uint32_t *const out32 = (uint32_t *)out;
    for (int y0 = 0; y0 < h; y0 += 64) {
        const int y = y0 + lane;
        const bool row = y < h;
        const int64_t a0 = (int64_t)(row ? y : 0) * stride + 1;  // the row's first sample byte
        const int ft = row ? (int)raw[a0 - 1] : 0;
        const bool all_paeth = __all(!row || ft == 4);
        const int64_t base = a0 & ~(int64_t)3;
        const uint32_t sh = (uint32_t)(a0 & 3);
        if (y0 > 0) {                                            // the row above the band was written by this wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const bool has_up = y0 > 0;
        if (has_up) {
            const uint32_t *uprow = out32 + (int64_t)(y0 - 1) * wp;
            for (int k = lane; k < wp; k += 64) up_row[k] = uprow[k];
            wave_lds_fence();
        }
        // The row's dwords go through EIGHT registers that change roles from step to step (the loop is unrolled by eight): a
        // register is loaded seven steps (~3 000 cycles: an HBM miss, which some lane of the wave has at nearly every step)
        // before it is used and never copied -- copying the freshly loaded register into the
        // next role, as a rolled loop must, waits for the load it has just issued, one memory round trip per step (what the
        // first pair version did: 1.7 ms per image for a lone wave).  To let every lane rotate in the same step, all of them
        // load "dword i - lane" at step i, whether their pair is inside the row yet or not (addresses clamped into the block).
        auto ld = [&](int k) {                                   // dword k of this lane's row (k < 0: bytes in front of it, unused)
            int64_t o = base + 4 * (int64_t)k;
            o = o < 0 ? 0 : (o < last ? o : last);
            return *(const uint32_t *)(raw + o);
        };
        uint32_t upn = has_up ? up_row[0] : 0u;
        uint32_t X = 0;                                          // this lane's last output pair, bytes in stream order: x0 hi, x0 lo, x1 hi, x1 lo
        uint32_t Uprev = 0;                                      // the pair above, one step earlier (its second pixel: upper-left)
        const int steps = wp + 63;
        auto step = [&](int i, uint32_t wc, uint32_t wn, uint32_t &reload) {
            const int j = i - lane;
            const bool on = row && j >= 0 && j < wp;
            reload = ld(j + 8);                                  // dwords j, j + 1 are wc, wn; j + 2 .. j + 7 on their way
            // the lane above finished pair j one step ago: its output through a DPP wave shift (one vector instruction; a
            // ds_bpermute would put an LDS round trip on every step's critical path)
            uint32_t U = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)X, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
            if (lane == 0) U = ((upn & 0x00FF00FFu) << 8) | ((upn >> 8) & 0x00FF00FFu);      // host-order samples -> stream order
            if (has_up) upn = up_row[i + 1 < wp ? i + 1 : 0];   // (wave-uniform address: one LDS broadcast read, a step ahead)
            if (on) {
                const uint32_t d = __builtin_amdgcn_alignbyte(wn, wc, sh);
                // pixels as (hi | lo << 16): stream bytes 0, 1 -> pixel 0, bytes 2, 3 -> pixel 1
                const uint32_t r0 = __builtin_amdgcn_perm(0u, d, 0x0C010C00u), r1 = __builtin_amdgcn_perm(0u, d, 0x0C030C02u);
                const uint32_t u0 = __builtin_amdgcn_perm(0u, U, 0x0C010C00u), u1 = __builtin_amdgcn_perm(0u, U, 0x0C030C02u);
                const uint32_t ul = __builtin_amdgcn_perm(0u, Uprev, 0x0C030C02u);       // the pair above, one step earlier: its pixel 1
                const uint32_t xl = __builtin_amdgcn_perm(0u, X, 0x0C030C02u);           // this lane's last output: its pixel 1
                const uint32_t x0 = unfilter_pixel(r0, xl, u0, ul, ft, all_paeth);
                const uint32_t x1 = unfilter_pixel(r1, x0, u1, u0, ft, all_paeth);
                X = __builtin_amdgcn_perm(x1, x0, 0x06040200u);                          // bytes x0 hi, x0 lo, x1 hi, x1 lo
                out32[(int64_t)y * wp + j] = __builtin_amdgcn_perm(x1, x0, 0x04060002u); // two host-order samples: x0 lo, x0 hi, x1 lo, x1 hi
            }
            Uprev = U;
        };
        uint32_t w0 = ld(0 - lane), w1 = ld(1 - lane), w2 = ld(2 - lane), w3 = ld(3 - lane);     // step 0's dwords j .. j + 7
        uint32_t w4 = ld(4 - lane), w5 = ld(5 - lane), w6 = ld(6 - lane), w7 = ld(7 - lane);
        int i = 0;
        for (; i + 7 < steps; i += 8) {
            step(i, w0, w1, w0);
            step(i + 1, w1, w2, w1);
            step(i + 2, w2, w3, w2);
            step(i + 3, w3, w4, w3);
            step(i + 4, w4, w5, w4);
            step(i + 5, w5, w6, w5);
            step(i + 6, w6, w7, w6);
            step(i + 7, w7, w0, w7);
        }
        for (int r = 0; i < steps; ++i, ++r) {                   // the last one to seven steps
            switch (r) {
            case 0: step(i, w0, w1, w0); break;
            case 1: step(i, w1, w2, w1); break;
            case 2: step(i, w2, w3, w2); break;
            case 3: step(i, w3, w4, w3); break;
            case 4: step(i, w4, w5, w4); break;
            case 5: step(i, w5, w6, w5); break;
            default: step(i, w6, w7, w6); break;
            }
        }
    }
}

__global__ __launch_bounds__(64) void png_unfilter_kernel(const uint8_t *__restrict__ raw_base, int64_t raw_pitch, int32_t h, int32_t w,
                                                          uint16_t *__restrict__ out_base, int32_t *__restrict__ status) {
    const int64_t k = blockIdx.x;
    if (status[k] != 0) return;
    const int lane = (int)threadIdx.x;
    const uint8_t *raw = raw_base + k * raw_pitch;
    uint16_t *out = out_base + k * (int64_t)h * w;
    const int64_t stride = (int64_t)w * 2 + 1;
    // the rows' filter types: anything above Paeth is not PNG; Average / Paeth anywhere sends the image down the skewed pipeline
    bool bad = false, hard = false;
    for (int y = lane; y < h; y += 64) {
        const int ft = raw[(int64_t)y * stride];
        bad = bad || ft > 4;
        hard = hard || ft > 2;
    }
    if (__any(bad)) {
        if (lane == 0) status[k] = 3;
        return;
    }
    const int nd = (w + 127) / 128;
    if (!__any(hard) && (w & 1) == 0 && nd <= 8 && (((uintptr_t)out) & 3u) == 0) {
        unfilter_rows(raw, raw_pitch, stride, h, w, out, lane, nd);
        return;
    }
    // Average / Paeth rows: the second launch (png_unfilter_hard_kernel) takes the image
    if (lane == 0) status[k] = kPendingHard;
}

// Images with Average / Paeth rows.  A launch of its own because it wants FEW waves per CU: every lane streams its own row, so a
// wave keeps 128 cache lines open (64 rows read, 64 written) for 32 steps each -- 3 584 such waves (14 per CU) are 59 MB of open
// lines against 32 MB of L2, and partly written lines travel to the memory side and back (16.3 ms per 3 584 smooth frames; with
// the waves per CU held to 4 by this kernel's 40 KB of LDS: 10.2 ms, profiles/r06_device_ingest.md).
__global__ __launch_bounds__(64) void png_unfilter_hard_kernel(const uint8_t *__restrict__ raw_base, int64_t raw_pitch, int32_t h, int32_t w,
                                                               uint16_t *__restrict__ out_base, int32_t *__restrict__ status) {
    __shared__ uint32_t lds[10240];                              // 40 KB: 2 048 dwords of it hold the row above the band
    const int64_t k = blockIdx.x;
    if (status[k] != kPendingHard) return;
    const int lane = (int)threadIdx.x;
    const uint8_t *raw = raw_base + k * raw_pitch;
    uint16_t *out = out_base + k * (int64_t)h * w;
    const int64_t stride = (int64_t)w * 2 + 1;
    if ((w & 1) == 0 && w >= 4 && w <= 2 * kUpRowPairs && (((uintptr_t)out) & 3u) == 0 && (((uintptr_t)raw) & 3u) == 0) {
        unfilter_skewed_pairs(raw, raw_pitch, stride, h, w, out, lane, lds);
        if (lane == 0) status[k] = 0;
        return;
    }
    for (int y0 = 0; y0 < h; y0 += 64) {
        const int y = y0 + lane;
        const bool row = y < h;
        const uint8_t *rp = raw + (int64_t)(row ? y : 0) * stride;
        const int ft = row ? (int)rp[0] : 0;
        uint16_t *op = out + (int64_t)(row ? y : 0) * w;
        const uint16_t *up = y0 > 0 ? out + (int64_t)(y0 - 1) * w : nullptr;      // the row above the band (lane 0's neighbour)
        if (y0 > 0) {                                                              // written by this wave in the previous band
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        int a_hi = 0, a_lo = 0;              // left pixel of this row
        int x_hi = 0, x_lo = 0;              // this lane's most recent output (what the lane below reads as "up")
        int b_hi = 0, b_lo = 0;              // up pixel received at the previous step = upper-left of the current one
        const int steps = w + 63;
        for (int t = 0; t < steps; ++t) {
            const int col = t - lane;
            const bool on = row && col >= 0 && col < w;
            // the lane above worked on this very column one step ago: its latest output is our "up"
            int top_hi = 0, top_lo = 0;
            if (up != nullptr && t < w) {                                          // lane 0: column t of the row above the band
                const uint16_t v = up[t];
                top_hi = v >> 8;
                top_lo = v & 0xFF;
            }
            const int nb_hi = from_lane_above(x_hi, top_hi, lane), nb_lo = from_lane_above(x_lo, top_lo, lane);
            const int c_hi = b_hi, c_lo = b_lo;                                    // received one step earlier: column col - 1
            const bool has_up = y > 0;
            const int ub_hi = has_up ? nb_hi : 0, ub_lo = has_up ? nb_lo : 0;
            const int uc_hi = (has_up && col > 0) ? c_hi : 0, uc_lo = (has_up && col > 0) ? c_lo : 0;
            if (on) {
                const int r_hi = rp[1 + 2 * col], r_lo = rp[2 + 2 * col];
                int p_hi, p_lo;
                if (ft == 0) { p_hi = 0; p_lo = 0; }
                else if (ft == 1) { p_hi = a_hi; p_lo = a_lo; }
                else if (ft == 2) { p_hi = ub_hi; p_lo = ub_lo; }
                else if (ft == 3) { p_hi = (a_hi + ub_hi) >> 1; p_lo = (a_lo + ub_lo) >> 1; }
                else { p_hi = paeth(a_hi, ub_hi, uc_hi); p_lo = paeth(a_lo, ub_lo, uc_lo); }
                x_hi = (r_hi + p_hi) & 0xFF;
                x_lo = (r_lo + p_lo) & 0xFF;
                a_hi = x_hi;
                a_lo = x_lo;
                op[col] = (uint16_t)((x_hi << 8) | x_lo);
            }
            b_hi = nb_hi;
            b_lo = nb_lo;
        }
    }
    if (lane == 0) status[k] = 0;
}

}  // namespace dinf
}  // namespace mspa

using namespace mspa;

extern "C" int mspa_inflate_blocks_device(const void *src_dev, const int64_t *src_offsets_dev, const int64_t *src_bytes_dev,
                                          int64_t src_capacity, int64_t n_blocks, int64_t block_bytes, void *dst_dev,
                                          int64_t dst_pitch, int32_t *status_dev, uint32_t *work_dev, void *stream) {
    if (n_blocks < 0 || block_bytes <= 0 || dst_pitch < block_bytes || src_capacity < 0)
        return fail(MSPA_EINVAL, "mspa_inflate_blocks_device: bad size");
    if (n_blocks == 0) return MSPA_OK;
    if (!src_dev || !src_offsets_dev || !src_bytes_dev || !dst_dev || !status_dev || !work_dev)
        return fail(MSPA_EINVAL, "mspa_inflate_blocks_device: null pointer");
    if (block_bytes > (1ll << 26)) return fail(MSPA_EINVAL, "mspa_inflate_blocks_device: block larger than 64 MiB");
    if (n_blocks > 0x7fffffffll) return fail(MSPA_EINVAL, "mspa_inflate_blocks_device: too many blocks");
    // dst: lines of 256 bytes leave the ring whole, and a cache line must never hold bytes from both sides of that frontier
    if (((uintptr_t)src_dev & 15u) || ((uintptr_t)dst_dev & 255u) || (dst_pitch & 255))
        return fail(MSPA_EINVAL, "mspa_inflate_blocks_device: src must be 16-byte, dst and dst_pitch 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dinf::inflate_kernel, dim3((unsigned)n_blocks), dim3(64), 0, st, (const uint8_t *)src_dev, src_offsets_dev,
                       src_bytes_dev, src_capacity, (uint8_t *)dst_dev, dst_pitch, block_bytes, status_dev, work_dev);
    hipLaunchKernelGGL(dinf::adler32_kernel, dim3((unsigned)n_blocks), dim3(256), 0, st, (const uint8_t *)dst_dev, dst_pitch, block_bytes,
                       status_dev, (const uint32_t *)work_dev);
    return check_hip(hipGetLastError(), "mspa_inflate_blocks_device");
}

extern "C" int mspa_png_unfilter_device(const void *raw_dev, int64_t raw_pitch, int64_t n_images, int32_t h, int32_t w,
                                        uint16_t *out_dev, int32_t *status_dev, void *stream) {
    if (n_images < 0 || h <= 0 || w <= 0 || raw_pitch < (int64_t)h * ((int64_t)w * 2 + 1))
        return fail(MSPA_EINVAL, "mspa_png_unfilter_device: bad size");
    if (n_images == 0) return MSPA_OK;
    if (!raw_dev || !out_dev || !status_dev) return fail(MSPA_EINVAL, "mspa_png_unfilter_device: null pointer");
    if (n_images > 0x7fffffffll) return fail(MSPA_EINVAL, "mspa_png_unfilter_device: too many images");
    hipLaunchKernelGGL(dinf::png_unfilter_kernel, dim3((unsigned)n_images), dim3(64), 0, (hipStream_t)stream, (const uint8_t *)raw_dev,
                       raw_pitch, h, w, out_dev, status_dev);
    hipLaunchKernelGGL(dinf::png_unfilter_hard_kernel, dim3((unsigned)n_images), dim3(64), 0, (hipStream_t)stream, (const uint8_t *)raw_dev,
                       raw_pitch, h, w, out_dev, status_dev);
    return check_hip(hipGetLastError(), "mspa_png_unfilter_device");
}
