// A table-driven DEFLATE (RFC 1951) / zlib (RFC 1950) decoder for the host-side ingest (no device code here).
//
// Why not zlib's own inflate: the from-disk scene sweep is bound by it.  A 640 x 480 16-bit depth frame is 1.2 MB of filtered
// scanlines that deflate only 2 : 1 -- literal-heavy streams -- and zlib's inflate_fast delivers ~155 MB/s of output per thread
// on them (4 ms of a frame's 5 ms; bench.py `dropin_sweep`), against 0.2 ms of kernels per SCENE.  This decoder is written for
// exactly this use: the whole compressed stream and the whole output buffer are in memory, the output size is known, so there
// is no streaming state machine -- a 64-bit bit buffer refilled by unaligned 8-byte loads, one 11-bit primary table lookup per
// literal / length code (2 048 entries, sub-tables for the longer codes) and one 8-bit lookup per distance code, literals stored
// without a bounds test inside the fast loop's margins, matches copied 8 bytes at a time.
//
// Contract: mspa::inflate_zlib(src, n_src, dst, n_dst) returns true iff src is a complete zlib stream that inflates to EXACTLY
// n_dst bytes AND its Adler-32 matches -- a stream this decoder gets wrong cannot pass (the callers then hand the frame to zlib
// itself).  Restates the published algorithm; checked against zlib on every compression level / strategy, stored and
// fixed-Huffman blocks, long matches, overlapping copies and truncated / corrupted streams by tests/test_sweep_cpu.py.
#pragma once
#include <cstdint>
#include <cstring>

namespace mspa {
namespace inflate_detail {

constexpr int kLitBits = 11;                   // primary table bits, literal / length alphabet
constexpr int kDistBits = 8;                   // primary table bits, distance alphabet
constexpr int kMaxCodeLen = 15;
constexpr int kLitSyms = 288, kDistSyms = 32, kPreSyms = 19;
// worst-case table sizes for these primary widths (zlib's enough.c: 852 / 592 for 9 / 6 bits; generous here)
constexpr int kLitTableSize = (1 << kLitBits) + 1024;
constexpr int kDistTableSize = (1 << kDistBits) + 512;

// Table entry (uint32): bits 0..7 code length to consume from the bit buffer, bits 8..15 kind, bits 16..31 payload.
//   kind 0  literal            payload = the byte
//   kind 1  length / distance  payload = base value, extra-bit count in bits 8..12 of `kind`'s field (see pack)
//   kind 2  end of block
//   kind 3  sub-table link     payload = sub-table offset, "length" = primary bits, extra = sub-table bits
//   kind 4  invalid code
struct Entry {
    uint32_t v;
};
inline Entry pack(uint32_t len, uint32_t kind, uint32_t extra, uint32_t payload) {
    return Entry{len | (kind << 8) | (extra << 11) | (payload << 16)};
}
inline uint32_t e_len(Entry e) { return e.v & 0xFFu; }
inline uint32_t e_kind(Entry e) { return (e.v >> 8) & 7u; }
inline uint32_t e_extra(Entry e) { return (e.v >> 11) & 31u; }
inline uint32_t e_payload(Entry e) { return e.v >> 16; }

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

inline uint32_t reverse_bits(uint32_t code, int len) {
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) r |= ((code >> i) & 1u) << (len - 1 - i);
    return r;
}

// Canonical Huffman decode table from code lengths.  `kind_of(sym)` gives the entry for a symbol at a given code length.
// Returns false for an over-subscribed set; an incomplete set is accepted (unused patterns decode to "invalid"), which is what
// zlib permits for a single distance code and what fixed / real streams never exercise otherwise.
template <typename MakeEntry>
bool build_table(const uint8_t *lens, int n_syms, int primary_bits, Entry *table, int table_cap, MakeEntry make) {
    int count[kMaxCodeLen + 1] = {0};
    for (int s = 0; s < n_syms; ++s) count[lens[s]]++;
    count[0] = 0;
    uint32_t next_code[kMaxCodeLen + 2];
    uint32_t code = 0;
    int64_t left = 1;
    for (int l = 1; l <= kMaxCodeLen; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;                            // over-subscribed
        code = (code + (uint32_t)count[l - 1]) << 1;
        next_code[l] = code;
    }
    const int psize = 1 << primary_bits;
    for (int i = 0; i < psize; ++i) table[i] = pack(1, 4, 0, 0);
    int next_sub = psize;
    // codes no longer than the primary width: replicated over the unused high bits
    // codes longer: grouped by their low `primary_bits` bits (bit-reversed: DEFLATE packs codes MSB first into an LSB-first stream)
    // first pass: sizes of the sub-tables (max code length per prefix)
    static thread_local uint8_t sub_bits[1 << kLitBits];
    memset(sub_bits, 0, (size_t)psize);
    {
        uint32_t nc[kMaxCodeLen + 2];
        memcpy(nc, next_code, sizeof nc);
        for (int s = 0; s < n_syms; ++s) {
            const int l = lens[s];
            if (l <= primary_bits) {
                if (l) nc[l]++;
                continue;
            }
            const uint32_t rev = reverse_bits(nc[l]++, l);
            const uint32_t prefix = rev & (uint32_t)(psize - 1);
            if ((int)sub_bits[prefix] < l - primary_bits) sub_bits[prefix] = (uint8_t)(l - primary_bits);
        }
    }
    static thread_local uint16_t sub_off[1 << kLitBits];
    for (int p = 0; p < psize; ++p) {
        if (!sub_bits[p]) continue;
        const int sz = 1 << sub_bits[p];
        if (next_sub + sz > table_cap) return false;
        sub_off[p] = (uint16_t)next_sub;
        for (int i = 0; i < sz; ++i) table[next_sub + i] = pack(1, 4, 0, 0);
        table[p] = pack((uint32_t)primary_bits, 3, sub_bits[p], (uint32_t)next_sub);
        next_sub += sz;
    }
    for (int s = 0; s < n_syms; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t rev = reverse_bits(next_code[l]++, l);
        if (l <= primary_bits) {
            const Entry e = make(s, l);
            for (uint32_t i = rev; i < (uint32_t)psize; i += 1u << l) table[i] = e;
        } else {
            const uint32_t prefix = rev & (uint32_t)(psize - 1);
            const int sb = sub_bits[prefix];
            const uint32_t hi = rev >> primary_bits;
            const Entry e = make(s, l - primary_bits);
            for (uint32_t i = hi; i < (1u << sb); i += 1u << (l - primary_bits)) table[sub_off[prefix] + i] = e;
        }
    }
    return true;
}

inline Entry lit_entry(int s, int l) {
    if (s < 256) return pack((uint32_t)l, 0, 0, (uint32_t)s);
    if (s == 256) return pack((uint32_t)l, 2, 0, 0);
    if (s > 285) return pack((uint32_t)l, 4, 0, 0);
    return pack((uint32_t)l, 1, kLenExtra[s - 257], kLenBase[s - 257]);
}
inline Entry dist_entry(int s, int l) {
    if (s > 29) return pack((uint32_t)l, 4, 0, 0);
    return pack((uint32_t)l, 1, kDistExtra[s], kDistBase[s]);
}

struct Bits {
    const uint8_t *p, *end;
    uint64_t buf = 0;
    int n = 0;                                               // valid bits in buf; negative once more bits were consumed than the input held
    // >= 56 valid bits whenever 8 input bytes remain (bits of not-yet-counted bytes may already sit above n: they are OR-ed in
    // again, identically, by the next refill); near the end of the input byte by byte, then nothing: reads past the end see
    // zeros and leave n negative, which every caller checks before it trusts what it decoded
    inline void refill() {
        if (end - p >= 8) {
            uint64_t w;
            memcpy(&w, p, 8);                                  // little-endian host
            buf |= w << n;
            const int take = (63 - n) >> 3;
            p += take;
            n += take * 8;
        } else {
            while (n <= 56 && p < end) {
                buf |= (uint64_t)(*p++) << n;
                n += 8;
            }
        }
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1ull)); }
    inline void drop(int k) {
        buf >>= k;
        n -= k;
    }
    inline uint32_t take(int k) {
        const uint32_t v = peek(k);
        drop(k);
        return v;
    }
    inline bool overrun() const { return n < 0; }
};

}  // namespace inflate_detail

inline uint32_t adler32_of(const uint8_t *d, size_t n) {
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;                        // largest run without overflow of the 32-bit sums
        n -= k;
        while (k >= 8) {
            a += d[0]; b += a; a += d[1]; b += a; a += d[2]; b += a; a += d[3]; b += a;
            a += d[4]; b += a; a += d[5]; b += a; a += d[6]; b += a; a += d[7]; b += a;
            d += 8;
            k -= 8;
        }
        while (k--) {
            a += *d++;
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}

// Raw DEFLATE: src[0 .. n_src) -> dst[0 .. n_dst).  True iff the stream ends exactly at n_dst bytes of output.
// *consumed (optional): bytes of src the stream occupied (rounded up to a whole byte).
inline bool inflate_raw(const uint8_t *src, size_t n_src, uint8_t *dst, size_t n_dst, size_t *consumed = nullptr) {
    using namespace inflate_detail;
    static thread_local Entry lit_table_tls[kLitTableSize];
    static thread_local Entry dist_table_tls[kDistTableSize];
    Entry *const lit_table = lit_table_tls, *const dist_table = dist_table_tls;     // the thread's tables: their address taken ONCE
    Bits br{src, src + n_src};                                                       // (in a shared library every TLS access is a call)
    uint8_t *out = dst, *const out_end = dst + n_dst;
    bool last = false;
    while (!last) {
        br.refill();
        last = br.take(1);
        const uint32_t type = br.take(2);
        if (type == 0) {                                       // stored
            br.drop(br.n & 7);                                 // to a byte boundary
            br.refill();
            const uint32_t len = br.take(16), nlen = br.take(16);
            if ((len ^ nlen) != 0xFFFFu || br.overrun()) return false;
            // give the whole bytes still in the bit buffer back to the byte stream
            const uint8_t *q = br.p - (br.n >> 3);
            if ((size_t)(br.end - q) < len || (size_t)(out_end - out) < len) return false;
            memcpy(out, q, len);
            out += len;
            br.p = q + len;
            br.buf = 0;
            br.n = 0;
            continue;
        }
        if (type == 3) return false;
        uint8_t lens[kLitSyms + kDistSyms];
        int n_lit, n_dist;
        if (type == 1) {                                       // fixed Huffman codes
            n_lit = 288;
            n_dist = 32;
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else {                                               // dynamic: the code-length code first
            br.refill();
            n_lit = (int)br.take(5) + 257;
            n_dist = (int)br.take(5) + 1;
            const int n_pre = (int)br.take(4) + 4;
            if (n_lit > 286 || n_dist > 30) return false;
            uint8_t pre_lens[kPreSyms] = {0};
            for (int i = 0; i < n_pre; ++i) {
                if (br.n < 3) br.refill();
                pre_lens[kPreOrder[i]] = (uint8_t)br.take(3);
            }
            Entry pre_table[1 << 7];
            if (!build_table(pre_lens, kPreSyms, 7, pre_table, 1 << 7, [](int s, int l) { return pack((uint32_t)l, 0, 0, (uint32_t)s); }))
                return false;
            int i = 0;
            while (i < n_lit + n_dist) {
                br.refill();
                const Entry e = pre_table[br.peek(7)];
                if (e_kind(e) != 0) return false;
                br.drop((int)e_len(e));
                const uint32_t sym = e_payload(e);
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return false;
                    val = lens[i - 1];
                    rep = 3 + br.take(2);
                } else if (sym == 17) {
                    rep = 3 + br.take(3);
                } else {
                    rep = 11 + br.take(7);
                }
                if (i + (int)rep > n_lit + n_dist) return false;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (br.overrun() || lens[256] == 0) return false;
            // the two alphabets sit back to back in `lens`; move the distance lengths to their fixed place
            uint8_t dl[kDistSyms] = {0};
            memcpy(dl, lens + n_lit, (size_t)n_dist);
            memset(lens + n_lit, 0, (size_t)(kLitSyms - n_lit));
            memcpy(lens + kLitSyms, dl, kDistSyms);
            n_lit = kLitSyms;
            n_dist = kDistSyms;
        }
        if (!build_table(lens, n_lit, kLitBits, lit_table, kLitTableSize, lit_entry)) return false;
        if (!build_table(lens + kLitSyms, n_dist, kDistBits, dist_table, kDistTableSize, dist_entry)) return false;

        // ---- the block's symbols ----------------------------------------------------------------------------------------------
        for (;;) {
            br.refill();                                       // >= 56 bits: a literal / length code (<= 15 + 5) and a distance
            Entry e = lit_table[br.peek(kLitBits)];            // code (<= 15 + 13) fit without another refill
            if (e_kind(e) == 3) {
                br.drop(kLitBits);
                e = lit_table[e_payload(e) + br.peek((int)e_extra(e))];
            }
            br.drop((int)e_len(e));
            const uint32_t kind = e_kind(e);
            if (kind == 0) {                                   // literal; try a second and third one from the same refill
                if (out >= out_end) return false;
                *out++ = (uint8_t)e_payload(e);
                Entry e2 = lit_table[br.peek(kLitBits)];
                if (e_kind(e2) == 0 && out < out_end) {
                    br.drop((int)e_len(e2));
                    *out++ = (uint8_t)e_payload(e2);
                    e2 = lit_table[br.peek(kLitBits)];
                    if (e_kind(e2) == 0 && out < out_end) {
                        br.drop((int)e_len(e2));
                        *out++ = (uint8_t)e_payload(e2);
                    }
                }
                continue;
            }
            if (kind == 2) break;                              // end of block
            if (kind != 1) return false;
            const uint32_t length = e_payload(e) + br.take((int)e_extra(e));
            Entry d = dist_table[br.peek(kDistBits)];
            if (e_kind(d) == 3) {
                br.drop(kDistBits);
                d = dist_table[e_payload(d) + br.peek((int)e_extra(d))];
            }
            if (e_kind(d) != 1) return false;
            br.drop((int)e_len(d));
            if (br.n < (int)e_extra(d)) br.refill();
            const uint32_t dist = e_payload(d) + br.take((int)e_extra(d));
            if (dist > (size_t)(out - dst) || length > (size_t)(out_end - out)) return false;
            const uint8_t *from = out - dist;
            if (dist >= 8 && (size_t)(out_end - out) >= length + 8) {      // 8 bytes at a time; may write up to 7 bytes past the match
                uint8_t *o = out;
                const uint8_t *const stop = out + length;
                do {
                    uint64_t w;
                    memcpy(&w, from, 8);
                    memcpy(o, &w, 8);
                    from += 8;
                    o += 8;
                } while (o < stop);
            } else {
                for (uint32_t k = 0; k < length; ++k) out[k] = from[k];     // overlapping (run-length) copies included
            }
            out += length;
        }
        if (br.overrun()) return false;
    }
    if (out != out_end) return false;
    if (consumed) *consumed = (size_t)(br.p - src) - (size_t)(br.n >> 3);
    return true;
}

// zlib stream (2-byte header, DEFLATE, Adler-32 of the output, big-endian).
inline bool inflate_zlib(const uint8_t *src, size_t n_src, uint8_t *dst, size_t n_dst) {
    if (n_src < 6) return false;
    const uint32_t cmf = src[0], flg = src[1];
    if ((cmf & 0x0F) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return false;   // no preset dictionary
    size_t used = 0;
    if (!inflate_raw(src + 2, n_src - 2, dst, n_dst, &used)) return false;
    if (n_src - 2 - used < 4) return false;
    const uint8_t *t = src + 2 + used;
    const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | (uint32_t)t[3];
    return adler32_of(dst, n_dst) == want;
}

}  // namespace mspa
