// Host-side ingest of a scene's depth frames from disk (no device work in this file).
//
// The reference reads one depth frame per call with `cv2.imread(path, -1)` (info_handler.py:149-155), inside the per-image
// loop of CFR.process_scene / MVI.process_scene (CFR:152-157, MVI:93-100) -- and fans SCENES out over a process pool to
// hide it (CFR:222-229, MVI:151-156).  Here one process feeds one GPU, so the frames of a scene are read and decoded by
// native threads that never touch the interpreter: `mspa_read_depth_png_host` takes the file names of a scene's frames
// and fills one contiguous [n, h, w] uint16 block -- what the H2D staging (mspa/upload.py) consumes.
//
// Format handled natively: what ScanNet's exporter writes (extract_posed_images.py:118-123, imageio.imwrite of a
// uint16 array): PNG, 16-bit greyscale (colour type 0), non-interlaced, any zlib level, any mix of the five row
// filters.  Anything else gets status 2 and the caller decodes that one frame with its general reader.
#include "mspa_common.h"
#include "inflate_fast.h"
#include "host_pool.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>
#include <thread>
#include <vector>
#include <zlib.h>

namespace mspa {

namespace {

inline uint32_t be32(const unsigned char *p) {
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

const unsigned char kPngSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

struct PngHeader {
    uint32_t w = 0, h = 0;
    int bit_depth = 0, color_type = 0, interlace = 0;
};

// 0 ok, 3 corrupt
int parse_header(const unsigned char *buf, size_t n, PngHeader &hd) {
    if (n < 33 || memcmp(buf, kPngSig, 8) != 0) return 3;
    if (be32(buf + 8) != 13 || memcmp(buf + 12, "IHDR", 4) != 0) return 3;
    hd.w = be32(buf + 16);
    hd.h = be32(buf + 20);
    hd.bit_depth = buf[24];
    hd.color_type = buf[25];
    hd.interlace = buf[28];
    return 0;
}

bool read_file(const char *path, std::vector<unsigned char> &buf) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    bool ok = false;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long sz = ftell(f);
        if (sz >= 0 && fseek(f, 0, SEEK_SET) == 0) {
            buf.resize((size_t)sz);
            ok = sz == 0 || fread(buf.data(), 1, (size_t)sz, f) == (size_t)sz;
        }
    }
    fclose(f);
    return ok;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// One 16-bit greyscale frame: file bytes -> dst[h * w] (host byte order).  `raw` is the thread's scratch buffer.
// 0 ok, 2 not the native format / size, 3 corrupt.
int decode_gray16(const unsigned char *buf, size_t n, int32_t h, int32_t w, uint16_t *dst, std::vector<unsigned char> &raw,
                  std::vector<unsigned char> &idat) {
    PngHeader hd;
    if (int rc = parse_header(buf, n, hd)) return rc;
    if (hd.bit_depth != 16 || hd.color_type != 0 || hd.interlace != 0 || hd.w != (uint32_t)w || hd.h != (uint32_t)h) return 2;
    const size_t stride = (size_t)w * 2 + 1;
    raw.resize(stride * (size_t)h);
    // The scanlines' zlib stream: the IDAT chunks' payloads back to back.  One chunk (what most writers emit for a frame of this
    // size is several 8-64 KB chunks; a single one needs no copy) or several gathered into `idat`; first through the
    // table-driven decoder (inflate_fast.h: exact output size + Adler-32 or it reports failure), then, for a stream it
    // declines, through zlib itself.
    size_t pos = 33;                                  // past the signature and IHDR (8 + 4 + 4 + 13 + 4)
    const unsigned char *one = nullptr;
    size_t one_len = 0;
    int n_idat = 0;
    bool bad = false;
    idat.clear();
    while (pos + 12 <= n) {
        const uint32_t len = be32(buf + pos);
        const unsigned char *type = buf + pos + 4;
        if ((size_t)len > n - pos - 12) { bad = true; break; }
        if (memcmp(type, "IDAT", 4) == 0) {
            if (n_idat == 0) { one = buf + pos + 8; one_len = len; }
            else {
                if (n_idat == 1) idat.assign(one, one + one_len);
                idat.insert(idat.end(), buf + pos + 8, buf + pos + 8 + len);
            }
            ++n_idat;
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (bad || n_idat == 0) return 3;
    const unsigned char *zsrc = n_idat == 1 ? one : idat.data();
    const size_t zlen = n_idat == 1 ? one_len : idat.size();
    static const bool zlib_only = getenv("MSPA_INGEST_ZLIB") != nullptr;      // A/B switch: skip the table-driven decoder
    if (zlib_only || !inflate_zlib(zsrc, zlen, raw.data(), raw.size())) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) return 3;
        zs.next_out = raw.data();
        zs.avail_out = (uInt)raw.size();
        zs.next_in = const_cast<Bytef *>(zsrc);
        zs.avail_in = (uInt)zlen;
        const int rc = inflate(&zs, Z_FINISH);
        const bool full = zs.avail_out == 0;
        inflateEnd(&zs);
        // Z_STREAM_END only: the stream ended where the scanlines end AND zlib checked its Adler-32 trailer.  Z_OK /
        // Z_BUF_ERROR with a full output is a stream that is longer than h * (2 w + 1) bytes or was cut before its trailer --
        // libpng (the reference's cv2.imread) rejects both, so does this reader (status 3).
        if (!full || rc != Z_STREAM_END) return 3;
    }
    // undo the row filters in place (bytes per pixel = 2), then swap to host order
    const unsigned char *prior = nullptr;
    for (int32_t y = 0; y < h; ++y) {
        unsigned char *row = raw.data() + stride * (size_t)y;
        const int ft = row[0];
        unsigned char *x = row + 1;
        const size_t nb = (size_t)w * 2;
        switch (ft) {
        case 0: break;
        case 1:
            for (size_t i = 2; i < nb; ++i) x[i] = (unsigned char)(x[i] + x[i - 2]);
            break;
        case 2:
            if (prior) for (size_t i = 0; i < nb; ++i) x[i] = (unsigned char)(x[i] + prior[i]);
            break;
        case 3:
            for (size_t i = 0; i < nb; ++i) {
                const int a = i >= 2 ? x[i - 2] : 0, b = prior ? prior[i] : 0;
                x[i] = (unsigned char)(x[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < nb; ++i) {
                const int a = i >= 2 ? x[i - 2] : 0, b = prior ? prior[i] : 0, c = (prior && i >= 2) ? prior[i - 2] : 0;
                x[i] = (unsigned char)(x[i] + paeth(a, b, c));
            }
            break;
        default: return 3;
        }
        uint16_t *out = dst + (size_t)y * (size_t)w;
        for (int32_t i = 0; i < w; ++i) out[i] = (uint16_t)(((uint16_t)x[2 * i] << 8) | (uint16_t)x[2 * i + 1]);
        prior = x;
    }
    return 0;
}

}  // namespace
}  // namespace mspa

using namespace mspa;

extern "C" int mspa_png_header_host(const char *path_host, int32_t *h, int32_t *w, int32_t *bit_depth, int32_t *color_type,
                                    int32_t *interlace) {
    if (!path_host) return fail(MSPA_EINVAL, "mspa_png_header_host: null path");
    FILE *f = fopen(path_host, "rb");
    if (!f) return fail(MSPA_EINVAL, std::string("mspa_png_header_host: cannot open ") + path_host);
    unsigned char buf[33];
    const size_t got = fread(buf, 1, sizeof buf, f);
    fclose(f);
    PngHeader hd;
    if (parse_header(buf, got, hd)) return fail(MSPA_EINVAL, std::string("mspa_png_header_host: not a PNG file: ") + path_host);
    if (hd.w > 0x7fffffffu || hd.h > 0x7fffffffu) return fail(MSPA_EINVAL, "mspa_png_header_host: image too large");
    if (h) *h = (int32_t)hd.h;
    if (w) *w = (int32_t)hd.w;
    if (bit_depth) *bit_depth = hd.bit_depth;
    if (color_type) *color_type = hd.color_type;
    if (interlace) *interlace = hd.interlace;
    return MSPA_OK;
}

extern "C" int mspa_inflate_zlib_fast_host(const void *src_host, int64_t src_bytes, void *dst_host, int64_t dst_bytes) {
    if (!src_host || !dst_host || src_bytes <= 0 || dst_bytes < 0) return fail(MSPA_EINVAL, "mspa_inflate_zlib_fast_host: bad argument");
    return inflate_zlib((const uint8_t *)src_host, (size_t)src_bytes, (uint8_t *)dst_host, (size_t)dst_bytes) ? 0 : 1;
}

extern "C" int mspa_read_depth_png_host(const char *const *paths_host, int64_t n_files, int32_t h, int32_t w,
                                        uint16_t *dst_host, int32_t n_threads, int32_t *status_host) {
    if (n_files < 0 || h <= 0 || w <= 0 || (n_files > 0 && (!paths_host || !dst_host || !status_host)))
        return fail(MSPA_EINVAL, "mspa_read_depth_png_host: bad argument");
    for (int64_t k = 0; k < n_files; ++k)
        if (!paths_host[k]) return fail(MSPA_EINVAL, "mspa_read_depth_png_host: null path");
    if (n_files == 0) return MSPA_OK;
    const int64_t nt = n_threads < 1 ? 1 : (n_threads > n_files ? n_files : (int64_t)n_threads);
    const size_t frame = (size_t)h * (size_t)w;
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        // scratch of the THREAD, not of the call: the pool's workers keep it (and its faulted-in pages) from scene to scene
        static thread_local std::vector<unsigned char> file, raw, idat;
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_files) return;
            int st;
            try {
                st = read_file(paths_host[k], file) ? decode_gray16(file.data(), file.size(), h, w, dst_host + (size_t)k * frame, raw, idat)
                                                    : 1;
            } catch (...) {                           // allocation failure on a damaged length field
                st = 3;
            }
            status_host[k] = st;
        }
    };
    HostPool::get().parallel((int)nt, work);                // persistent worker threads (host_pool.h); the caller works too
    return MSPA_OK;
}


// The host half of the on-device decode (csrc/device_ingest.hip): file bytes -> the scanlines' zlib stream, packed for ONE H2D copy.
extern "C" int mspa_png_pack_idat_host(const char *const *paths_host, int64_t n_files, int32_t h, int32_t w, void *dst_host,
                                       int64_t dst_capacity, int64_t *offsets_host, int64_t *bytes_host, int32_t *status_host,
                                       int64_t *capacity_needed, int32_t n_threads) {
    if (n_files < 0 || h <= 0 || w <= 0 || (n_files > 0 && (!paths_host || !offsets_host || !bytes_host || !status_host)))
        return fail(MSPA_EINVAL, "mspa_png_pack_idat_host: bad argument");
    for (int64_t k = 0; k < n_files; ++k)
        if (!paths_host[k]) return fail(MSPA_EINVAL, "mspa_png_pack_idat_host: null path");
    // slot k = the file's size rounded up to 16 bytes (+ 16: the device reader may touch the 8-byte unit a stream ends in); the
    // file is read INTO its slot and the IDAT payloads are then moved to the slot's front, so nothing is copied twice
    int64_t need = 0;
    std::vector<int64_t> fsize((size_t)n_files, -1);
    for (int64_t k = 0; k < n_files; ++k) {
        struct stat st;
        offsets_host[k] = need;
        if (stat(paths_host[k], &st) == 0 && S_ISREG(st.st_mode)) {
            fsize[(size_t)k] = (int64_t)st.st_size;
            need += (((int64_t)st.st_size + 15) & ~(int64_t)15) + 16;
        } else {
            need += 16;
        }
    }
    if (capacity_needed) *capacity_needed = need;
    if (!dst_host) return MSPA_OK;
    if (dst_capacity < need) return fail(MSPA_EINVAL, "mspa_png_pack_idat_host: destination smaller than capacity_needed");
    if (n_files == 0) return MSPA_OK;
    const int64_t nt = n_threads < 1 ? 1 : (n_threads > n_files ? n_files : (int64_t)n_threads);
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t k = next.fetch_add(1);
            if (k >= n_files) return;
            bytes_host[k] = 0;
            unsigned char *slot = (unsigned char *)dst_host + offsets_host[k];
            const int64_t sz = fsize[(size_t)k];
            int st = 1;
            if (sz >= 0) {
                FILE *f = fopen(paths_host[k], "rb");
                if (f) {
                    const bool got = sz == 0 || fread(slot, 1, (size_t)sz, f) == (size_t)sz;
                    fclose(f);
                    if (got) {
                        PngHeader hd;
                        if (parse_header(slot, (size_t)sz, hd)) st = 3;
                        else if (hd.bit_depth != 16 || hd.color_type != 0 || hd.interlace != 0 || hd.w != (uint32_t)w || hd.h != (uint32_t)h) st = 2;
                        else {
                            size_t pos = 33, out = 0;
                            bool bad = false, any = false;
                            while (pos + 12 <= (size_t)sz) {
                                const uint32_t len = be32(slot + pos);
                                const unsigned char *type = slot + pos + 4;
                                if ((size_t)len > (size_t)sz - pos - 12) { bad = true; break; }
                                if (memcmp(type, "IDAT", 4) == 0) {
                                    memmove(slot + out, slot + pos + 8, len);      // always towards the front: out < pos + 8
                                    out += len;
                                    any = true;
                                } else if (memcmp(type, "IEND", 4) == 0) {
                                    break;
                                }
                                pos += 12 + (size_t)len;
                            }
                            if (bad || !any) st = 3;
                            else {
                                st = 0;
                                bytes_host[k] = (int64_t)out;
                            }
                        }
                    }
                }
            }
            status_host[k] = st;
        }
    };
    HostPool::get().parallel((int)nt, work);
    return MSPA_OK;
}
