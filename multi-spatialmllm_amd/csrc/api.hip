// Library-level entry points of libmspa.so: version, error string, device facts.
#include "mspa_common.h"

#include <cmath>
#include <cstring>

namespace mspa {

std::string &last_error() {
    static thread_local std::string e;
    return e;
}

int fail(int code, const std::string &what) {
    last_error() = what;
    return code;
}

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return MSPA_OK;
    last_error() = std::string(what) + ": " + hipGetErrorString(e);
    return MSPA_EHIP;
}

// Workgroups of a 1-D grid are dealt round-robin over the XCDs.  An MI355X in SPX mode exposes 256 CUs = 8 XCDs of 32; a
// partitioned device (CPX: 32 CUs) or any other part gets 1, i.e. no XCD-aware regrouping (results never depend on it).
int xcd_count() {
    static thread_local int cached = 0;
    if (cached) return cached;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    cached = (prop.multiProcessorCount == 256) ? 8 : 1;
    return cached;
}

}  // namespace mspa

using namespace mspa;

extern "C" int mspa_version(void) { return MSPA_VERSION; }

extern "C" const char *mspa_last_error_string(void) { return last_error().c_str(); }

extern "C" int mspa_device_info(int device, int *n_cu, int *wave_size, int64_t *hbm_bytes, int *clock_khz,
                                char *name_host, int name_len) {
    hipDeviceProp_t prop;
    int rc = check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (rc) return rc;
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (name_host && name_len > 0) {
        std::strncpy(name_host, prop.gcnArchName, (size_t)name_len - 1);
        name_host[name_len - 1] = 0;
    }
    return MSPA_OK;
}

// A stream whose kernels stay off `reserve_cus` of the device's compute units (every (n_cu / reserve_cus)-th one, so the reserved
// units are spread over the XCDs).  The on-device depth decode keeps ~3 600 waves resident for ~100 ms each and fills every CU's
// LDS; geometry kernels launched beside it (K1 needs 13.4 KB of LDS per workgroup) would wait milliseconds for a wave to retire.
// Decode streams made here never touch the reserved units, so those kernels always find room.
extern "C" int mspa_stream_create_reserving(int32_t reserve_cus, void **stream_out) {
    if (!stream_out) return fail(MSPA_EINVAL, "mspa_stream_create_reserving: null output");
    int dev = 0;
    hipDeviceProp_t prop;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    rc = check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    if (rc) return rc;
    const int n_cu = prop.multiProcessorCount;
    if (reserve_cus < 0 || reserve_cus >= n_cu) return fail(MSPA_EINVAL, "mspa_stream_create_reserving: reserve_cus out of range");
    hipStream_t st = nullptr;
    if (reserve_cus == 0) {
        rc = check_hip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    } else {
        const int words = (n_cu + 31) / 32;
        uint32_t mask[64] = {0};
        if (words > 64) return fail(MSPA_EUNSUPPORTED, "mspa_stream_create_reserving: more than 2 048 compute units");
        const int every = n_cu / reserve_cus;
        int reserved = 0;
        for (int cu = 0; cu < n_cu; ++cu) {
            const bool hold = (cu % every == every - 1) && reserved < reserve_cus;
            if (hold) ++reserved;
            else mask[cu >> 5] |= 1u << (cu & 31);
        }
        rc = check_hip(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask), "hipExtStreamCreateWithCUMask");
    }
    if (rc) return rc;
    *stream_out = (void *)st;
    return MSPA_OK;
}

extern "C" int mspa_stream_destroy(void *stream) {
    if (!stream) return MSPA_OK;
    return check_hip(hipStreamDestroy((hipStream_t)stream), "hipStreamDestroy");
}

// Slot MSPA_MAT_BOUNDS of the frame records (include/mspa.h): magnitudes of the two halves of the pair pipe's matrix chain,
// from which the fast kernels derive their guard band per tile.  Products of non-negative numbers: no cancellation, the few
// roundings of this function itself are covered by MSPA_GUARD_C's slack.
static void abs_mul(const double *X, const double *Y, double *out) {       // out = |X| |Y|, 4x4 row-major
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += std::fabs(X[4 * r + k]) * std::fabs(Y[4 * k + c]);
            out[4 * r + c] = acc;
        }
}

extern "C" int mspa_frame_bounds_host(double *frame_mats_host, int32_t n_frames) {
    if (n_frames < 0 || (!frame_mats_host && n_frames > 0)) return fail(MSPA_EINVAL, "mspa_frame_bounds_host: bad table");
    const double c = MSPA_GUARD_C * 0x1p-53;
    for (int32_t f = 0; f < n_frames; ++f) {
        double *rec = frame_mats_host + (int64_t)f * (MSPA_FRAME_MATS * 16);
        double AE[16], Ua[16], Na[16];
        abs_mul(rec + MSPA_MAT_A * 16, rec + MSPA_MAT_E * 16, AE);
        abs_mul(AE, rec + MSPA_MAT_KINV * 16, Ua);
        abs_mul(rec + MSPA_MAT_K * 16, rec + MSPA_MAT_EINV_ALIGNED * 16, Na);
        double *b = rec + MSPA_MAT_BOUNDS * 16;
        for (int k = 0; k < 16; ++k) b[k] = 0.0;
        for (int j = 0; j < 4; ++j) {
            const double m = std::fmax(Ua[j], std::fmax(Ua[4 + j], Ua[8 + j]));
            b[j] = j < 3 ? m : 1000.0 * m;
        }
        for (int k = 0; k < 3; ++k) {
            b[4 + k] = c * ((Na[4 * k + 0] + Na[4 * k + 1]) + Na[4 * k + 2]);
            b[8 + k] = c * 1000.0 * Na[4 * k + 3];
        }
    }
    return MSPA_OK;
}

extern "C" int mspa_camera_bounds_host(double *cam_mats_host, int32_t n_images) {
    if (n_images < 0 || (!cam_mats_host && n_images > 0)) return fail(MSPA_EINVAL, "mspa_camera_bounds_host: bad table");
    const double c = MSPA_GUARD_C * 0x1p-53 * 1000.0;
    for (int32_t i = 0; i < n_images; ++i) {
        double *rec = cam_mats_host + (int64_t)i * (MSPA_CAM_MATS * 16);
        double Na[16];
        abs_mul(rec + MSPA_CAM_K * 16, rec + MSPA_CAM_EINV * 16, Na);
        double *b = rec + MSPA_CAM_BOUNDS * 16;
        for (int k = 0; k < 16; ++k) b[k] = 0.0;
        double nr[3];
        for (int k = 0; k < 3; ++k) nr[k] = (Na[4 * k + 0] + Na[4 * k + 1]) + Na[4 * k + 2];
        b[0] = c * (nr[0] + nr[1]);
        b[1] = c * nr[2];
        b[2] = c * (Na[3] + Na[7]);
        b[3] = c * Na[11];
    }
    return MSPA_OK;
}
