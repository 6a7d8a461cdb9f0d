// Library-level entry points of libmspa.so: version, error string, device facts.
#include "mspa_common.h"

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
