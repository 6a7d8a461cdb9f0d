// Shared device helpers for libmspa.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/mspa.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libmspa is written for gfx950 (MI355X) only: 16-byte LDS-DMA, v_mfma_i32_32x32x32_i8, native f64 min/max atomics"
#endif

namespace mspa {

constexpr int kWave = 64;

// thread-local error text behind mspa_last_error_string()
std::string &last_error();
int fail(int code, const std::string &what);
int check_hip(hipError_t e, const char *what);
int xcd_count();               // XCDs a 1-D grid's workgroups are dealt over: 8 (MI355X, SPX mode) or 1

// Lanes of one wave exchange data through LDS (every lane stores its slot, then loads another lane's).  The hardware executes a
// wave's LDS operations in order, so no instruction is needed -- but the compiler must not move the loads above the stores
// either, and under the HIP memory model nothing but a fence says so: a wavefront-scope release / acquire pair plus a wave
// barrier (none of the three emits an instruction on gfx950).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One row of an affine 4x4 (row-major, uniform address -> scalar loads) applied to (x, y, z, 1),
// in the order NumPy/OpenBLAS evaluates a K=4 product: m0*x, fma(m1,y,.), fma(m2,z,.), then +m3
// (fma(m3, 1.0, acc) and acc + m3 round identically).  Compiled with -ffp-contract=off: nothing
// here may be fused or split by the compiler.
__device__ __forceinline__ double affine_row(const double *__restrict__ m, double x, double y, double z) {
    double acc = m[0] * x;
    acc = __builtin_fma(m[1], y, acc);
    acc = __builtin_fma(m[2], z, acc);
    return acc + m[3];
}

// The same row applied to a general homogeneous point (x, y, z, w): the K = 4 chain ends with fma(m3, w, .) (IH:57-66 take
// any [N, 4] input; with w == 1 this rounds exactly like affine_row).
__device__ __forceinline__ double affine_row_w(const double *__restrict__ m, double x, double y, double z, double w) {
    double acc = m[0] * x;
    acc = __builtin_fma(m[1], y, acc);
    acc = __builtin_fma(m[2], z, acc);
    return __builtin_fma(m[3], w, acc);
}

// np.round(v).astype(int) then np.clip(.., 0, hi) (IH:362-366, OPS:285-290): half-to-even, the
// x86-64 float64->int64 conversion (NaN / out of range -> INT64_MIN), clip.  hi < 32768.
__device__ __forceinline__ int round_clip(double v, int hi) {
    double r = __builtin_rint(v);
    // r in [0, hi] -> itself; r > hi and r < 2^63 -> hi; everything else (negative, NaN, >= 2^63) -> 0
    int idx = 0;
    if (r >= 0.0) {
        if (r <= (double)hi) idx = (int)r;
        else if (r < 9223372036854775808.0) idx = hi;
    }
    return idx;
}

// IH:337-386 for one projected point; the depth gather is skipped for lanes that cannot pass.
// `inview` reports the two tests that do not involve the depth buffer (inside the image, in front of
// the camera): it is the condition under which (xi, yi) is a meaningful correspondence.
__device__ __forceinline__ bool depth_test(bool enable, double u, double v, double d,
                                           const uint16_t *__restrict__ depth_img, int dh, int dw, int H, int W,
                                           double sx, double sy, int &xi, int &yi, bool *inview = nullptr,
                                           double depth_value_scale = 0.001) {
    bool inb = (u >= 0.0) && (u < (double)W) && (v >= 0.0) && (v < (double)H);
    xi = round_clip(u * sx, dw - 1);
    yi = round_clip(v * sy, dh - 1);
    bool vis = false;
    if (inview) *inview = enable && inb && d > 0.0;
    if (enable && inb && d > 0.0) {
        double dv = (double)depth_img[yi * dw + xi] * depth_value_scale;      // IH:368
        vis = d < dv;
    }
    return vis;
}

}  // namespace mspa
