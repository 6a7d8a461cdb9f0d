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

// ---------------------------------------------------------------------------------------------------------------------
// Guard band of the composed ("fast") kernels, as a BOUND.  The fast kernels evaluate K2 inv(A E2) A E1 inv(K1) p through one
// composed 3x4 matrix; the reference multiplies the five matrices one after the other.  Both are float64 evaluations of the
// same real number, and both stay within c 2^-53 (|K2| |inv(A E2)| |A| |E1| |inv(K1)| |p|) of it (c counts the roundings of the
// longer chain; MSPA_GUARD_C in include/mspa.h).  Slot MSPA_MAT_BOUNDS of the two frame records carries that product of
// magnitudes in factored form; for a block of pixels (columns <= xmax, rows <= ymax, depth samples <= dmax millimetres) it
// gives B_k >= |q_k(fast) - q_k(reference)| for the homogeneous image coordinates q (pixel * millimetre; k = 2: the camera-2
// depth in millimetres).  From B:
//   zmin   a lane with |q_2| <= zmin is never trusted: above it |u(fast) - u(reference)| <= (B_0 + |u| B_2) / q_2 <= guard / 2
//          for every |u| <= max(W, H) + 1, so rounding ties / integer bounds within the pixel guard catch every other flip;
//   zsafe  a FAST-evaluated q_2 above zsafe means every evaluation order is above zmin (and positive);
//   gz     depth-test guard: |q_2 - sample| <= gz may flip the strict comparison.
// (Sums stand in for maxima -- B_0 + B_1 >= max(B_0, B_1) -- which only widens the band: v_max_f64 drags a canonicalising
// instruction per operand along.)  The frustum culling keeps its constant margins (1 pixel * mm, 1e-3 mm) and checks, only
// when it is about to cull, that they are at least four times the evaluation error anywhere in the image (cull_margins_hold).
// Round 3 used constants (1e-6 px, 1e-6 mm) for all of these: wrong when camera 2 sits within micrometres of a frame-1
// surface point (the error grows like 1 / q_2) or when world coordinates are huge (tools/guard_bound_emulation.py).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef MSPA_GUARD_PX
#define MSPA_GUARD_PX 1e-6
#endif
constexpr double kGuardPx = MSPA_GUARD_PX;       // distance to a rounding tie / an integer bound (pixels); zmin scales with 1 / it
constexpr double kGuardZmmFloor = 1e-6;          // smallest depth-test guard (mm): covers the rounding of sample * 0.001 itself
constexpr double kCullMarginXY = 1.0;            // frustum culling: homogeneous x / y (pixel * millimetre)
constexpr double kCullMarginZ = 1e-3;            // frustum culling: homogeneous depth (millimetres)

struct Guard {
    double zmin, zsafe, gz;
};

// b1 = slot MSPA_MAT_BOUNDS of frame 1, b2 = of frame 2 (wave-uniform addresses: scalar loads); all arguments wave-uniform.
__device__ __forceinline__ Guard guard_from_bounds(const double *__restrict__ b1, const double *__restrict__ b2, double xmax,
                                                   double ymax, double dmax_mm, double wh_max) {
    const double w = __builtin_fma(__builtin_fma(b1[0], xmax, __builtin_fma(b1[1], ymax, b1[2])), dmax_mm, b1[3]);
    const double B0 = __builtin_fma(b2[4], w, b2[8]);
    const double B1 = __builtin_fma(b2[5], w, b2[9]);
    const double B2 = __builtin_fma(b2[6], w, b2[10]);
    Guard g;
    // A record whose bound slot was never filled (zeros: a caller that skipped mspa_frame_bounds_host) or holds NaN must not
    // yield a thin band: then nothing is trusted -- every lane takes the exact chain, results stay right, only speed is lost.
    const bool sane = (B0 + B1 + B2) > 0.0;                   // false for zeros and for NaN (wave-uniform: a scalar select)
    g.zmin = sane ? __builtin_fma(wh_max + 1.0, B2, B0 + B1) * (2.0 / kGuardPx) : __builtin_inf();
    g.zsafe = sane ? __builtin_fma(2.0, B2, g.zmin) : __builtin_inf();
    g.gz = sane ? __builtin_fma(2.0, B2, kGuardZmmFloor) : __builtin_inf();
    return g;
}

// The culling margins against the pair's bound over the whole image (columns <= xmax, rows <= ymax) and the full sample range.
// NaN coefficients compare false and an unfilled (all-zero) slot is refused: nothing is culled then.
__device__ __forceinline__ bool cull_margins_hold(const double *__restrict__ b1, const double *__restrict__ b2, double xmax,
                                                  double ymax, double wh_max) {
    const double w = __builtin_fma(__builtin_fma(b1[0], xmax, __builtin_fma(b1[1], ymax, b1[2])), 65535.0, b1[3]);
    const double B0 = __builtin_fma(b2[4], w, b2[8]);
    const double B1 = __builtin_fma(b2[5], w, b2[9]);
    const double B2 = __builtin_fma(b2[6], w, b2[10]);
    return ((B0 + B1 + B2) > 0.0) & (__builtin_fma(wh_max, B2, B0 + B1) < 0.25 * kCullMarginXY) & (B2 < 0.25 * kCullMarginZ);
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
