// Accuracy of v_rcp_f64 on gfx950, raw and after the one Newton step K3 / K1 apply: maximum error against 1 / x (correctly rounded on
// the host) over log-uniform random x, in units of 2^-53 relative.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/rcp_accuracy.hip -o /tmp/rcp_accuracy && /tmp/rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const double *x, double *r0, double *r1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    r0[i] = r;
    r1[i] = __builtin_fma(__builtin_fma(-v, r, 1.0), r, r);
}

int main() {
    const int n = 1 << 22;
    std::vector<double> x(n), r0(n), r1(n);
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> e(-20.0, 20.0), m(1.0, 2.0);
    for (int i = 0; i < n; ++i) x[i] = std::ldexp(m(g), (int)e(g)) * ((i & 1) ? -1.0 : 1.0);
    double *dx, *d0, *d1;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&d0, n * 8); (void)hipMalloc(&d1, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
    (void)hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    double w0 = 0, w1 = 0;
    long exact1 = 0;
    for (int i = 0; i < n; ++i) {
        // error of r against the real 1/x: (1 - x r) / (x r) evaluated with an FMA residual (exact to first order)
        const double res0 = std::fma(-x[i], r0[i], 1.0), res1 = std::fma(-x[i], r1[i], 1.0);
        w0 = std::fmax(w0, std::fabs(res0));
        w1 = std::fmax(w1, std::fabs(res1));
        exact1 += r1[i] == 1.0 / x[i];
    }
    printf("# v_rcp_f64 on gfx950, %d log-uniform samples in 2^-20 .. 2^20, both signs\n", n);
    printf("raw:              max |1 - x r| = %.3e = %.1f x 2^-53  (2^%.1f)\n", w0, w0 / 1.1102230246251565e-16, std::log2(w0));
    printf("one Newton step:  max |1 - x r| = %.3e = %.2f x 2^-53; equal to the correctly rounded 1/x in %.2f %% of the samples\n", w1,
           w1 / 1.1102230246251565e-16, 100.0 * exact1 / n);
    return 0;
}
