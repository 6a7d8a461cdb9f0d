// Micro-benchmark: issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU
// instructions the geometry kernels use.  Each test runs N independent chains x UNROLL so the
// result is throughput-bound, not latency-bound.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
#define KERNEL(name, decl, body, post)                                                   \
    __global__ void name(double *out, long long *cyc, double seed) {                      \
        decl;                                                                             \
        long long t0 = __builtin_readcyclecounter();                                      \
        for (int i = 0; i < REP; ++i) { body; }                                           \
        long long t1 = __builtin_readcyclecounter();                                      \
        post;                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                  \
    }
#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; double b = seed * 0.5 + threadIdx.x
#define X8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)

#define FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(b));
#define ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define RND(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define MAXF(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define CMP(x) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");
#define CMPS(x) asm volatile("v_cmp_lt_f64 s[20:21], %0, %1" : : "v"(x), "v"(b) : "s20", "s21");
#define I8 int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7; int jj = 3
#define F8 float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; float gg = 0.5f
#define L8 unsigned long long l0 = threadIdx.x, l1 = l0 + 1, l2 = l0 + 2, l3 = l0 + 3, l4 = l0 + 4, l5 = l0 + 5, l6 = l0 + 6, l7 = l0 + 7, mm = 5
#define XI8(OP) OP(i0) OP(i1) OP(i2) OP(i3) OP(i4) OP(i5) OP(i6) OP(i7)
#define XF8(OP) OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7)
#define XL8(OP) OP(l0) OP(l1) OP(l2) OP(l3) OP(l4) OP(l5) OP(l6) OP(l7)
#define ISUM a0 += (double)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
#define CVTI2(d, x) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(d) : "v"(x));
#define CVTD2(x, s) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(x) : "v"(s));
#define ADD32(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(jj));
#define CND(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(jj) : "vcc");
#define CNDS(x) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(x) : "v"(jj) : "s20", "s21");
#define FMA32(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(gg));
#define LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(x) : "v"(mm));
#define MED3(x) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(jj));
#define LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x) : "v"(jj));
#define CMPU(x) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(x), "v"(jj) : "vcc");
#define BCNT(x) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(x) : "v"(jj));
#define AND32(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(jj));
#define BITOP3(x) asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(x) : "v"(jj));
#define MULLO(x) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(jj));
#define MAD24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(jj));
#define WRLANE(x) asm volatile("v_writelane_b32 %0, s20, 5" : "+v"(x) : : "s20");
#define DIVFIX(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));

KERNEL(k_fma, D8, X8(FMA), (void)0)
KERNEL(k_add, D8, X8(ADD), (void)0)
KERNEL(k_mul, D8, X8(MUL), (void)0)
KERNEL(k_rnd, D8, X8(RND), (void)0)
KERNEL(k_rcp, D8, X8(RCP), (void)0)
KERNEL(k_max, D8, X8(MAXF), (void)0)
KERNEL(k_cmp_vcc, D8, X8(CMP), (void)0)
KERNEL(k_cmp_sgpr, D8, X8(CMPS), (void)0)
KERNEL(k_cvt_i32, D8; I8, CVTI2(i0, a0) CVTI2(i1, a1) CVTI2(i2, a2) CVTI2(i3, a3) CVTI2(i4, a4) CVTI2(i5, a5) CVTI2(i6, a6) CVTI2(i7, a7), ISUM)
KERNEL(k_cvt_f64, D8; I8, CVTD2(a0, i0) CVTD2(a1, i1) CVTD2(a2, i2) CVTD2(a3, i3) CVTD2(a4, i4) CVTD2(a5, i5) CVTD2(a6, i6) CVTD2(a7, i7), (void)0)
KERNEL(k_add_u32, D8; I8, XI8(ADD32), ISUM)
KERNEL(k_cndmask, D8; I8, XI8(CND), ISUM)
KERNEL(k_cndmask_s, D8; I8, XI8(CNDS), ISUM)
KERNEL(k_fma_f32, D8; F8, XF8(FMA32), a0 += f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;)
KERNEL(k_lshl_add_u64, D8; L8, XL8(LSHLADD64), a0 += (double)(l0 + l1 + l2 + l3 + l4 + l5 + l6 + l7);)
KERNEL(k_med3, D8; I8, XI8(MED3), ISUM)
KERNEL(k_lshl_or, D8; I8, XI8(LSHLOR), ISUM)
KERNEL(k_cmp_u32, D8; I8, XI8(CMPU), ISUM)
KERNEL(k_bcnt, D8; I8, XI8(BCNT), ISUM)
KERNEL(k_and32, D8; I8, XI8(AND32), ISUM)
KERNEL(k_bitop3, D8; I8, XI8(BITOP3), ISUM)
KERNEL(k_mullo, D8; I8, XI8(MULLO), ISUM)
KERNEL(k_mad24, D8; I8, XI8(MAD24), ISUM)
KERNEL(k_wrlane, D8; I8, XI8(WRLANE), ISUM)
KERNEL(k_rcp_f32, D8; F8, XF8(DIVFIX), a0 += f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;)

template <typename F>
void run(const char *name, F kern, int waves_per_simd) {
    const int blocks = 256 * 4, threads = 64 * waves_per_simd;   // 4 blocks per CU -> one block per SIMD (roughly)
    double *out;
    long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.25);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.25);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    // readcyclecounter on gfx9 = s_memtime (shader clock); instructions per wave = REP*8
    printf("%-16s waves/block %d : %7.2f cycles per wave-instruction per wave (block-local)\n", name, waves_per_simd,
           avg / (REP * 8.0));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {1, 4, 8}) {
        run("v_fma_f64", k_fma, w);
        run("v_add_f64", k_add, w);
        run("v_mul_f64", k_mul, w);
        run("v_rndne_f64", k_rnd, w);
        run("v_rcp_f64", k_rcp, w);
        run("v_max_f64", k_max, w);
        run("v_cmp_f64->vcc", k_cmp_vcc, w);
        run("v_cmp_f64->sgpr", k_cmp_sgpr, w);
        run("v_cvt_i32_f64", k_cvt_i32, w);
        run("v_cvt_f64_u32", k_cvt_f64, w);
        run("v_add_u32", k_add_u32, w);
        run("v_cndmask_b32", k_cndmask, w);
        run("v_cndmask sgpr", k_cndmask_s, w);
        run("v_lshl_or_b32", k_lshl_or, w);
        run("v_cmp_ne_u32", k_cmp_u32, w);
        run("v_fma_f32", k_fma_f32, w);
        run("v_lshl_add_u64", k_lshl_add_u64, w);
        run("v_med3_i32", k_med3, w);
        run("v_bcnt_u32_b32", k_bcnt, w);
        run("v_and_b32", k_and32, w);
        run("v_bitop3_b32", k_bitop3, w);
        run("v_mul_lo_u32", k_mullo, w);
        run("v_mad_u32_u24", k_mad24, w);
        run("v_writelane_b32", k_wrlane, w);
        run("v_rcp_f32", k_rcp_f32, w);
    }
    return 0;
}
