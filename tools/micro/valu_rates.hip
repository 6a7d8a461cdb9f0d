// Issue rate of the VALU instructions K3's row loop is made of, measured: one kernel per instruction, every wave runs
// ITER x 32 independent copies of it (8 register sets, no dependency chain shorter than 8 issues), 8 waves per SIMD on all
// CUs; cycles per instruction and SIMD = elapsed x clock / (instructions per SIMD).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
constexpr int ITER = 2048;

#define KERNEL(NAME, ASM)                                                                                   \
    __global__ void __launch_bounds__(256) k_##NAME(double *out, double seed, unsigned iseed) {              \
        double d[8], e[8];                                                                                    \
        unsigned u[8], w[8];                                                                                  \
        for (int i = 0; i < 8; ++i) {                                                                         \
            d[i] = seed + i + threadIdx.x * 1e-3;                                                             \
            e[i] = seed * 0.5 + i;                                                                            \
            u[i] = iseed + i + threadIdx.x;                                                                   \
            w[i] = iseed * 3 + i;                                                                             \
        }                                                                                                     \
        unsigned long long m = 0x5555555555555555ull + iseed;                                                 \
        for (int it = 0; it < ITER; ++it) {                                                                   \
            REP32(ASM)                                                                                        \
        }                                                                                                     \
        double s = 0;                                                                                         \
        for (int i = 0; i < 8; ++i) s += d[i] + e[i] + u[i] + w[i];                                           \
        if (s == 1234.5678) out[threadIdx.x] = s + (double)m;                                                 \
    }

#define A_ADD(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
#define A_FMA(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e[i]));
#define A_MUL(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
#define A_RCP(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define A_FRACT(i) asm volatile("v_fract_f64 %0, %0" : "+v"(d[i]));
#define A_RNDNE(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(d[i]));
#define A_MAX(i) asm volatile("v_max_f64 %0, |%0|, |%1|" : "+v"(d[i]) : "v"(e[i]));
#define A_CVTFU(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
#define A_CVTIF(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
#define A_CMP(i) asm volatile("v_cmp_lt_f64 %0, |%1|, %2" : "=s"(m) : "v"(d[i]), "v"(e[i]));
#define A_CMPU(i) asm volatile("v_cmp_ne_u32 %0, 0, %1" : "=s"(m) : "v"(u[i]));
#define A_PK(i) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_DOT(i) asm volatile("v_dot2_u32_u16 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(w[i]), "s"(m));
#define A_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MED3(i) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_WRLANE(i) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(u[i]) : "s"((unsigned)m));
#define A_MBCNT(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(u[i]) : "s"((unsigned)m));
#define A_PKMIN(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(w[i]));
#define A_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(u[i]));
#define A_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e[i]));
#define A_LDEXP(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(u[i]));

#define A_CNDVCC(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(w[i]) : "vcc");
#define A_CMPVCC(i) asm volatile("v_cmp_ne_u32 vcc, 0, %0" : : "v"(u[i]) : "vcc");
#define A_CMPF64VCC(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(e[i]) : "vcc");
#define A_CMPF32VCC(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(u[i]), "v"(w[i]) : "vcc");
#define A_MINI(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
#define A_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i]));
#define A_ADDF32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MULF32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MAXF32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_CVTF32F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
#define A_CVTF64F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
#define A_CVTF32U32(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(u[i]));
#define A_CVTI32F32(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(u[i]));
#define A_RNDNE32(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(u[i]));
#define A_FRACT32(i) asm volatile("v_fract_f32 %0, %0" : "+v"(u[i]));
#define A_ADDCO(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(u[i]) : "v"(w[i]) : "vcc");
#define A_MBCNTHI(i) asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(u[i]) : "s"((unsigned)m));
#define A_READLANE(i) { unsigned sr; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sr) : "v"(u[i])); m += sr; }
#define A_MADF32(i) asm volatile("v_mad_f32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_FMAC32(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_FMAC64(i) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(d[i]) : "v"(e[i]));
#define A_PKADD32(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
#define A_PKMUL32(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
#define A_SUBU(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_MAXU(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(u[i]) : "v"(w[i]));
#define A_ADD64SGPR(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "s"(seed));
KERNEL(cnd_vcc, A_CNDVCC) KERNEL(cmp_ne_u32_vcc, A_CMPVCC) KERNEL(cmp_lt_f64_vcc, A_CMPF64VCC) KERNEL(cmp_lt_f32_vcc, A_CMPF32VCC)
KERNEL(min_i32, A_MINI) KERNEL(lshlrev_b32, A_LSHL) KERNEL(and_b32, A_AND) KERNEL(mul_u32_u24, A_MUL24) KERNEL(lshl_add_u32, A_LSHLADD)
KERNEL(add3_u32, A_ADD3) KERNEL(lshl_or_b32, A_LSHLOR) KERNEL(perm_b32, A_PERM) KERNEL(bfe_u32, A_BFE) KERNEL(add_f32, A_ADDF32)
KERNEL(mul_f32, A_MULF32) KERNEL(max_f32, A_MAXF32) KERNEL(cvt_f32_f64, A_CVTF32F64) KERNEL(cvt_f64_f32, A_CVTF64F32)
KERNEL(cvt_f32_u32, A_CVTF32U32) KERNEL(cvt_i32_f32, A_CVTI32F32) KERNEL(rndne_f32, A_RNDNE32) KERNEL(fract_f32, A_FRACT32)
KERNEL(addc_co_u32, A_ADDCO) KERNEL(mbcnt_hi, A_MBCNTHI) KERNEL(readlane_b32, A_READLANE)
KERNEL(fmac_f32, A_FMAC32) KERNEL(fmac_f64, A_FMAC64) KERNEL(pk_add_f32, A_PKADD32) KERNEL(pk_mul_f32, A_PKMUL32)
KERNEL(sub_u32, A_SUBU) KERNEL(max_u32, A_MAXU) KERNEL(add_f64_sgpr, A_ADD64SGPR)
KERNEL(add_f64, A_ADD) KERNEL(fma_f64, A_FMA) KERNEL(mul_f64, A_MUL) KERNEL(rcp_f64, A_RCP) KERNEL(fract_f64, A_FRACT)
KERNEL(rndne_f64, A_RNDNE) KERNEL(max_f64_abs, A_MAX) KERNEL(cvt_f64_u32, A_CVTFU) KERNEL(cvt_i32_f64, A_CVTIF)
KERNEL(cmp_lt_f64, A_CMP) KERNEL(cmp_ne_u32, A_CMPU) KERNEL(cvt_pk_i16_i32, A_PK) KERNEL(dot2_u32_u16, A_DOT)
KERNEL(cndmask_b32, A_CND) KERNEL(add_u32, A_ADDU) KERNEL(mad_u32_u24, A_MAD24) KERNEL(mul_lo_u32, A_MULLO)
KERNEL(med3_i32, A_MED3) KERNEL(writelane_b32, A_WRLANE) KERNEL(mbcnt_lo, A_MBCNT) KERNEL(pk_min_u16, A_PKMIN)
KERNEL(mov_b32, A_MOV) KERNEL(fma_f32, A_FMA32) KERNEL(rcp_f32, A_RCP32) KERNEL(pk_fma_f32, A_PKFMA) KERNEL(ldexp_f64, A_LDEXP)

struct Entry { const char *name; void (*fn)(double *, double, unsigned); };
#define E(NAME) {#NAME, k_##NAME}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int clock_khz = 0;
    hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeClockRate, 0);
    double *out;
    hipMalloc(&out, 4096);
    std::vector<Entry> es = {E(add_f64), E(fma_f64), E(mul_f64), E(rcp_f64), E(fract_f64), E(rndne_f64), E(max_f64_abs), E(cvt_f64_u32),
                             E(cvt_i32_f64), E(cmp_lt_f64), E(cmp_ne_u32), E(cvt_pk_i16_i32), E(dot2_u32_u16), E(cndmask_b32), E(add_u32),
                             E(mad_u32_u24), E(mul_lo_u32), E(med3_i32), E(writelane_b32), E(mbcnt_lo), E(pk_min_u16), E(mov_b32),
                             E(fma_f32), E(rcp_f32), E(pk_fma_f32), E(ldexp_f64), E(add_f64), E(cnd_vcc), E(cmp_ne_u32_vcc), E(cmp_lt_f64_vcc),
                             E(cmp_lt_f32_vcc), E(min_i32), E(lshlrev_b32), E(and_b32), E(mul_u32_u24), E(lshl_add_u32), E(add3_u32),
                             E(lshl_or_b32), E(perm_b32), E(bfe_u32), E(add_f32), E(mul_f32), E(max_f32), E(cvt_f32_f64), E(cvt_f64_f32),
                             E(cvt_f32_u32), E(cvt_i32_f32), E(rndne_f32), E(fract_f32), E(addc_co_u32), E(mbcnt_hi), E(readlane_b32),
                             E(fmac_f32), E(fmac_f64), E(pk_add_f32), E(pk_mul_f32), E(sub_u32), E(max_u32), E(add_f64_sgpr)};
    const int blocks = cus * 8;          // 8 blocks of 4 waves per CU = 8 waves per SIMD
    printf("# %s, %d CUs, nominal clock %d MHz; %d waves per SIMD, %d instructions per wave\n", p.gcnArchName, cus, clock_khz / 1000, 8, ITER * 32);
    printf("| instruction | us | cycles per wave instruction and SIMD (at the nominal clock) | relative to v_add_f64 |\n|---|---|---|---|\n");
    double base = 0;
    for (auto &e : es) {
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5, 7u);   // spin the clock up
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(a);
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5, 7u);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        const double insts_per_simd = 8.0 * ITER * 32;
        const double cyc = best * 1e-3 * (clock_khz * 1e3) / insts_per_simd;
        if (base == 0) base = cyc;
        printf("| v_%s | %.1f | %.2f | %.2f |\n", e.name, best * 1e3, cyc, cyc / base);
    }
    return 0;
}
