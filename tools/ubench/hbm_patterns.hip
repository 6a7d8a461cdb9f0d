// Calibration of rocprofv3's FETCH_SIZE for the load shapes the K3 kernel uses: each kernel reads a known number of
// bytes once (buffers are larger than L2 + Infinity Cache and touched in a streaming order).  Run under
// `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (tools/calibrate.sh) and compare reported KiB with the bytes below.
//   read_b128   16 B per lane, coalesced   (1 KiB per wave instruction)
//   read_b32     4 B per lane, coalesced   (256 B per wave instruction)
//   read_lds_b32 4 B per lane through LDS-DMA (global_load_lds), the depth-1 tile path of K3
//   gather_b16   2 B per lane, lanes spread over a 600 KB window (the depth-2 gather of K3): counts 64-B sectors
// Build: hipcc --offload-arch=gfx950 -O2 -o hbm_patterns hbm_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr size_t kBytes = 1ull << 30;

__global__ void read_b128(const uint4 *__restrict__ p, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void read_b32(const uint32_t *__restrict__ p, size_t n, uint32_t *out) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void read_lds_b32(const uint32_t *__restrict__ p, size_t n, uint32_t *out) {
    __shared__ uint32_t lds[4][64 * 8];
    typedef __attribute__((address_space(1))) const void gvoid_t;
    typedef __attribute__((address_space(3))) void lvoid_t;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    const size_t waves = (size_t)gridDim.x * 4, w = blockIdx.x * 4 + wave;
    for (size_t base = w * 512; base + 512 <= n; base += waves * 512) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_global_load_lds((gvoid_t *)(p + base + k * 64 + lane), (lvoid_t *)&lds[wave][k * 64], 4, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += lds[wave][k * 64 + lane];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void gather_b16(const uint16_t *__restrict__ p, size_t n_windows, uint32_t *out) {
    // every wave instruction: 64 lanes at pseudo-random 2-byte positions inside one 614400-byte window (a 640x480
    // depth frame), 4800 instructions per window = 307200 gathers (one per pixel, as K3 issues them)
    uint32_t acc = 0;
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t job = wave; job < n_windows * 4800; job += waves) {
        const size_t win = job / 4800, k = job % 4800;
        const uint32_t h = (uint32_t)(k * 64 + lane) * 2654435761u;
        acc += p[win * 307200 + (h % 307200u)];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void copy_b128(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// K3's mix: read 2 bytes per pixel, write 4 bytes per pixel (+ the 1/8 byte bitset is ignored): 16-byte loads, 2 x 16-byte stores
__global__ void read1_write2(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        dst[i] = v;                                               // two fully coalesced output streams
        dst[n + i] = make_uint4(v.y, v.x, v.w, v.z);
    }
}

__global__ void copy_b128_nt(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 *s = (const u4 *)src;
    u4 *d = (u4 *)dst;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(&s[i]), &d[i]);
}

__global__ void read1_write2_nt(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 *s = (const u4 *)src;
    u4 *d = (u4 *)dst;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u4 v = __builtin_nontemporal_load(&s[i]);
        __builtin_nontemporal_store(v, &d[i]);                    // two fully coalesced output streams
        __builtin_nontemporal_store(v.yxwz, &d[n + i]);
    }
}

template <typename F>
static void timed(const char *name, double bytes, F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipEventRecord(a);
    for (int k = 0; k < 10; ++k) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    printf("| %s | %.2f GB | %.3f ms | %.2f TB/s |\n", name, bytes / 1e9, ms / 10, bytes * 10 / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv) {
    if (argc > 1) {                      // `hbm_patterns time`: hand-written streaming kernels, timed with HIP events
        void *s, *d;
        uint32_t *o;
        hipMalloc(&s, kBytes);
        hipMalloc(&d, 2 * kBytes);
        hipMalloc(&o, 4);
        hipMemset(s, 1, kBytes);
        hipMemset(d, 2, 2 * kBytes);
        printf("| kernel (16 B per lane, grid-stride) | bytes moved | time | rate |\n|---|---|---|---|\n");
        for (int blocks : {2048, 8192, 32768}) {
            char name[64];
            snprintf(name, sizeof name, "read only, %d blocks", blocks);
            timed(name, (double)kBytes, [&] { read_b128<<<blocks, 256>>>((const uint4 *)s, kBytes / 16, o); });
            snprintf(name, sizeof name, "copy 1:1, %d blocks", blocks);
            timed(name, 2.0 * kBytes, [&] { copy_b128<<<blocks, 256>>>((const uint4 *)s, (uint4 *)d, kBytes / 16); });
            snprintf(name, sizeof name, "read 1 : write 2, %d blocks", blocks);
            timed(name, 3.0 * kBytes, [&] { read1_write2<<<blocks, 256>>>((const uint4 *)s, (uint4 *)d, kBytes / 16); });
            snprintf(name, sizeof name, "copy 1:1 nontemporal, %d blocks", blocks);
            timed(name, 2.0 * kBytes, [&] { copy_b128_nt<<<blocks, 256>>>((const uint4 *)s, (uint4 *)d, kBytes / 16); });
            snprintf(name, sizeof name, "read 1 : write 2, nontemporal stores, %d blocks", blocks);
            timed(name, 3.0 * kBytes, [&] { read1_write2_nt<<<blocks, 256>>>((const uint4 *)s, (uint4 *)d, kBytes / 16); });
        }
        return 0;
    }
    void *buf;
    uint32_t *out;
    hipMalloc(&buf, kBytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, kBytes);
    for (int rep = 0; rep < 2; ++rep) {
        read_b128<<<4096, 256>>>((const uint4 *)buf, kBytes / 16, out);
        read_b32<<<4096, 256>>>((const uint32_t *)buf, kBytes / 4, out);
        read_lds_b32<<<4096, 256>>>((const uint32_t *)buf, kBytes / 4, out);
        gather_b16<<<4096, 256>>>((const uint16_t *)buf, kBytes / 614400, out);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: read_* %zu; gather_b16 touches %zu windows x 614400 B, issues %zu 2-byte loads\n", kBytes,
           kBytes / 614400, kBytes / 614400 * 307200);
    return 0;
}
