// What does a 64-lane gather cost the vector-memory front end (texture-address unit + L1) of a gfx950 CU?  K3's depth-2
// gather is a 2-byte load per lane along a slanted line of frame 2; DESIGN.md prices it from PMC counters (TA busy cycles /
// gather instructions = ~59 cycles).  This micro-benchmark measures the same thing directly, L2-resident data, every CU
// saturated (5 workgroups of 4 waves, 4 independent gathers in flight per wave like K3's row group):
//   row      lane L reads pixel (y, x + L)                 -- fully coalesced: 128 consecutive bytes
//   slant    lane L reads pixel (y + L / 5, x + L)         -- K3-like: ~13 rows, ~36 distinct dwords
//   column   lane L reads pixel (y + L, x)                 -- 64 rows: 64 distinct lines
//   scatter  hashed positions in the frame
// each as a 2-byte load (buffer_load_ushort, what K3 issues), as the aligned dword that holds the sample, and as a 16-byte
// load per lane (would a wider piece per address be as cheap as a narrow one?).
// Prints wave-instructions per microsecond per CU and the equivalent front-end cycles per instruction at the measured clock.
// Build: hipcc --offload-arch=gfx950 -O2 -o gather_rates gather_rates.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int W = 640, H = 480;

template <int PATTERN, int BYTES>
__global__ __launch_bounds__(256) void gather_kernel(const uint16_t *__restrict__ frame, int iters, uint32_t *out, long long *cycles) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    uint32_t acc = 0;
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t s = (wave * 7919u + (uint32_t)it * 104729u + (uint32_t)j * 13u);
            uint32_t x, y;
            if (PATTERN == 0) { y = (s >> 4) % H; x = (s % (W - 64)) + lane; }
            else if (PATTERN == 1) { y = (s >> 4) % (H - 16) + lane / 5 + j; x = (s % (W - 64)) + lane; }
            else if (PATTERN == 2) { y = (s >> 4) % (H - 64) + lane; x = s % W; }
            else { const uint32_t h = (s + lane) * 2654435761u; y = (h >> 8) % H; x = h % W; }
            const uint32_t pix = y * W + x;
            if (BYTES == 2) v[j] = frame[pix];
            else if (BYTES == 4) v[j] = reinterpret_cast<const uint32_t *>(frame)[pix >> 1];
            else {
                const uint4 q = reinterpret_cast<const uint4 *>(frame)[min(pix >> 3, (uint32_t)(W * H / 8 - 1))];
                v[j] = q.x ^ q.y ^ q.z ^ q.w;
            }
        }
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    if (acc == 0x12345678u) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cycles[0] = c1 - c0; cycles[1] = t1 - t0; }
}

template <int PATTERN, int BYTES>
static void run(const char *name, const uint16_t *frame, uint32_t *out, long long *cycles, int n_cu) {
    const int blocks = n_cu * 5, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((gather_kernel<PATTERN, BYTES>), dim3(blocks), dim3(256), 0, 0, frame, 200, out, cycles);
    hipEventRecord(e0);
    hipLaunchKernelGGL((gather_kernel<PATTERN, BYTES>), dim3(blocks), dim3(256), 0, 0, frame, iters, out, cycles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c[2];
    hipMemcpy(c, cycles, sizeof(c), hipMemcpyDeviceToHost);
    const double instr_per_cu = 5.0 * 4 * iters * 4;                      // waves per CU x iterations x gathers
    const double us = ms * 1e3;
    const double ghz = (double)c[0] / (c[1] * 10.0);                        // wall_clock64 ticks at 100 MHz
    printf("%-8s %2d B/lane: %8.2f wave-instr / us / CU   = %6.1f front-end cycles per instruction at %.2f GHz\n", name, BYTES,
           instr_per_cu / us, us * ghz * 1e3 / instr_per_cu, ghz);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    uint16_t *frame;
    uint32_t *out;
    long long *cycles;
    hipMalloc(&frame, W * H * 2 + 64);
    hipMalloc(&out, 64);
    hipMalloc(&cycles, 64);
    std::vector<uint16_t> h(W * H + 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(i * 2654435761u >> 13);
    hipMemcpy(frame, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("gather rates, %d CUs, one 640x480 u16 frame (L2-resident), 20 waves per CU, 4 gathers in flight per wave\n", n_cu);
    run<0, 2>("row", frame, out, cycles, n_cu);
    run<1, 2>("slant", frame, out, cycles, n_cu);
    run<2, 2>("column", frame, out, cycles, n_cu);
    run<3, 2>("scatter", frame, out, cycles, n_cu);
    run<0, 4>("row", frame, out, cycles, n_cu);
    run<1, 4>("slant", frame, out, cycles, n_cu);
    run<3, 4>("scatter", frame, out, cycles, n_cu);
    run<0, 16>("row", frame, out, cycles, n_cu);
    run<1, 16>("slant", frame, out, cycles, n_cu);
    run<3, 16>("scatter", frame, out, cycles, n_cu);
    return 0;
}
