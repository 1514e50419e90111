// Microbenchmark: sustained integer VALU issue rate per SIMD on gfx950, for
// 1..8 waves per SIMD.  Sets the VALU roofline used in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OPS>
__global__ void k_valu(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < OPS / 8; j++) {
            a0 = (a0 ^ a1) & 0x7F7F7F7Fu; a1 = (a1 | a2) ^ 0x01010101u;
            a2 = (a2 & a3) | 0x10u;        a3 = (a3 ^ a4) + 3u;
            a4 = (a4 | a5) & 0xFFFEFFFFu;  a5 = (a5 ^ a6) | 0x20u;
            a6 = (a6 & a7) ^ 0x33u;        a7 = (a7 ^ a0) & 0xF0F0F0FFu;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
    uint32_t *out; hipMalloc(&out, 64 << 20);
    const int iters = 2000;
    constexpr int OPS = 64;  // source-level ops per iteration (2 instr each: 16 per line)
    for (int wps = 1; wps <= 8; wps++) {
        // blocks of 256 threads = 4 waves = 1 wave per SIMD per block; wps blocks per CU
        dim3 grid(cus * wps), block(256);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_valu<OPS>, grid, block, 0, 0, out, iters, 1u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_valu<OPS>, grid, block, 0, 0, out, iters, 2u);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // instruction count per wave: iters * OPS/8 * 16 VALU (each statement = 2 ops)
        double instr_per_wave = (double)iters * (OPS / 8) * 16;
        double waves_per_simd = wps;
        double cyc = ms * 1e-3 * 2.4e9;
        printf("waves/SIMD %d: %.3f ms, cycles/instr/SIMD (at 2.4 GHz) = %.2f\n", wps, ms,
               cyc / (instr_per_wave * waves_per_simd));
    }
    return 0;
}
