// Microbenchmark: what a select costs on gfx950 -- v_cndmask on a mask in SGPRs / in VCC, the compare that makes the
// mask, both together, and the all-VGPR alternatives (v_bfi_b32, v_bitop3_b32 on a 0 / ~0 word).  Cycles per wave64
// instruction per SIMD at 4 waves per SIMD (issue-bound loops of 64 instructions).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_select.hip -o gpurun_out/valu_select
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define LOOP(INS, ...)                                                                                                  \
    for (int i = 0; i < iters; i++) {                                                                                     \
        asm volatile(REP8(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k0), "v"(k1), "s"(m) __VA_ARGS__); \
    }

template <int KIND>
__global__ void k(uint32_t *out, int iters, uint64_t m) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t k0 = 0x0F0F0F0Fu + blockIdx.x, k1 = 0x33333333u;
    if (KIND == 0) LOOP("v_cndmask_b32_e64 %0, %0, %1, %10\n v_cndmask_b32_e64 %1, %1, %2, %10\n v_cndmask_b32_e64 %2, %2, %3, %10\n v_cndmask_b32_e64 %3, %3, %4, %10\n v_cndmask_b32_e64 %4, %4, %5, %10\n v_cndmask_b32_e64 %5, %5, %6, %10\n v_cndmask_b32_e64 %6, %6, %7, %10\n v_cndmask_b32_e64 %7, %7, %0, %10\n ")
    if (KIND == 1) { asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
        LOOP("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_cndmask_b32_e32 %5, %5, %6, vcc\n v_cndmask_b32_e32 %6, %6, %7, vcc\n v_cndmask_b32_e32 %7, %7, %0, vcc\n ", : "vcc") }
    if (KIND == 2) LOOP("v_cmp_lt_u32_e64 s[20:21], %0, %1\n v_cmp_lt_u32_e64 s[22:23], %1, %2\n v_cmp_lt_u32_e64 s[20:21], %2, %3\n v_cmp_lt_u32_e64 s[22:23], %3, %4\n v_cmp_lt_u32_e64 s[20:21], %4, %5\n v_cmp_lt_u32_e64 s[22:23], %5, %6\n v_cmp_lt_u32_e64 s[20:21], %6, %7\n v_cmp_lt_u32_e64 s[22:23], %7, %0\n ", : "s20", "s21", "s22", "s23")
    if (KIND == 3) LOOP("v_cmp_lt_u32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc\n v_cmp_lt_u32_e32 vcc, %2, %3\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cmp_lt_u32_e32 vcc, %4, %5\n v_cndmask_b32_e32 %4, %4, %6, vcc\n v_cmp_lt_u32_e32 vcc, %6, %7\n v_cndmask_b32_e32 %6, %6, %0, vcc\n ", : "vcc")
    if (KIND == 4) LOOP("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n ")
    if (KIND == 5) LOOP("v_bitop3_b32 %0, %8, %0, %1 bitop3:0xca\n v_bitop3_b32 %1, %8, %1, %2 bitop3:0xca\n v_bitop3_b32 %2, %8, %2, %3 bitop3:0xca\n v_bitop3_b32 %3, %8, %3, %4 bitop3:0xca\n v_bitop3_b32 %4, %8, %4, %5 bitop3:0xca\n v_bitop3_b32 %5, %8, %5, %6 bitop3:0xca\n v_bitop3_b32 %6, %8, %6, %7 bitop3:0xca\n v_bitop3_b32 %7, %8, %7, %0 bitop3:0xca\n ")
    if (KIND == 6) LOOP("v_cmp_lt_u32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cmp_lt_u32_e64 s[22:23], %2, %3\n s_nop 1\n v_cndmask_b32_e64 %2, %2, %4, s[22:23]\n v_cmp_lt_u32_e64 s[20:21], %4, %5\n s_nop 1\n v_cndmask_b32_e64 %4, %4, %6, s[20:21]\n v_cmp_lt_u32_e64 s[22:23], %6, %7\n s_nop 1\n v_cndmask_b32_e64 %6, %6, %0, s[22:23]\n ", : "s20", "s21", "s22", "s23")
    if (KIND == 7) LOOP("v_min_u32_e32 %0, %0, %1\n v_min_u32_e32 %1, %1, %2\n v_min_u32_e32 %2, %2, %3\n v_min_u32_e32 %3, %3, %4\n v_min_u32_e32 %4, %4, %5\n v_min_u32_e32 %5, %5, %6\n v_min_u32_e32 %6, %6, %7\n v_min_u32_e32 %7, %7, %0\n ")
    if (KIND == 8) LOOP("v_ffbh_u32_e32 %0, %1\n v_ffbh_u32_e32 %1, %2\n v_ffbh_u32_e32 %2, %3\n v_ffbh_u32_e32 %3, %4\n v_ffbh_u32_e32 %4, %5\n v_ffbh_u32_e32 %5, %6\n v_ffbh_u32_e32 %6, %7\n v_ffbh_u32_e32 %7, %0\n ")
    if (KIND == 9) LOOP("v_ashrrev_i32_e32 %0, 31, %1\n v_ashrrev_i32_e32 %1, 31, %2\n v_ashrrev_i32_e32 %2, 31, %3\n v_ashrrev_i32_e32 %3, 31, %4\n v_ashrrev_i32_e32 %4, 31, %5\n v_ashrrev_i32_e32 %5, 31, %6\n v_ashrrev_i32_e32 %6, 31, %7\n v_ashrrev_i32_e32 %7, 31, %0\n ")
    if (KIND == 10) LOOP("v_lshrrev_b32_e32 %0, %1, %0\n v_lshrrev_b32_e32 %1, %2, %1\n v_lshrrev_b32_e32 %2, %3, %2\n v_lshrrev_b32_e32 %3, %4, %3\n v_lshrrev_b32_e32 %4, %5, %4\n v_lshrrev_b32_e32 %5, %6, %5\n v_lshrrev_b32_e32 %6, %7, %6\n v_lshrrev_b32_e32 %7, %0, %7\n ")
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int KIND>
void run(const char *name, uint32_t *out, int cus, double per_rep) {
    const int iters = 400;
    dim3 grid(cus * 4), block(256);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters, 0x5555AAAA0F0F3333ull);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters, 0x5555AAAA0F0F3333ull);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.2f cycles per instruction and SIMD (2.4 GHz assumed; %g vector instructions per loop body)\n", name,
           ms * 1e-3 * 2.4e9 / ((double)iters * per_rep * 4), per_rep);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    uint32_t *out; hipMalloc(&out, 64 << 20);
    const int cus = prop.multiProcessorCount;
    run<0>("v_cndmask_b32_e64 (mask in an SGPR pair)", out, cus, 64);
    run<1>("v_cndmask_b32_e32 (mask in VCC)", out, cus, 64);
    run<2>("v_cmp_lt_u32_e64 -> SGPR pair", out, cus, 64);
    run<3>("v_cmp_e32 -> vcc + v_cndmask_e32 pairs", out, cus, 64);
    run<6>("v_cmp_e64 + s_nop 1 + v_cndmask_e64 triples", out, cus, 64);
    run<4>("v_bfi_b32", out, cus, 64);
    run<5>("v_bitop3_b32 0xca (select by a VGPR mask)", out, cus, 64);
    run<7>("v_min_u32", out, cus, 64);
    run<8>("v_ffbh_u32", out, cus, 64);
    run<9>("v_ashrrev_i32", out, cus, 64);
    run<10>("v_lshrrev_b32 (variable)", out, cus, 64);
    return 0;
}
