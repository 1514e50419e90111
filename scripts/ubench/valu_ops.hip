// Microbenchmark: issue cost of individual VALU instruction kinds on gfx950 (cycles per
// wave64 instruction per SIMD at 4 waves/SIMD), to build the kernel's cost model.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define BODY(INS)                                                                                  \
    for (int i = 0; i < iters; i++) {                                                              \
        asm volatile(REP8(INS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k0), "v"(k1)); \
    }

template <int KIND>
__global__ void k(uint32_t *out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t k0 = 0x0F0F0F0Fu + blockIdx.x, k1 = 0x33333333u;
    if (KIND == 0) BODY("v_and_b32_e32 %0, %8, %0\n v_and_b32_e32 %1, %8, %1\n v_and_b32_e32 %2, %8, %2\n v_and_b32_e32 %3, %8, %3\n v_and_b32_e32 %4, %8, %4\n v_and_b32_e32 %5, %8, %5\n v_and_b32_e32 %6, %8, %6\n v_and_b32_e32 %7, %8, %7\n")
    if (KIND == 1) BODY("v_bfi_b32 %0, %8, %0, %1\n v_bfi_b32 %1, %8, %1, %2\n v_bfi_b32 %2, %8, %2, %3\n v_bfi_b32 %3, %8, %3, %4\n v_bfi_b32 %4, %8, %4, %5\n v_bfi_b32 %5, %8, %5, %6\n v_bfi_b32 %6, %8, %6, %7\n v_bfi_b32 %7, %8, %7, %0\n")
    if (KIND == 2) BODY("v_perm_b32 %0, %0, %1, %9\n v_perm_b32 %1, %1, %2, %9\n v_perm_b32 %2, %2, %3, %9\n v_perm_b32 %3, %3, %4, %9\n v_perm_b32 %4, %4, %5, %9\n v_perm_b32 %5, %5, %6, %9\n v_perm_b32 %6, %6, %7, %9\n v_perm_b32 %7, %7, %0, %9\n")
    if (KIND == 3) BODY("v_bitop3_b32 %0, %0, %1, %8 bitop3:0x48\n v_bitop3_b32 %1, %1, %2, %8 bitop3:0x48\n v_bitop3_b32 %2, %2, %3, %8 bitop3:0x48\n v_bitop3_b32 %3, %3, %4, %8 bitop3:0x48\n v_bitop3_b32 %4, %4, %5, %8 bitop3:0x48\n v_bitop3_b32 %5, %5, %6, %8 bitop3:0x48\n v_bitop3_b32 %6, %6, %7, %8 bitop3:0x48\n v_bitop3_b32 %7, %7, %0, %8 bitop3:0x48\n")
    if (KIND == 4) BODY("v_lshlrev_b32_e32 %0, 3, %0\n v_lshlrev_b32_e32 %1, 3, %1\n v_lshlrev_b32_e32 %2, 3, %2\n v_lshlrev_b32_e32 %3, 3, %3\n v_lshlrev_b32_e32 %4, 3, %4\n v_lshlrev_b32_e32 %5, 3, %5\n v_lshlrev_b32_e32 %6, 3, %6\n v_lshlrev_b32_e32 %7, 3, %7\n")
    if (KIND == 5) BODY("v_xor_b32_e32 %0, %1, %0\n v_xor_b32_e32 %1, %2, %1\n v_xor_b32_e32 %2, %3, %2\n v_xor_b32_e32 %3, %4, %3\n v_xor_b32_e32 %4, %5, %4\n v_xor_b32_e32 %5, %6, %5\n v_xor_b32_e32 %6, %7, %6\n v_xor_b32_e32 %7, %0, %7\n")
    if (KIND == 6) BODY("v_or3_b32 %0, %0, %1, %2\n v_or3_b32 %1, %1, %2, %3\n v_or3_b32 %2, %2, %3, %4\n v_or3_b32 %3, %3, %4, %5\n v_or3_b32 %4, %4, %5, %6\n v_or3_b32 %5, %5, %6, %7\n v_or3_b32 %6, %6, %7, %0\n v_or3_b32 %7, %7, %0, %1\n")
    if (KIND == 7) BODY("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
    if (KIND == 8) BODY("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %4\n v_lshl_or_b32 %4, %4, 3, %5\n v_lshl_or_b32 %5, %5, 3, %6\n v_lshl_or_b32 %6, %6, 3, %7\n v_lshl_or_b32 %7, %7, 3, %0\n")
    if (KIND == 9) BODY("v_add_u32_e32 %0, %1, %0\n v_add_u32_e32 %1, %2, %1\n v_add_u32_e32 %2, %3, %2\n v_add_u32_e32 %3, %4, %3\n v_add_u32_e32 %4, %5, %4\n v_add_u32_e32 %5, %6, %5\n v_add_u32_e32 %6, %7, %6\n v_add_u32_e32 %7, %0, %7\n ")
    if (KIND == 10) BODY("v_lshrrev_b32_e32 %0, 3, %0\n v_lshrrev_b32_e32 %1, 3, %1\n v_lshrrev_b32_e32 %2, 3, %2\n v_lshrrev_b32_e32 %3, 3, %3\n v_lshrrev_b32_e32 %4, 3, %4\n v_lshrrev_b32_e32 %5, 3, %5\n v_lshrrev_b32_e32 %6, 3, %6\n v_lshrrev_b32_e32 %7, 3, %7\n ")
    if (KIND == 11) BODY("v_mul_u32_u24_e32 %0, 0x81, %0\n v_mul_u32_u24_e32 %1, 0x81, %1\n v_mul_u32_u24_e32 %2, 0x81, %2\n v_mul_u32_u24_e32 %3, 0x81, %3\n v_mul_u32_u24_e32 %4, 0x81, %4\n v_mul_u32_u24_e32 %5, 0x81, %5\n v_mul_u32_u24_e32 %6, 0x81, %6\n v_mul_u32_u24_e32 %7, 0x81, %7\n ")
    if (KIND == 12) BODY("v_alignbit_b32 %0, %0, %1, 4\n v_alignbit_b32 %1, %1, %2, 4\n v_alignbit_b32 %2, %2, %3, 4\n v_alignbit_b32 %3, %3, %4, 4\n v_alignbit_b32 %4, %4, %5, 4\n v_alignbit_b32 %5, %5, %6, 4\n v_alignbit_b32 %6, %6, %7, 4\n v_alignbit_b32 %7, %7, %0, 4\n ")
    if (KIND == 13) BODY("v_and_or_b32 %0, %0, %8, %1\n v_and_or_b32 %1, %1, %8, %2\n v_and_or_b32 %2, %2, %8, %3\n v_and_or_b32 %3, %3, %8, %4\n v_and_or_b32 %4, %4, %8, %5\n v_and_or_b32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0\n ")
    if (KIND == 14) BODY("v_xad_u32 %0, %0, %8, %1\n v_xad_u32 %1, %1, %8, %2\n v_xad_u32 %2, %2, %8, %3\n v_xad_u32 %3, %3, %8, %4\n v_xad_u32 %4, %4, %8, %5\n v_xad_u32 %5, %5, %8, %6\n v_xad_u32 %6, %6, %8, %7\n v_xad_u32 %7, %7, %8, %0\n ")
    if (KIND == 15) BODY("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc\n v_cndmask_b32_e32 %4, %4, %5, vcc\n v_cndmask_b32_e32 %5, %5, %6, vcc\n v_cndmask_b32_e32 %6, %6, %7, vcc\n v_cndmask_b32_e32 %7, %7, %0, vcc\n ")
    if (KIND == 16) BODY("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8\n ")
    if (KIND == 17) BODY("v_pk_lshlrev_b16 %0, 3, %0\n v_pk_lshlrev_b16 %1, 3, %1\n v_pk_lshlrev_b16 %2, 3, %2\n v_pk_lshlrev_b16 %3, 3, %3\n v_pk_lshlrev_b16 %4, 3, %4\n v_pk_lshlrev_b16 %5, 3, %5\n v_pk_lshlrev_b16 %6, 3, %6\n v_pk_lshlrev_b16 %7, 3, %7\n ")
    if (KIND == 18) BODY("v_pk_lshrrev_b16 %0, 7, %0\n v_pk_lshrrev_b16 %1, 7, %1\n v_pk_lshrrev_b16 %2, 7, %2\n v_pk_lshrrev_b16 %3, 7, %3\n v_pk_lshrrev_b16 %4, 7, %4\n v_pk_lshrrev_b16 %5, 7, %5\n v_pk_lshrrev_b16 %6, 7, %6\n v_pk_lshrrev_b16 %7, 7, %7\n ")
    if (KIND == 19) BODY("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %3, %4\n v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %5, %6\n v_mov_b32_e32 %6, %7\n v_mov_b32_e32 %7, %0\n ")
    if (KIND == 20) BODY("v_ffbl_b32_e32 %0, %0\n v_ffbl_b32_e32 %1, %1\n v_ffbl_b32_e32 %2, %2\n v_ffbl_b32_e32 %3, %3\n v_ffbl_b32_e32 %4, %4\n v_ffbl_b32_e32 %5, %5\n v_ffbl_b32_e32 %6, %6\n v_ffbl_b32_e32 %7, %7\n ")
    if (KIND == 21) BODY("v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n v_bcnt_u32_b32 %3, %3, %4\n v_bcnt_u32_b32 %4, %4, %5\n v_bcnt_u32_b32 %5, %5, %6\n v_bcnt_u32_b32 %6, %6, %7\n v_bcnt_u32_b32 %7, %7, %0\n ")
    if (KIND == 22) BODY("v_not_b32_e32 %0, %0\n v_not_b32_e32 %1, %1\n v_not_b32_e32 %2, %2\n v_not_b32_e32 %3, %3\n v_not_b32_e32 %4, %4\n v_not_b32_e32 %5, %5\n v_not_b32_e32 %6, %6\n v_not_b32_e32 %7, %7\n ")
    if (KIND == 23) BODY("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n ")
    if (KIND == 24) BODY("v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %1, %1, 3, %2\n v_lshl_add_u32 %2, %2, 3, %3\n v_lshl_add_u32 %3, %3, 3, %4\n v_lshl_add_u32 %4, %4, 3, %5\n v_lshl_add_u32 %5, %5, 3, %6\n v_lshl_add_u32 %6, %6, 3, %7\n v_lshl_add_u32 %7, %7, 3, %0\n ")
    if (KIND == 25) BODY("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %1, %1, %2\n v_pk_add_u16 %2, %2, %3\n v_pk_add_u16 %3, %3, %4\n v_pk_add_u16 %4, %4, %5\n v_pk_add_u16 %5, %5, %6\n v_pk_add_u16 %6, %6, %7\n v_pk_add_u16 %7, %7, %0\n ")
    if (KIND == 26) BODY("v_sub_u32_e32 %0, %1, %0\n v_sub_u32_e32 %1, %2, %1\n v_sub_u32_e32 %2, %3, %2\n v_sub_u32_e32 %3, %4, %3\n v_sub_u32_e32 %4, %5, %4\n v_sub_u32_e32 %5, %6, %5\n v_sub_u32_e32 %6, %7, %6\n v_sub_u32_e32 %7, %0, %7\n ")
    if (KIND == 27) BODY("v_or_b32_e32 %0, %1, %0\n v_or_b32_e32 %1, %2, %1\n v_or_b32_e32 %2, %3, %2\n v_or_b32_e32 %3, %4, %3\n v_or_b32_e32 %4, %5, %4\n v_or_b32_e32 %5, %6, %5\n v_or_b32_e32 %6, %7, %6\n v_or_b32_e32 %7, %0, %7\n ")
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int KIND>
void run(const char *name, uint32_t *out, int cus) {
    const int iters = 400;  // x 64 instructions
    for (int wps : {4}) {
        dim3 grid(cus * wps), block(256);
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, grid, block, 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)iters * 64;
        printf("%-16s waves/SIMD %d: %.2f cycles/instr/SIMD (assuming 2.4 GHz)\n", name, wps, ms * 1e-3 * 2.4e9 / (instr * wps));
    }
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    uint32_t *out; hipMalloc(&out, 64 << 20);
    run<0>("v_and_b32 (VOP2)", out, prop.multiProcessorCount);
    run<5>("v_xor_b32 2vgpr", out, prop.multiProcessorCount);
    run<4>("v_lshlrev_b32", out, prop.multiProcessorCount);
    run<1>("v_bfi_b32", out, prop.multiProcessorCount);
    run<2>("v_perm_b32", out, prop.multiProcessorCount);
    run<3>("v_bitop3_b32", out, prop.multiProcessorCount);
    run<6>("v_or3_b32", out, prop.multiProcessorCount);
    run<8>("v_lshl_or_b32", out, prop.multiProcessorCount);
    run<7>("v_add_u32_dpp", out, prop.multiProcessorCount);
    run<9>("v_add_u32", out, prop.multiProcessorCount);
    run<10>("v_lshrrev_b32", out, prop.multiProcessorCount);
    run<11>("v_mul_u32_u24", out, prop.multiProcessorCount);
    run<12>("v_alignbit_b32", out, prop.multiProcessorCount);
    run<13>("v_and_or_b32", out, prop.multiProcessorCount);
    run<14>("v_xad_u32", out, prop.multiProcessorCount);
    run<15>("v_cndmask_b32", out, prop.multiProcessorCount);
    run<16>("v_bfe_u32", out, prop.multiProcessorCount);
    run<17>("v_pk_lshlrev_b16", out, prop.multiProcessorCount);
    run<18>("v_pk_lshrrev_b16", out, prop.multiProcessorCount);
    run<19>("v_mov_b32", out, prop.multiProcessorCount);
    run<20>("v_ffbl_b32", out, prop.multiProcessorCount);
    run<21>("v_bcnt_u32_b32", out, prop.multiProcessorCount);
    run<22>("v_not_b32", out, prop.multiProcessorCount);
    run<23>("v_mad_u32_u24", out, prop.multiProcessorCount);
    run<24>("v_lshl_add_u32", out, prop.multiProcessorCount);
    run<25>("v_pk_add_u16", out, prop.multiProcessorCount);
    run<26>("v_sub_u32", out, prop.multiProcessorCount);
    run<27>("v_or_b32", out, prop.multiProcessorCount);
    return 0;
}
