// Probe (round 4): why did the product kernel beat the trivial same-mix kernel?  Variants of hbm_ceilings.hip's walk at the
// minified ratio (200 quads out per 4 KiB tile in): SPIN dummy vector operations between the arrival of a range's bytes
// and its stores (the product computes ~1000 instructions there: do the waves have to fall out of lock-step?), the grid
// (waves per CU), stores deferred by one range.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/ceiling_probe.hip -o scripts/bin/ceiling_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT_LOAD, unsigned ROUNDS, int SPIN, int OCC, int ROT = 0>
__global__ __launch_bounds__(256, OCC) void k_walk(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                                   size_t ntiles, unsigned wquads, u32x4 *sink) {
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const size_t nranges = (ntiles + 7) / 8;
    u32x4 acc = {0u, 0u, 0u, 0u};
    u32x4 v[2][4], nx[2][4];
    auto load = [&](size_t range, u32x4 (&dst)[2][4]) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const size_t tt = range * 8 + 4 * j + w;
            const unsigned char *base = in + (tt < ntiles ? tt : ntiles - 1) * 4096;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32x4 *p = reinterpret_cast<const u32x4 *>(base + 16u * lane + 1024u * k);
                dst[j][k] = NT_LOAD ? __builtin_nontemporal_load(p) : *p;
            }
        }
    };
    // ROT: which workgroup takes which range of a round.  0: range = round * grid + blockIdx (a workgroup -- and, as
    // blockIdx mod 8 is the XCD, an XCD -- always takes the same residue of the range number mod 8: every 8th 32 KiB
    // chunk of the input); k: the residue moves by k per round; -1: the product's sharded-ticket order is emulated by a
    // hash of (round, blockIdx)
    const size_t G = gridDim.x;
    auto range_of = [&](size_t round) -> size_t {
        size_t slot = blockIdx.x;
        if (ROT > 0) slot = (blockIdx.x + round * ROT) % G;
        if (ROT < 0) slot = (blockIdx.x * 2654435761u + round * 40503u) % G;  // G is a power of two here: odd multiplier = permutation
        return round * G + slot;
    };
    size_t round = 0;
    size_t R = range_of(0);
    if (R < nranges) load(R, v);
    for (; round * G < nranges; round++, R = range_of(round)) {
        if (R >= nranges) continue;
        if (SPIN) {
            // dependent dummy work on the bytes that have arrived (kept: it feeds the stored value)
            unsigned x = v[0][0].x;
#pragma unroll 8
            for (int i = 0; i < SPIN; i++) x = x * 1664525u + 1013904223u;
            v[0][0].x = x;
        }
        { const size_t Rn = range_of(round + 1); load(Rn < nranges ? Rn : nranges - 1, nx); }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const size_t t = R * 8 + 4 * j + w;
            if (ROUNDS) {
                u32x4 *o = reinterpret_cast<u32x4 *>(out + (t < ntiles ? t : ntiles - 1) * (size_t)wquads * 16u);
#pragma unroll
                for (unsigned r = 0; r < ROUNDS; r++) {
                    const unsigned q = lane + 64u * r;
                    const u32x4 val = v[j][r & 3u];
                    if (r + 1 < ROUNDS || q < wquads) __builtin_nontemporal_store(val, o + q);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) acc ^= v[j][k];
            }
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) v[j][k] = nx[j][k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const size_t n = 1ull << 30, ntiles = n / 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned char *a, *b; u32x4 *sink;
    CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, n));
    auto run = [&](const char *name, unsigned wquads, int grid, auto kernel) {
        const double bytes = (double)n + (double)ntiles * wquads * 16.0;
        auto launch = [&] { hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, a, b, ntiles, wquads, sink); };
        for (int it = 0; it < 300; it++) launch();
        CK(hipEventRecord(e0));
        for (int it = 0; it < 300; it++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float tt; CK(hipEventElapsedTime(&tt, e0, e1));
        printf("%-64s %.4f ms %7.1f GB/s\n", name, tt / 300, bytes / (tt / 300) * 1e-6);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; rep++) {
        run("mix 1:0.78 plain ld, grid 1024 (16 waves/CU), no spin", 200, 1024, k_walk<false, 4, 0, 4>);
        run("mix 1:0.78 plain ld, grid 1024, XCD residue +1 per round", 200, 1024, k_walk<false, 4, 0, 4, 1>);
        run("mix 1:0.78 plain ld, grid 1024, XCD residue +3 per round", 200, 1024, k_walk<false, 4, 0, 4, 3>);
        run("mix 1:0.78 plain ld, grid 1024, hashed order", 200, 1024, k_walk<false, 4, 0, 4, -1>);
        run("mix 1:0.78 nt ld, grid 1024, XCD residue +1 per round", 200, 1024, k_walk<true, 4, 0, 4, 1>);
        run("mix 1:0.78 nt ld, grid 1024, hashed order", 200, 1024, k_walk<true, 4, 0, 4, -1>);
        run("read plain, grid 1024, XCD residue +1 per round", 0, 1024, k_walk<false, 0, 0, 4, 1>);
        run("read plain, grid 1024, hashed order", 0, 1024, k_walk<false, 0, 0, 4, -1>);
        run("read nt, grid 1024, hashed order", 0, 1024, k_walk<true, 0, 0, 4, -1>);
        run("mix 1:0.78 plain ld, grid 1024, spin 64", 200, 1024, k_walk<false, 4, 64, 4>);
        run("mix 1:0.78 plain ld, grid 1024, spin 256", 200, 1024, k_walk<false, 4, 256, 4>);
        run("mix 1:0.78 plain ld, grid 1024, spin 1024", 200, 1024, k_walk<false, 4, 1024, 4>);
        run("mix 1:0.78 plain ld, grid 1024, spin 2048", 200, 1024, k_walk<false, 4, 2048, 4>);
        run("mix 1:0.78 nt ld, grid 1024, spin 1024", 200, 1024, k_walk<true, 4, 1024, 4>);
        run("mix 1:0.78 plain ld, grid 512 (8 waves/CU), no spin", 200, 512, k_walk<false, 4, 0, 4>);
        run("mix 1:0.78 plain ld, grid 768 (12 waves/CU), no spin", 200, 768, k_walk<false, 4, 0, 4>);
        run("mix 1:0.78 plain ld, grid 2048 (32 waves/CU), no spin", 200, 2048, k_walk<false, 4, 0, 8>);
        run("mix 1:0.78 plain ld, grid 2048 (32 waves/CU), spin 1024", 200, 2048, k_walk<false, 4, 1024, 8>);
        run("read plain, grid 1024, no spin", 0, 1024, k_walk<false, 0, 0, 4>);
        run("read plain, grid 1024, spin 1024", 0, 1024, k_walk<false, 0, 1024, 4>);
        run("read nt, grid 1024, spin 1024", 0, 1024, k_walk<true, 0, 1024, 4>);
    }
    return 0;
}
