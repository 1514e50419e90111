// Does the SHAPE of the stage-1 kernel's tile loads cost read bandwidth?  (a sparse input is a
// pure read test -- 4.9-5.2 TB/s where trivial kernels read 5.6-6.4.)  Persistent grid of 1024 x 256 threads like the
// kernel's, every wave walks 4 KiB tiles with a fixed stride and XOR-folds what it loaded; TILES tiles are requested
// before the first is used (the kernel: 2).  Shapes:
//   block      lane l loads 4 x 16 B at tile + 64 l + 16 k    -- the kernel's: a lane owns one 64-byte block, each
//              wave instruction touches 64 pieces of 64 separate 64-byte blocks (the other three hit L1)
//   coalesced  lane l loads 4 x 16 B at tile + 16 l + 1024 k  -- each wave instruction is 1 KiB contiguous
//   lds-dma    global_load_lds_dwordx4: 1 KiB contiguous per wave instruction straight into LDS (no VGPRs while in
//              flight), then the lane's own 64 bytes as 4 x ds_read_b128 (quarter c of block L at slot 4 L + (c ^ ((L >> 1) & 3)):
//              conflict-free) -- what north_star literally describes
// each with plain and with non-temporal loads.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/read_shape.hip -o gpurun_out/read_shape
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SHAPE, bool NT, int TILES>
__global__ __launch_bounds__(256, 4) void k_tiles(const unsigned char *__restrict__ in, size_t ntiles, u32x4 *sink) {
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (size_t t = wave * TILES; t < ntiles; t += nwaves * TILES) {
        u32x4 v[TILES][4];
#pragma unroll
        for (int j = 0; j < TILES; j++) {
            const size_t tt = t + j < ntiles ? t + j : ntiles - 1;
            const unsigned char *base = in + tt * 4096;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32x4 *p = reinterpret_cast<const u32x4 *>(base + (SHAPE == 0 ? 64u * lane + 16u * k : 16u * lane + 1024u * k));
                v[j][k] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        }
#pragma unroll
        for (int j = 0; j < TILES; j++)
#pragma unroll
            for (int k = 0; k < 4; k++) acc ^= v[j][k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;  // never true in practice
}

// LDS-DMA: TILES x 4 KiB per wave in flight in LDS, double-buffered (2 x TILES tiles of LDS per wave)
template <bool NT, int TILES>
__global__ __launch_bounds__(256, 4) void k_tiles_lds(const unsigned char *__restrict__ in, size_t ntiles, u32x4 *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4][2][TILES][4096];
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const size_t wave = (size_t)blockIdx.x * 4 + w, nwaves = (size_t)gridDim.x * 4;
    u32x4 acc = {0u, 0u, 0u, 0u};
    // lane l of DMA instruction k fetches the 16-byte chunk that belongs at LDS slot 64 k + l: slot s holds quarter
    // (s & 3) ^ ((s >> 3) & 3) of block s >> 2 (the swizzle that makes the 64-byte-stride reads conflict-free)
    auto issue = [&](size_t t, int buf) {
#pragma unroll
        for (int j = 0; j < TILES; j++) {
            const size_t tt = t + j < ntiles ? t + j : ntiles - 1;
            const unsigned char *base = in + tt * 4096;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned s = 64u * k + lane, blk = s >> 2, q = (s & 3u) ^ ((blk >> 1) & 3u);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + 64u * blk + 16u * q),
                                                 (__attribute__((address_space(3))) void *)(&lds[w][buf][j][1024 * k]), 16, 0,
                                                 NT ? 2 : 0);
            }
        }
    };
    size_t t = wave * TILES;
    int buf = 0;
    if (t < ntiles) issue(t, 0);
    for (; t < ntiles; t += nwaves * TILES) {
        const size_t tn = t + nwaves * TILES;
        if (tn < ntiles) {
            issue(tn, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * TILES) : "memory");  // the older batch has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int j = 0; j < TILES; j++)
#pragma unroll
            for (int c = 0; c < 4; c++)
                acc ^= *reinterpret_cast<const u32x4 *>(&lds[w][buf][j][64u * lane + 16u * (c ^ ((lane >> 1) & 3u))]);
        buf ^= 1;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs; persistent grid 1024 x 256 threads, waves walk 4 KiB tiles\n", p.name, p.multiProcessorCount);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t gib : {1, 4}) {
        const size_t n = gib << 30, ntiles = n / 4096;
        unsigned char *a; u32x4 *sink; CK(hipMalloc(&a, n)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(a, 1, n));
        auto timeit = [&](const char *name, auto launch) {
            std::vector<float> ms;
            for (int it = 0; it < 25; it++) {
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 5) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("%zu GiB %-44s median %.4f ms  %.1f GB/s (best %.1f)\n", gib, name, ms[ms.size() / 2], (double)n / ms[ms.size() / 2] * 1e-6,
                   (double)n / ms[0] * 1e-6);
        };
#define RUN(name, ...) timeit(name, [&] { hipLaunchKernelGGL((__VA_ARGS__), dim3(1024), dim3(256), 0, 0, a, ntiles, sink); })
        for (int rep = 0; rep < 2; rep++) {
            RUN("block shape, plain, 2 tiles in flight", k_tiles<0, false, 2>);
            RUN("block shape, nt, 2 tiles in flight", k_tiles<0, true, 2>);
            RUN("coalesced, plain, 2 tiles in flight", k_tiles<1, false, 2>);
            RUN("coalesced, nt, 2 tiles in flight", k_tiles<1, true, 2>);
            RUN("block shape, plain, 4 tiles in flight", k_tiles<0, false, 4>);
            RUN("coalesced, plain, 4 tiles in flight", k_tiles<1, false, 4>);
            RUN("coalesced, nt, 4 tiles in flight", k_tiles<1, true, 4>);
            RUN("block shape, plain, 1 tile in flight", k_tiles<0, false, 1>);
            RUN("lds-dma, plain, 1 + 1 tiles (8 KiB LDS/wave)", k_tiles_lds<false, 1>);
            RUN("lds-dma, nt, 1 + 1 tiles", k_tiles_lds<true, 1>);
            RUN("lds-dma, plain, 2 + 2 tiles (16 KiB LDS/wave)", k_tiles_lds<false, 2>);
            RUN("lds-dma, nt, 2 + 2 tiles", k_tiles_lds<true, 2>);
        }
        CK(hipFree(a)); CK(hipFree(sink));
    }
    return 0;
}
