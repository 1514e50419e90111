// Why do the trivial kernels (and the product) dip at 8 KiB of output per 4 KiB tile (d = 0.5)?  Write-only and read + write
// walks of scripts/ubench/hbm_ceilings.hip at w = 6 .. 16 KiB per tile, with the 1 KiB store rounds of a tile issued
// (0) ascending, (1) rotated by the tile's position in its range, (2) rotated by a hash of the tile number.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/ceiling_probe4.hip -o scripts/bin/ceiling_probe4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int ORDER, unsigned ROUNDS>
__global__ __launch_bounds__(256, 4) void k_walk(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                                 size_t ntiles, u32x4 *sink) {
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const size_t G = gridDim.x;
    auto load_tile = [&](size_t tt, u32x4 (&dst)[4]) {
        const unsigned char *base = in + (tt < ntiles ? tt : ntiles - 1) * 4096;
#pragma unroll
        for (int k = 0; k < 4; k++) dst[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(base + 16u * lane + 1024u * k));
    };
    auto store_tile = [&](size_t tt, const u32x4 (&src)[4]) {
        const size_t t = tt < ntiles ? tt : ntiles - 1;
        u32x4 *o = reinterpret_cast<u32x4 *>(out + t * (size_t)ROUNDS * 1024u);
        const u32x4 f = src[0] ^ src[1] ^ src[2] ^ src[3];
        unsigned rot = 0;
        if (ORDER == 1) rot = ((unsigned)t & 7u) * ROUNDS / 8u;
        if (ORDER == 2) rot = (((unsigned)t * 2654435761u) >> 16) % ROUNDS;
#pragma unroll
        for (unsigned r = 0; r < ROUNDS; r++) {
            unsigned rr = r + rot;
            if (rr >= ROUNDS) rr -= ROUNDS;
            __builtin_nontemporal_store(f, o + lane + 64u * rr);
        }
    };
    if (MODE == 1) {
        const u32x4 c[4] = {{lane, 1, 2, 3}, {lane, 5, 6, 7}, {lane, 9, 10, 11}, {lane, 13, 14, 15}};
        for (size_t R = blockIdx.x; R * 8 < ntiles; R += G) {
            store_tile(R * 8 + w, c);
            store_tile(R * 8 + 4 + w, c);
        }
    } else {
        u32x4 v[2][4], nx[2][4];
        size_t R = blockIdx.x;
        load_tile(R * 8 + w, v[0]);
        load_tile(R * 8 + 4 + w, v[1]);
        for (; R * 8 < ntiles; R += G) {
            load_tile((R + G) * 8 + w, nx[0]);
            load_tile((R + G) * 8 + 4 + w, nx[1]);
            store_tile(R * 8 + w, v[0]);
            store_tile(R * 8 + 4 + w, v[1]);
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) v[j][k] = nx[j][k];
        }
    }
}

template <int MODE, int ORDER>
static void launch(int grid, unsigned rounds, const unsigned char *in, unsigned char *out, size_t ntiles, u32x4 *sink) {
#define C(R) case R: hipLaunchKernelGGL((k_walk<MODE, ORDER, R>), dim3(grid), dim3(256), 0, 0, in, out, ntiles, sink); break;
    switch (rounds) { C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(12) C(16) default: break; }
#undef C
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const size_t n = (size_t)1 << 30, ntiles = n / 4096;
    const int grid = p.multiProcessorCount * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned char *a, *b; u32x4 *sink;
    CK(hipMalloc(&a, n)); CK(hipMalloc(&b, 4 * n + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, 4 * n));
    printf("GB/s of (bytes read + bytes written), 200 launches behind 200 untimed ones; rows: KiB written per 4 KiB tile\n");
    printf("%8s | %10s %10s %10s | %10s %10s %10s\n", "KiB/tile", "write asc", "write rot", "write hash", "mix asc", "mix rot", "mix hash");
    const unsigned rs[] = {3, 4, 5, 6, 7, 8, 9, 10, 12, 16};
    for (unsigned rounds : rs) {
        double r[6];
        for (int v = 0; v < 6; v++) {
            auto go = [&] {
                switch (v) {
                    case 0: launch<1, 0>(grid, rounds, a, b, ntiles, sink); break;
                    case 1: launch<1, 1>(grid, rounds, a, b, ntiles, sink); break;
                    case 2: launch<1, 2>(grid, rounds, a, b, ntiles, sink); break;
                    case 3: launch<0, 0>(grid, rounds, a, b, ntiles, sink); break;
                    case 4: launch<0, 1>(grid, rounds, a, b, ntiles, sink); break;
                    case 5: launch<0, 2>(grid, rounds, a, b, ntiles, sink); break;
                }
            };
            for (int i = 0; i < 200; i++) go();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 200; i++) go();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            const double bytes = (v < 3 ? 0.0 : (double)n) + (double)ntiles * rounds * 1024.0;
            r[v] = bytes / (t / 200) * 1e-6;
        }
        printf("%8u | %10.1f %10.1f %10.1f | %10.1f %10.1f %10.1f\n", rounds, r[0], r[1], r[2], r[3], r[4], r[5]);
        fflush(stdout);
    }
    return 0;
}
