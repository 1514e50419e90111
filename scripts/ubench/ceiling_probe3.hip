// Probe 3 (round 4): do the trivial kernels lose to the product because every tile's output is exactly 3 200 bytes (all
// concurrent writers at multiples of one stride) where the product's tiles emit data-dependent amounts?  Tile t writes
// wq(t) quads, a period-8 pattern around the same mean of 200 (offsets in closed form).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/ceiling_probe3.hip -o scripts/bin/ceiling_probe3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Pat { unsigned wq[8], off[8], total; };
__constant__ Pat g_pat;

template <int MODE, bool NT_LOAD, bool VARY>
__global__ __launch_bounds__(256, 4) void k(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, size_t ntiles) {
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const size_t G = gridDim.x;
    auto load_tile = [&](size_t tt, u32x4 (&dst)[4]) {
        const unsigned char *base = in + (tt < ntiles ? tt : ntiles - 1) * 4096;
#pragma unroll
        for (int kq = 0; kq < 4; kq++) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(base + 16u * lane + 1024u * kq);
            dst[kq] = NT_LOAD ? __builtin_nontemporal_load(p) : *p;
        }
    };
    auto store_tile = [&](size_t tt, const u32x4 (&src)[4]) {
        if (tt >= ntiles) tt = ntiles - 1;
        const unsigned wq = VARY ? g_pat.wq[tt & 7] : 200u;
        const size_t off = VARY ? (tt >> 3) * (size_t)g_pat.total + g_pat.off[tt & 7] : tt * 200u;
        u32x4 *o = reinterpret_cast<u32x4 *>(out) + off;
#pragma unroll
        for (unsigned r = 0; r < 4; r++) {
            const unsigned q = lane + 64u * r;
            if (q < wq) __builtin_nontemporal_store(src[r & 3u], o + q);
        }
    };
    if (MODE == 1) {
        const u32x4 c[4] = {{lane, 1, 2, 3}, {lane, 5, 6, 7}, {lane, 9, 10, 11}, {lane, 13, 14, 15}};
        for (size_t R = blockIdx.x; R * 8 < ntiles; R += G) {
            store_tile(R * 8 + w, c);
            store_tile(R * 8 + 4 + w, c);
        }
    } else {
        u32x4 a0[2][4], a1[2][4], a2[2][4];
        size_t R = blockIdx.x;
        load_tile(R * 8 + w, a0[0]); load_tile(R * 8 + 4 + w, a0[1]);
        load_tile((R + G) * 8 + w, a1[0]); load_tile((R + G) * 8 + 4 + w, a1[1]);
        for (; R * 8 < ntiles; R += 3 * G) {
            load_tile((R + 2 * G) * 8 + w, a2[0]); load_tile((R + 2 * G) * 8 + 4 + w, a2[1]);
            store_tile(R * 8 + w, a0[0]); store_tile(R * 8 + 4 + w, a0[1]);
            load_tile((R + 3 * G) * 8 + w, a0[0]); load_tile((R + 3 * G) * 8 + 4 + w, a0[1]);
            if ((R + G) * 8 < ntiles) { store_tile((R + G) * 8 + w, a1[0]); store_tile((R + G) * 8 + 4 + w, a1[1]); }
            load_tile((R + 4 * G) * 8 + w, a1[0]); load_tile((R + 4 * G) * 8 + 4 + w, a1[1]);
            if ((R + 2 * G) * 8 < ntiles) { store_tile((R + 2 * G) * 8 + w, a2[0]); store_tile((R + 2 * G) * 8 + 4 + w, a2[1]); }
        }
    }
}

int main() {
    const size_t n = 1ull << 30, ntiles = n / 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned char *a, *b;
    CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n + 65536));
    CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, n));
    Pat p = {{168, 232, 184, 216, 152, 248, 200, 200}, {}, 0};  // whole 128-byte lines (multiples of 8 quads), mean 200
    for (int i = 0; i < 8; i++) { p.off[i] = p.total; p.total += p.wq[i]; }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pat), &p, sizeof p));
    auto run = [&](const char *name, double bytes, auto kernel) {
        auto launch = [&] { hipLaunchKernelGGL(kernel, dim3(1024), dim3(256), 0, 0, a, b, ntiles); };
        for (int it = 0; it < 300; it++) launch();
        CK(hipEventRecord(e0));
        for (int it = 0; it < 300; it++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float tt; CK(hipEventElapsedTime(&tt, e0, e1));
        printf("%-72s %.4f ms %7.1f GB/s\n", name, tt / 300, bytes / (tt / 300) * 1e-6);
        fflush(stdout);
    };
    const double W = (double)ntiles * 3200.0, N = (double)n;
    for (int rep = 0; rep < 2; rep++) {
        run("write only, 200 quads per tile", W, k<1, false, false>);
        run("write only, 152..248 quads per tile (mean 200)", W, k<1, false, true>);
        run("mix deferred, plain ld, 200 quads per tile", N + W, k<2, false, false>);
        run("mix deferred, plain ld, 152..248 quads per tile", N + W, k<2, false, true>);
        run("mix deferred, nt ld, 200 quads per tile", N + W, k<2, true, false>);
        run("mix deferred, nt ld, 152..248 quads per tile", N + W, k<2, true, true>);
    }
    return 0;
}
