// Probe 2 (round 4): the trivial same-mix kernel runs below the sum of its read time and its write time.  Pure write
// rate, stores deferred by two ranges (the product's emission), coarser read / write phases per wave.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/ceiling_probe2.hip -o scripts/bin/ceiling_probe2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: load next range, store this one (hbm_ceilings.hip)   MODE 1: write only   MODE 2: stores deferred by two ranges
// MODE 3: per wave four tiles read, then four tiles written (ranges of 16 tiles)
template <int MODE, bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256, 4) void k(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, size_t ntiles,
                                            u32x4 *sink) {
    constexpr unsigned WQ = 200, ROUNDS = 4;
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
        u32x4 *o = reinterpret_cast<u32x4 *>(out + (tt < ntiles ? tt : ntiles - 1) * (size_t)WQ * 16u);
#pragma unroll
        for (unsigned r = 0; r < ROUNDS; r++) {
            const unsigned q = lane + 64u * r;
            if (r + 1 < ROUNDS || q < WQ) {
                if (NT_STORE) __builtin_nontemporal_store(src[r & 3u], o + q);
                else o[q] = src[r & 3u];
            }
        }
    };
    if (MODE == 1) {
        const u32x4 c[4] = {{lane, 1, 2, 3}, {lane, 5, 6, 7}, {lane, 9, 10, 11}, {lane, 13, 14, 15}};
        for (size_t R = blockIdx.x; R * 8 < ntiles; R += G) {
            store_tile(R * 8 + w, c);
            store_tile(R * 8 + 4 + w, c);
        }
    } else if (MODE == 0) {
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
                for (int kq = 0; kq < 4; kq++) v[j][kq] = nx[j][kq];
        }
    } else if (MODE == 2) {
        // ring of three ranges in registers: store the range loaded two rounds ago
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
    } else {
        // MODE 3: ranges of 16 tiles, wave w tiles 16R + w + 4j; four tiles read, then four tiles written
        u32x4 v[4][4], nx[4][4];
        size_t R = blockIdx.x;
#pragma unroll
        for (int j = 0; j < 4; j++) load_tile(R * 16 + 4 * j + w, v[j]);
        for (; R * 16 < ntiles; R += G) {
#pragma unroll
            for (int j = 0; j < 4; j++) load_tile((R + G) * 16 + 4 * j + w, nx[j]);
#pragma unroll
            for (int j = 0; j < 4; j++) store_tile(R * 16 + 4 * j + w, v[j]);
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int kq = 0; kq < 4; kq++) v[j][kq] = nx[j][kq];
        }
    }
}

int main() {
    const size_t n = 1ull << 30, ntiles = n / 4096;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned char *a, *b; u32x4 *sink;
    CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, n));
    auto run = [&](const char *name, double bytes, auto kernel) {
        auto launch = [&] { hipLaunchKernelGGL(kernel, dim3(1024), dim3(256), 0, 0, a, b, ntiles, sink); };
        for (int it = 0; it < 300; it++) launch();
        CK(hipEventRecord(e0));
        for (int it = 0; it < 300; it++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float tt; CK(hipEventElapsedTime(&tt, e0, e1));
        printf("%-64s %.4f ms %7.1f GB/s\n", name, tt / 300, bytes / (tt / 300) * 1e-6);
        fflush(stdout);
    };
    const double W = (double)ntiles * 3200.0, N = (double)n;
    for (int rep = 0; rep < 2; rep++) {
        run("write only 0.78 N, nt stores", W, k<1, false, true>);
        run("write only 0.78 N, plain stores", W, k<1, false, false>);
        run("mix 1:0.78 load next / store this (plain ld, nt st)", N + W, k<0, false, true>);
        run("mix 1:0.78 stores deferred two ranges (plain ld, nt st)", N + W, k<2, false, true>);
        run("mix 1:0.78 stores deferred two ranges (nt ld, nt st)", N + W, k<2, true, true>);
        run("mix 1:0.78 four tiles read, four written (plain ld, nt st)", N + W, k<3, false, true>);
        run("mix 1:0.78 four tiles read, four written (nt ld, nt st)", N + W, k<3, true, true>);
    }
    return 0;
}
