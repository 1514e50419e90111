// ceiling_probe.hip -- the four probes of round 4 that looked for a trivial read + write kernel faster than the product
// (profiles/r04/ceiling_probe*.txt), in one file with a mode switch (round 5: four files before):
//   hipcc --offload-arch=gfx950 -O3 -DMSJ_PROBE=<1|2|3|4> scripts/ubench/ceiling_probe.hip -o scripts/bin/ceiling_probe<n>
//   1  dummy vector work between arrival and stores, waves per CU, stores deferred by one range   (ceiling_probe_spin_grid.txt)
//   2  the workgroup -> range order (fixed XCD residue, rotating, hashed), four tiles read then written (ceiling_probe_range_order.txt)
//   3  data-dependent (irregular) output sizes, stores deferred, write only                        (ceiling_probe_irregular_sizes.txt, _deferred_writeonly.txt)
//   4  3 .. 16 KiB of output per tile and the order of a tile's store rounds: the dip at exactly 8 KiB (ceiling_probe4_store_order.txt)
#ifndef MSJ_PROBE
#define MSJ_PROBE 1
#endif

#if MSJ_PROBE == 1  // ---------------------------------------------------------------- (was scripts/ubench/ceiling_probe.hip)
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
#endif  // MSJ_PROBE == 1

#if MSJ_PROBE == 2  // ---------------------------------------------------------------- (was scripts/ubench/ceiling_probe2.hip)
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
#endif  // MSJ_PROBE == 2

#if MSJ_PROBE == 3  // ---------------------------------------------------------------- (was scripts/ubench/ceiling_probe3.hip)
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
#endif  // MSJ_PROBE == 3

#if MSJ_PROBE == 4  // ---------------------------------------------------------------- (was scripts/ubench/ceiling_probe4.hip)
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
#endif  // MSJ_PROBE == 4
