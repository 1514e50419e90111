// Measured HBM3E bandwidth of one MI355X with trivial kernels (the 60 %
// target of north_star is against the MEASURED read bandwidth, not the 8 TB/s vendor figure).
//   read : every byte of a buffer is loaded once (16 B per lane per load), XOR-folded
//   copy : read N, write N
//   mix  : read N, write 0.775 N  -- the stage-1 kernel's own read:write ratio on the minified workload
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/hbm_bw.hip -o gpurun_out/hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ in, size_t n16, u32x4 *sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(in + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    for (; i < n16; i += stride) { u32x4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;  // never true in practice
}

// read n16 quads; write the first (num/den) of every block's quads
__global__ __launch_bounds__(256) void k_mix(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n16,
                                             unsigned num, unsigned den) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        u32x4 v = __builtin_nontemporal_load(in + i);
        const size_t blk = i / den;  // den consecutive quads -> num quads out
        const unsigned r = (unsigned)(i % den);
        if (r < num) __builtin_nontemporal_store(v, out + blk * num + r);
    }
}

// the same with plain (temporal) loads and stores / plain loads and non-temporal stores
template <bool kNtLoad, bool kNtStore>
__global__ __launch_bounds__(256) void k_mix_t(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n16,
                                               unsigned num, unsigned den) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        u32x4 v = kNtLoad ? __builtin_nontemporal_load(in + i) : in[i];
        const size_t blk = i / den;
        const unsigned r = (unsigned)(i % den);
        if (r < num) {
            if (kNtStore) __builtin_nontemporal_store(v, out + blk * num + r);
            else out[blk * num + r] = v;
        }
    }
}

// read n16 quads; write every quad `mult` times (mult coalesced output streams): the write-heavy mixes of the
// dense extremes of config 4 (4 bytes out per structural: d = 0.5 -> 2 bytes out per byte in, d = 1 -> 4)
__global__ __launch_bounds__(256) void k_expand(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n16, unsigned mult) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        u32x4 v = __builtin_nontemporal_load(in + i);
        for (unsigned m = 0; m < mult; m++) __builtin_nontemporal_store(v, out + (size_t)m * n16 + i);
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t gib : {1, 4}) {
        const size_t n = gib << 30, n16 = n / 16;
        u32x4 *a, *b, *sink; CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, n));
        auto timeit = [&](const char *name, double bytes, auto launch) {
            std::vector<float> ms;
            for (int it = 0; it < 13; it++) {
                CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 3) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("%zu GiB %-28s median %.4f ms  %.1f GB/s (best %.1f)\n", gib, name, ms[ms.size() / 2],
                   bytes / ms[ms.size() / 2] * 1e-6, bytes / ms[0] * 1e-6);
        };
        for (int wgs : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
            char nm[64];
            snprintf(nm, sizeof nm, "read  u4 grid %d", wgs);
            timeit(nm, (double)n, [&] { hipLaunchKernelGGL(k_read<4>, dim3(wgs), dim3(256), 0, 0, a, n16, sink); });
            snprintf(nm, sizeof nm, "read  u8 grid %d", wgs);
            timeit(nm, (double)n, [&] { hipLaunchKernelGGL(k_read<8>, dim3(wgs), dim3(256), 0, 0, a, n16, sink); });
        }
        for (int wgs : {256 * 8, 256 * 32}) {
            char nm[64];
            snprintf(nm, sizeof nm, "copy  grid %d (2N bytes)", wgs);
            timeit(nm, 2.0 * n, [&] { hipLaunchKernelGGL(k_mix, dim3(wgs), dim3(256), 0, 0, a, b, n16, 1u, 1u); });
            snprintf(nm, sizeof nm, "mix 31/40 grid %d (1.775N)", wgs);
            timeit(nm, 1.775 * n, [&] { hipLaunchKernelGGL(k_mix, dim3(wgs), dim3(256), 0, 0, a, b, n16, 31u, 40u); });
        }
        for (int rep = 0; rep < 2; rep++) {
            timeit("mix 31/40 g8192 plain ld, plain st", 1.775 * n, [&] { hipLaunchKernelGGL((k_mix_t<false, false>), dim3(8192), dim3(256), 0, 0, a, b, n16, 31u, 40u); });
            timeit("mix 31/40 g8192 plain ld, nt st", 1.775 * n, [&] { hipLaunchKernelGGL((k_mix_t<false, true>), dim3(8192), dim3(256), 0, 0, a, b, n16, 31u, 40u); });
            timeit("mix 31/40 g8192 nt ld, plain st", 1.775 * n, [&] { hipLaunchKernelGGL((k_mix_t<true, false>), dim3(8192), dim3(256), 0, 0, a, b, n16, 31u, 40u); });
            timeit("mix 31/40 g8192 nt ld, nt st", 1.775 * n, [&] { hipLaunchKernelGGL((k_mix_t<true, true>), dim3(8192), dim3(256), 0, 0, a, b, n16, 31u, 40u); });
        }
        timeit("hipMemcpyDtoD (2N bytes)", 2.0 * n, [&] { CK(hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, 0)); });
        if (gib == 1) {
            u32x4 *big; CK(hipMalloc(&big, 4 * n));
            for (unsigned mult : {2u, 4u}) {
                char nm[64];
                snprintf(nm, sizeof nm, "read N, write %uN grid 8192", mult);
                timeit(nm, (1.0 + mult) * n, [&] { hipLaunchKernelGGL(k_expand, dim3(8192), dim3(256), 0, 0, a, big, n16, mult); });
            }
            CK(hipFree(big));
        }
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
    }
    return 0;
}
