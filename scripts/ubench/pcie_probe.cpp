// What the host-pointer entry point can build on (one MI355X box): cost of pinning user memory in place,
// host memcpy bandwidth by thread count, pinned H2D / D2H bandwidth alone and both at once.
// build: hipcc -O2 scripts/ubench/pcie_probe.cpp -o gpurun_out/pcie_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 256u << 20, M = 208u << 20;
    char *in = (char *)aligned_alloc(4096, N), *out = (char *)aligned_alloc(4096, M);
    memset(in, 1, N); memset(out, 2, M);
    void *d_in, *d_out; CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_out, M));
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now(); CK(hipHostRegister(in, N, hipHostRegisterDefault)); double t1 = now();
        CK(hipHostRegister(out, M, hipHostRegisterDefault)); double t2 = now();
        hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
        double a = now(); CK(hipMemcpyAsync(d_in, in, N, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double b = now();
        CK(hipMemcpyAsync(out, d_out, M, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); double c = now();
        CK(hipMemcpyAsync(d_in, in, N, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(out, d_out, M, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double d = now();
        double u0 = now(); CK(hipHostUnregister(in)); CK(hipHostUnregister(out)); double u1 = now();
        printf("rep %d: register 256 MiB %.2f ms, 208 MiB %.2f ms, unregister both %.2f ms | registered H2D %.1f GB/s, D2H %.1f GB/s, both at once %.2f ms (%.1f GB/s of input)\n",
               rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (u1 - u0) * 1e3, N / (b - a) / 1e9, M / (c - b) / 1e9, (d - c) * 1e3, N / (d - c) / 1e9);
        CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2));
    }
    { double a = now(); CK(hipMemcpy(d_in, in, N, hipMemcpyHostToDevice)); double b = now(); CK(hipMemcpy(out, d_out, M, hipMemcpyDeviceToHost)); double c = now();
      printf("pageable: H2D %.1f GB/s, D2H %.1f GB/s\n", N / (b - a) / 1e9, M / (c - b) / 1e9); }
    char *pin; CK(hipHostMalloc((void **)&pin, N, hipHostMallocDefault));
    for (int th : {1, 2, 4, 8}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            double a = now();
            std::vector<std::thread> ts;
            for (int t = 0; t < th; t++) ts.emplace_back([=] { size_t lo = N / th * t, hi = t + 1 == th ? N : N / th * (t + 1); memcpy(pin + lo, in + lo, hi - lo); });
            for (auto &t : ts) t.join();
            double b = now(); if (b - a < best) best = b - a;
        }
        printf("memcpy pageable -> pinned, %d thread(s): %.1f GB/s\n", th, N / best / 1e9);
    }
    // reading pinned memory that a DMA has just written (the download side), by allocation flavour
    struct Flavour { const char *name; unsigned flags; bool reg; };
    for (Flavour f : {Flavour{"hipHostMallocDefault", hipHostMallocDefault, false}, Flavour{"hipHostMallocNonCoherent", hipHostMallocNonCoherent, false},
                      Flavour{"hipHostMallocCoherent", hipHostMallocCoherent, false}, Flavour{"aligned_alloc + hipHostRegister", 0, true}}) {
        char *p2 = nullptr;
        if (f.reg) { p2 = (char *)aligned_alloc(4096, M); memset(p2, 3, M); CK(hipHostRegister(p2, M, hipHostRegisterDefault)); }
        else CK(hipHostMalloc((void **)&p2, M, f.flags));
        for (int th : {1, 4}) {
            double best = 1e9, dma = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                double a0 = now(); CK(hipMemcpy(p2, d_out, M, hipMemcpyDeviceToHost)); double a1 = now(); if (a1 - a0 < dma) dma = a1 - a0;
                double a = now();
                std::vector<std::thread> ts;
                for (int t = 0; t < th; t++) ts.emplace_back([=] { size_t lo = M / th * t, hi = t + 1 == th ? M : M / th * (t + 1); memcpy(out + lo, p2 + lo, hi - lo); });
                for (auto &t : ts) t.join();
                double b = now(); if (b - a < best) best = b - a;
            }
            printf("%-34s D2H %.1f GB/s; memcpy pinned -> pageable, %d thread(s): %.1f GB/s\n", f.name, M / dma / 1e9, th, M / best / 1e9);
        }
    }
    return 0;
}
