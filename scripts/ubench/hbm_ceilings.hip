// HBM ceilings for the stage-1 kernel, measured the way the kernel uses the memory system (the
// round-1 figures in profiles/traffic.json came from a grid-stride kernel with per-lane-strided stores, and the product
// kernel beat its own "same-mix peak").  Here: the kernel's PERSISTENT grid (1024 workgroups x 256 threads, 4 per CU),
// every wave walks 4 KiB tiles; a tile is read as four 1 KiB-contiguous wave loads (the kernel's coalesced shape, plain
// or non-temporal) and `w` bytes per tile leave as whole 128-byte lines of non-temporal 16-byte stores at the tile's
// dense output position (the kernel's copy_out).  No computation between load and store: nothing can move these bytes
// faster with this grid, so every product kernel must sit at or below its row.
//   read           w = 0            (a sparse input: spaces + one scalar)
//   mix 1:0.41     w = 1664 B/tile  (UTF-8-heavy d = 0.104, pretty-printed d = 0.097: 4 d = 0.39 .. 0.42)
//   mix 1:0.78     w = 3200 B/tile  (minified d = 0.193: 4 d = 0.775)
//   copy 1:1, 1:1.6 (d = 0.4), 1:2 (d = 0.5), 1:2.67 (d = 0.667), 1:4 (d = 1.0)
// Timed two ways: 13 single launches between their own events (median), and 400 back-to-back launches behind 400
// untimed ones (the settled clock bench.py's `value` is taken at).  Prints one machine-readable line "CEILINGS {...}"
// that scripts/ceilings_update.py writes into profiles/traffic.json.
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/hbm_ceilings.hip -o scripts/bin/hbm_ceilings
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// The number of store ROUNDS per tile (64 quads = 1 KiB each; the last one may be partial: `wquads` at run time) is a
// template constant: the stores are unrolled and counted, so that waiting for the NEXT range's loads (issued in front
// of them) is `s_waitcnt vmcnt(<stores>)` and the stores stay in flight.  (With a run-time count the compiler must wait
// for vmcnt(0) -- loads and stores share the counter on gfx9 -- and the wave idles until its stores have landed: that
// version of this kernel ran 7 % BELOW the product kernel.)
// The walk is the product's: a workgroup takes RANGES of 8 consecutive tiles (round robin over the grid), wave w the
// tiles 8R + w and 8R + 4 + w; both tiles of the next range are requested before the two of this range are stored
// (8 KiB of loads in flight per wave, 128 KiB per CU: one tile per wave does not cover the HBM latency).
// MODE 0  load the next range, store this one
// MODE 1  write only (no loads): what the same output costs alone
// MODE 2  stores deferred by two ranges like the product's emission (three ranges of bytes in registers)
// Measured while looking for what the product does better (profiles/r04/ceiling_probe*.txt): dummy vector work between
// arrival and stores, 8 / 12 / 32 waves per CU, the workgroup -> range order (fixed XCD residue, rotating, hashed),
// four tiles read then four written per wave: all within -4 .. +2 % of MODE 0; MODE 2 with non-temporal loads is the
// best trivial mix (+4 %).
template <int MODE, bool NT_LOAD, bool NT_STORE, unsigned ROUNDS>
__global__ __launch_bounds__(256, 4) void k_walk(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                                 size_t ntiles, unsigned wquads, u32x4 *sink) {
    const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const size_t G = gridDim.x;
    auto load_tile = [&](size_t tt, u32x4 (&dst)[4]) {
        const unsigned char *base = in + (tt < ntiles ? tt : ntiles - 1) * 4096;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32x4 *p = reinterpret_cast<const u32x4 *>(base + 16u * lane + 1024u * k);
            dst[k] = NT_LOAD ? __builtin_nontemporal_load(p) : *p;
        }
    };
    auto store_tile = [&](size_t tt, const u32x4 (&src)[4]) {
        u32x4 *o = reinterpret_cast<u32x4 *>(out + (tt < ntiles ? tt : ntiles - 1) * (size_t)wquads * 16u);
        // every byte that was loaded goes into what is stored: a tile that writes fewer than four rounds would
        // otherwise leave some of its four loads dead, and the compiler removes dead loads (the first version of this
        // file "read" 1 GiB at 25 TB/s that way)
        const u32x4 f = src[0] ^ src[1] ^ src[2] ^ src[3];
#pragma unroll
        for (unsigned r = 0; r < ROUNDS; r++) {
            const unsigned q = lane + 64u * r;
            if (r + 1 < ROUNDS || q < wquads) {  // only the last round is partial
                if (NT_STORE) __builtin_nontemporal_store(f, o + q);
                else o[q] = f;
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
        u32x4 acc = {0u, 0u, 0u, 0u};
        u32x4 v[2][4], nx[2][4];
        size_t R = blockIdx.x;
        load_tile(R * 8 + w, v[0]);
        load_tile(R * 8 + 4 + w, v[1]);
        for (; R * 8 < ntiles; R += G) {
            load_tile((R + G) * 8 + w, nx[0]);
            load_tile((R + G) * 8 + 4 + w, nx[1]);
            if (ROUNDS) {
                store_tile(R * 8 + w, v[0]);
                store_tile(R * 8 + 4 + w, v[1]);
            } else {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int k = 0; k < 4; k++) acc ^= v[j][k];
            }
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int k = 0; k < 4; k++) v[j][k] = nx[j][k];
        }
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc;  // never true in practice
    } else {
        // ring of three ranges in registers: a range is stored two rounds after it was requested
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

template <int MODE, bool NT_LOAD, bool NT_STORE>
static void launch_walk(int grid, hipStream_t stream, const unsigned char *in, unsigned char *out, size_t ntiles, unsigned wquads,
                        u32x4 *sink) {
    const unsigned rounds = (wquads + 63u) / 64u;
#define MSJ_CASE(R)                                                                                                                \
    case R:                                                                                                                        \
        hipLaunchKernelGGL((k_walk<MODE, NT_LOAD, NT_STORE, R>), dim3(grid), dim3(256), 0, stream, in, out, ntiles, wquads, sink); \
        break;
    switch (rounds) {
        MSJ_CASE(0) MSJ_CASE(1) MSJ_CASE(2) MSJ_CASE(3) MSJ_CASE(4) MSJ_CASE(5) MSJ_CASE(6) MSJ_CASE(7) MSJ_CASE(8)
        MSJ_CASE(9) MSJ_CASE(10) MSJ_CASE(11) MSJ_CASE(12) MSJ_CASE(13) MSJ_CASE(14) MSJ_CASE(15) MSJ_CASE(16)
        default: break;
    }
#undef MSJ_CASE
}

// The same kernels behind a C entry point, for bench.py: the ceilings are taken on the SAME box in the SAME process as
// the product's number (boxes of the pool differ by 3-5 %, so a ceiling recorded on another box can sit below the
// product).  Enqueues `reps` launches on `stream`; the caller times them with events.  wquads <= 1024, out must hold
// ntiles * wquads * 16 bytes, sink 64 bytes.  policy: bit 0 non-temporal loads, bit 1 plain (instead of nt) stores,
// bits 2..3 the mode (0 load next / store this, 1 write only, 2 stores deferred by two ranges).
extern "C" int msj_ceiling_launch(const void *d_in, void *d_out, void *d_sink, uint64_t ntiles, uint32_t wquads, uint32_t policy,
                                  uint32_t grid, uint32_t reps, void *stream) {
    if (!d_in || !d_sink || ntiles == 0 || wquads > 1024u || (wquads && !d_out) || grid == 0 || (policy >> 2) > 2u) return -1;
    if ((policy >> 2) != 0u && wquads == 0u) return -1;
    const unsigned char *in = static_cast<const unsigned char *>(d_in);
    unsigned char *out = static_cast<unsigned char *>(d_out);
    hipStream_t s = static_cast<hipStream_t>(stream);
    u32x4 *sink = static_cast<u32x4 *>(d_sink);
    for (uint32_t i = 0; i < reps; i++) {
        switch (policy & 15u) {
            case 0: launch_walk<0, false, true>((int)grid, s, in, out, ntiles, wquads, sink); break;
            case 1: launch_walk<0, true, true>((int)grid, s, in, out, ntiles, wquads, sink); break;
            case 2: launch_walk<0, false, false>((int)grid, s, in, out, ntiles, wquads, sink); break;
            case 3: launch_walk<0, true, false>((int)grid, s, in, out, ntiles, wquads, sink); break;
            case 4: case 5: launch_walk<1, false, true>((int)grid, s, in, out, ntiles, wquads, sink); break;
            case 6: case 7: launch_walk<1, false, false>((int)grid, s, in, out, ntiles, wquads, sink); break;
            case 8: launch_walk<2, false, true>((int)grid, s, in, out, ntiles, wquads, sink); break;
            case 9: launch_walk<2, true, true>((int)grid, s, in, out, ntiles, wquads, sink); break;
            default: return -1;
        }
    }
    return (int)hipGetLastError();
}

#ifndef MSJ_CEILING_LIBRARY
int main(int argc, char **argv) {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 1;
    const size_t n = gib << 30, ntiles = n / 4096;
    printf("device %s (%s), %d CUs; persistent grid %d x 256 threads; %zu GiB read per launch\n", p.name, p.gcnArchName,
           p.multiProcessorCount, p.multiProcessorCount * 4, gib);
    const int grid = p.multiProcessorCount * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned char *a, *b; u32x4 *sink;
    CK(hipMalloc(&a, n)); CK(hipMalloc(&b, 4 * n + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, n)); CK(hipMemset(b, 2, 4 * n));
    std::string json = "{";
    auto run = [&](const char *key, const char *name, unsigned wquads, unsigned policy) {
        const double bytes = (((policy >> 2) == 1u) ? 0.0 : (double)n) + (double)ntiles * wquads * 16.0;
        auto launch = [&] { msj_ceiling_launch(a, b, sink, ntiles, wquads, policy, (uint32_t)grid, 1, nullptr); };
        std::vector<float> ms;
        for (int it = 0; it < 16; it++) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 3) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const int reps = gib > 1 ? 100 : 400;
        for (int it = 0; it < reps; it++) launch();
        CK(hipEventRecord(e0));
        for (int it = 0; it < reps; it++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float tt; CK(hipEventElapsedTime(&tt, e0, e1));
        const double single = bytes / ms[ms.size() / 2] * 1e-6, settled = bytes / (tt / reps) * 1e-6;
        printf("%-44s r:w 1:%.3f  single launch median %.4f ms %7.1f GB/s | %d back to back %.4f ms %7.1f GB/s\n", name,
               wquads * 16.0 / 4096.0, ms[ms.size() / 2], single, reps, tt / reps, settled);
        fflush(stdout);
        char buf[256];
        snprintf(buf, sizeof buf, "%s\"%s\": {\"w_per_r\": %.4f, \"single_launch\": %.1f, \"settled\": %.1f}", json.size() > 1 ? ", " : "",
                 key, wquads * 16.0 / 4096.0, single, settled);
        json += buf;
    };
    for (int rep = 0; rep < 2; rep++) {
        const std::string sfx = rep ? "_again" : "";
        run(("read_plain" + sfx).c_str(), "read, plain loads", 0, 0);
        run(("read_nt" + sfx).c_str(), "read, non-temporal loads", 0, 1);
        run(("mix041_plain_nt" + sfx).c_str(), "mix 1:0.41 plain loads, nt whole-line stores", 104, 0);
        run(("mix041_nt_nt" + sfx).c_str(), "mix 1:0.41 nt loads, nt whole-line stores", 104, 1);
        run(("mix078_plain_nt" + sfx).c_str(), "mix 1:0.78 plain loads, nt whole-line stores", 200, 0);
        run(("mix078_nt_nt" + sfx).c_str(), "mix 1:0.78 nt loads, nt whole-line stores", 200, 1);
        run(("mix078_plain_plain" + sfx).c_str(), "mix 1:0.78 plain loads, plain stores", 200, 2);
        run(("mix041_deferred_nt_nt" + sfx).c_str(), "mix 1:0.41 stores deferred two ranges, nt loads, nt stores", 104, 9);
        run(("mix078_deferred_plain_nt" + sfx).c_str(), "mix 1:0.78 stores deferred two ranges, plain loads, nt stores", 200, 8);
        run(("mix078_deferred_nt_nt" + sfx).c_str(), "mix 1:0.78 stores deferred two ranges, nt loads, nt stores", 200, 9);
        run(("write041_nt" + sfx).c_str(), "write only 0.41 N, nt stores", 104, 4);
        run(("write078_nt" + sfx).c_str(), "write only 0.78 N, nt stores", 200, 4);
        run(("write078_plain" + sfx).c_str(), "write only 0.78 N, plain stores", 200, 6);
        run(("write2_nt" + sfx).c_str(), "write only 2 N, nt stores", 512, 4);
        run(("write2_plain" + sfx).c_str(), "write only 2 N, plain stores", 512, 6);
        run(("write4_plain" + sfx).c_str(), "write only 4 N, plain stores", 1024, 6);
        run(("mix2_deferred_nt_nt" + sfx).c_str(), "mix 1:2 stores deferred two ranges, nt loads, nt stores", 512, 9);
        run(("copy_plain_nt" + sfx).c_str(), "copy 1:1 plain loads, nt stores", 256, 0);
        run(("copy_nt_nt" + sfx).c_str(), "copy 1:1 nt loads, nt stores", 256, 1);
        run(("mix16_plain_nt" + sfx).c_str(), "mix 1:1.6 (d = 0.4) plain loads, nt stores", 408, 0);
        run(("mix2_plain_nt" + sfx).c_str(), "mix 1:2 (d = 0.5) plain loads, nt stores", 512, 0);
        run(("mix2_plain_plain" + sfx).c_str(), "mix 1:2 (d = 0.5) plain loads, plain stores", 512, 2);
        run(("mix267_plain_nt" + sfx).c_str(), "mix 1:2.67 (d = 0.667) plain loads, nt stores", 680, 0);
        run(("mix4_plain_nt" + sfx).c_str(), "mix 1:4 (d = 1.0) plain loads, nt stores", 1024, 0);
        run(("mix4_plain_plain" + sfx).c_str(), "mix 1:4 (d = 1.0) plain loads, plain stores", 1024, 2);
    }
    json += "}";
    printf("CEILINGS {\"gib\": %zu, \"grid\": %d, \"device\": \"%s\", \"gbps\": %s}\n", gib, grid, p.gcnArchName, json.c_str());
    return 0;
}
#endif  // MSJ_CEILING_LIBRARY
