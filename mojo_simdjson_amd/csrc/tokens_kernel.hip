// tokens_kernel.hip -- token-stream pre-pass for stage 2.
//
// What the reference's stage 2 recomputes one structural at a time:
//   * JsonIterator.advance / peek / last_structural dereference buf[structural_indexes[i]]
//     (src/mojo_simdjson/generic/stage2/json_iterator.mojo:256-288) -- here one coalesced
//     array type[i] = buf[idx[i]];
//   * walk_document keeps a running container depth, +1 at '{' '[' and -1 at '}' ']'
//     (json_iterator.mojo:84-90,173-180 and the scope_end state) -- here depth[i], a prefix sum
//     over the type bytes: the nesting depth of token i (a bracket has the depth of the
//     container it sits in, so an opening bracket and its closing bracket carry the same value
//     and everything between them is deeper), plus the minimum / maximum / final running depth,
//     which is what decides underflow, DEPTH_ERROR and "document not closed".
// DERIVED quantities: the reference has no array like this and no fixture for it; the CPU
// definition used by the tests is a definition, not a pin against the reference's own outputs.
//
// Three small kernels over blocks of 2 048 structurals: (1) gather the type bytes and reduce each
// block to (sum, min prefix, max prefix) of its depth deltas, (2) one workgroup scans the block
// aggregates in order, (3) every block re-scans its type bytes from the exact depth at its
// start.  HBM-bound: 4 B index + the gathered byte + 1 B type out, then 1 B type in + 4 B depth
// out per structural.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msj_stage1.h"

namespace msj_tokens {

constexpr int kThreads = 256;
constexpr int kPer = 8;                        // structurals per thread
constexpr uint32_t kBlock = kThreads * kPer;   // per workgroup

struct Agg {
    int32_t sum, mn, mx;  // total delta; min / max of the running sum after each token (relative)
};
constexpr int32_t kNone = 0x7FFFFFFF;  // mn == kNone / mx == -kNone: no token in this aggregate
__device__ __forceinline__ Agg combine(const Agg &a, const Agg &b) {
    Agg r;
    r.sum = a.sum + b.sum;
    r.mn = (b.mn == kNone) ? a.mn : min(a.mn, a.sum + b.mn);
    r.mx = (b.mx == -kNone) ? a.mx : max(a.mx, a.sum + b.mx);
    return r;
}
__device__ __forceinline__ int delta_of(uint32_t c) {
    return (c == '{' || c == '[') ? 1 : ((c == '}' || c == ']') ? -1 : 0);
}
__device__ __forceinline__ Agg shfl_up(const Agg &a, int off) {
    Agg r;
    r.sum = __shfl_up(a.sum, off);
    r.mn = __shfl_up(a.mn, off);
    r.mx = __shfl_up(a.mx, off);
    return r;
}

// (1) type bytes + per-block aggregate
__global__ __launch_bounds__(kThreads) void gather_reduce(const uint8_t *__restrict__ buf, const uint32_t *__restrict__ idx,
                                                          uint64_t n, uint8_t *__restrict__ type, int32_t *__restrict__ block_agg) {
    __shared__ Agg wave_agg[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kBlock + (uint64_t)threadIdx.x * kPer;
    uint32_t off[kPer];
    if (base + kPer <= n) {  // 32 bytes of indices per thread, two 16-byte loads
        const uint4 a = *reinterpret_cast<const uint4 *>(idx + base);
        const uint4 b = *reinterpret_cast<const uint4 *>(idx + base + 4);
        off[0] = a.x; off[1] = a.y; off[2] = a.z; off[3] = a.w;
        off[4] = b.x; off[5] = b.y; off[6] = b.z; off[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < kPer; k++) off[k] = (base + k < n) ? idx[base + k] : 0xFFFFFFFFu;
    }
    uint32_t c[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) c[k] = (base + k < n) ? buf[off[k]] : (uint32_t)' ';
    if (base + kPer <= n) {
        *reinterpret_cast<uint2 *>(type + base) =
            make_uint2(c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24), c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24));
    } else {
#pragma unroll
        for (int k = 0; k < kPer; k++)
            if (base + k < n) type[base + k] = (uint8_t)c[k];
    }
    Agg a = {0, kNone, -kNone};
    int run = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        if (base + k < n) {
            run += delta_of(c[k]);
            a.mn = min(a.mn, run);
            a.mx = max(a.mx, run);
        }
    }
    a.sum = run;
    // ordered reduction: inclusive scan inside the wave, the last lane holds the wave's aggregate
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const Agg p = shfl_up(a, o);
        if (lane >= o) a = combine(p, a);
    }
    if (lane == 63) wave_agg[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        Agg t = wave_agg[0];
#pragma unroll
        for (int w = 1; w < kThreads / 64; w++) t = combine(t, wave_agg[w]);
        block_agg[3 * (uint64_t)blockIdx.x + 0] = t.sum;
        block_agg[3 * (uint64_t)blockIdx.x + 1] = t.mn;
        block_agg[3 * (uint64_t)blockIdx.x + 2] = t.mx;
    }
}

// (2) one workgroup: exclusive scan of the block sums, global min / max / final depth.  Each
//     thread folds kScanPer consecutive block aggregates serially (so a pass covers 8 192 blocks).
constexpr int kScanPer = 8;
__global__ __launch_bounds__(1024) void scan_blocks(const int32_t *__restrict__ block_agg, uint32_t nblocks, int32_t *__restrict__ block_start,
                                                    msj_tokens_result *__restrict__ result, uint64_t n) {
    __shared__ Agg wave_agg[16];
    __shared__ Agg carry;
    if (threadIdx.x == 0) carry = Agg{0, kNone, -kNone};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024 * kScanPer) {
        const uint32_t first = b0 + threadIdx.x * kScanPer;
        Agg own[kScanPer];
        Agg a = {0, kNone, -kNone};
#pragma unroll
        for (int k = 0; k < kScanPer; k++) {
            const uint32_t b = first + k;
            own[k] = Agg{0, kNone, -kNone};
            if (b < nblocks) own[k] = Agg{block_agg[3 * (uint64_t)b], block_agg[3 * (uint64_t)b + 1], block_agg[3 * (uint64_t)b + 2]};
            a = combine(a, own[k]);
        }
        const Agg mine = a;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const Agg p = shfl_up(a, o);
            if (lane >= o) a = combine(p, a);
        }
        if (lane == 63) wave_agg[wave] = a;
        __syncthreads();
        Agg before = carry;  // everything in front of this wave
        for (int w = 0; w < wave; w++) before = combine(before, wave_agg[w]);
        const Agg incl = combine(before, a);
        int32_t run = incl.sum - mine.sum;  // depth at this thread's first block
#pragma unroll
        for (int k = 0; k < kScanPer; k++) {
            if (first + k < nblocks) block_start[first + k] = run;
            run += own[k].sum;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = incl;  // the last thread's inclusive value covers the whole pass
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        result->n = n;
        result->final_depth = carry.sum;
        result->min_depth = n ? carry.mn : 0;
        result->max_depth = n ? carry.mx : 0;
        result->reserved = 0;
    }
}

// (3) depth of every token
__global__ __launch_bounds__(kThreads) void apply_depth(const uint8_t *__restrict__ type, uint64_t n,
                                                        const int32_t *__restrict__ block_start, int32_t *__restrict__ depth) {
    __shared__ int wave_sum[kThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kBlock + (uint64_t)threadIdx.x * kPer;
    uint32_t c[kPer];
    if (base + kPer <= n) {
        const uint2 t = *reinterpret_cast<const uint2 *>(type + base);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c[k] = (t.x >> (8 * k)) & 0xFFu;
            c[4 + k] = (t.y >> (8 * k)) & 0xFFu;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPer; k++) c[k] = (base + k < n) ? type[base + k] : (uint32_t)' ';
    }
    int d[kPer], run = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        d[k] = delta_of(c[k]);
        run += d[k];
    }
    // exclusive prefix of the thread sums inside the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int p = __shfl_up(incl, o);
        if (lane >= o) incl += p;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    int before = block_start[blockIdx.x] + incl - run;
    for (int w = 0; w < wave; w++) before += wave_sum[w];
    int out[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        out[k] = before - (d[k] < 0 ? 1 : 0);  // a closing bracket sits at the depth of its container
        before += d[k];
    }
    if (base + kPer <= n) {
        *reinterpret_cast<int4 *>(depth + base) = make_int4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<int4 *>(depth + base + 4) = make_int4(out[4], out[5], out[6], out[7]);
    } else {
#pragma unroll
        for (int k = 0; k < kPer; k++)
            if (base + k < n) depth[base + k] = out[k];
    }
}

}  // namespace msj_tokens

// workspace: 3 int32 per block (aggregates) + 1 int32 per block (start depth)
extern "C" uint64_t msj_tokens_workspace_bytes(uint64_t n) {
    const uint64_t nb = (n + msj_tokens::kBlock - 1) / msj_tokens::kBlock;
    return (nb ? nb : 1) * 4 * sizeof(int32_t);
}

extern "C" int msj_launch_tokens(const uint8_t *d_buf, const uint32_t *d_idx, uint64_t n, uint8_t *d_type, int32_t *d_depth,
                                 msj_tokens_result *d_result, int32_t *d_ws, void *stream) {
    using namespace msj_tokens;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t nb64 = (n + kBlock - 1) / kBlock;
    if (nb64 > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const uint32_t nb = (uint32_t)nb64;
    int32_t *agg = d_ws, *start = d_ws + 3 * (uint64_t)(nb ? nb : 1);
    if (nb) hipLaunchKernelGGL(gather_reduce, dim3(nb), dim3(kThreads), 0, s, d_buf, d_idx, n, d_type, agg);
    hipLaunchKernelGGL(scan_blocks, dim3(1), dim3(1024), 0, s, agg, nb, start, d_result, n);
    if (nb) hipLaunchKernelGGL(apply_depth, dim3(nb), dim3(kThreads), 0, s, d_type, n, start, d_depth);
    return (int)hipGetLastError();
}
