// Multi-document (NDJSON / concatenated documents) mode -- SURVEY.md section 8, row f3.
//
// The reference has no streaming mode: it left upstream simdjson's `partial` stage-1 modes out
// (generic/stage1/json_structural_indexer.mojo:153 "Is more complicated in the original
// implementation", :169 "used later in partial == stage1_mode::streaming_final";
// generic/stage2/tape_builder.mojo:25 "TODO: add streaming").  What upstream does there, one
// structural at a time and backwards from the end of a window (find_next_document_index), falls
// out of the token stream of row f1 in one parallel pass: a document starts at every token that
// sits at depth 0 and is not a closing bracket.
//
//   doc_first[k]  = index (into idx[] / type[] / depth[]) of the first token of document k
//   result        = number of documents that start in the window, how many of them are complete,
//                   how many leading tokens belong to complete documents, and the byte offset at
//                   which the next window has to start (the first byte of the cut document).
//
// The last document is complete when it is a container whose closing bracket -- a closing bracket
// at depth 0 -- comes after it; a string that is closed; any other scalar if the window is the end
// of the stream or ends in a blank (a number or a literal that touches the end of the window may go
// on in the next one -- upstream counts it as complete and documents the truncation as a limitation).
//
// Four launches: per-block (count, last start, last depth-0 closing bracket); a two-level scan of
// those (inside runs of 1 024 blocks in parallel, then one workgroup over the runs, which also
// settles the result); ordered compaction of the starts.
// DERIVED quantities: defined by the CPU statement the tests use, which they also check against a
// restatement of upstream's backward scan on well-formed streams.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msj_stage1.h"

namespace msj_docs {

constexpr int kThreads = 256;
constexpr int kPer = 8;                       // tokens per thread
constexpr uint32_t kBlock = kThreads * kPer;  // tokens per workgroup
static_assert(kBlock == 2048, "apply_depth (tokens_kernel.hip) writes the same aggregates with its own block size");

__device__ __forceinline__ bool is_open(uint32_t c) { return c == '{' || c == '['; }
__device__ __forceinline__ bool is_close(uint32_t c) { return c == '}' || c == ']'; }

// bit k of `starts`: token base+k starts a document; of `closes`: it is a closing bracket at depth 0
__device__ __forceinline__ void classify8(const uint8_t *__restrict__ type, const int32_t *__restrict__ depth, uint64_t n,
                                          uint64_t base, uint32_t &starts, uint32_t &closes) {
    starts = 0;
    closes = 0;
    if (base >= n) return;
    uint32_t t[kPer];
    int32_t d[kPer];
    if (base + kPer <= n) {
        const uint2 tw = *reinterpret_cast<const uint2 *>(type + base);
        const int4 d0 = *reinterpret_cast<const int4 *>(depth + base);
        const int4 d1 = *reinterpret_cast<const int4 *>(depth + base + 4);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            t[k] = (tw.x >> (8 * k)) & 0xFFu;
            t[4 + k] = (tw.y >> (8 * k)) & 0xFFu;
        }
        d[0] = d0.x, d[1] = d0.y, d[2] = d0.z, d[3] = d0.w;
        d[4] = d1.x, d[5] = d1.y, d[6] = d1.z, d[7] = d1.w;
    } else {
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const bool in = base + k < n;
            t[k] = in ? type[base + k] : (uint32_t)'}';
            d[k] = in ? depth[base + k] : 1;
        }
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        const bool zero = d[k] == 0, cl = is_close(t[k]);
        starts |= (uint32_t)(zero && !cl) << k;
        closes |= (uint32_t)(zero && cl) << k;
    }
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
    return v;
}

// (1) per block of 2 048 tokens: number of starts, last start + 1, last depth-0 closing bracket + 1
__global__ __launch_bounds__(kThreads) void doc_count(const uint8_t *__restrict__ type, const int32_t *__restrict__ depth, uint64_t n,
                                                      uint4 *__restrict__ block_agg) {
    __shared__ uint32_t w_cnt[kThreads / 64], w_start[kThreads / 64], w_close[kThreads / 64];
    const uint64_t base = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) * kPer;
    uint32_t starts, closes;
    classify8(type, depth, n, base, starts, closes);
    const uint32_t cnt = wave_sum(__popc(starts));
    const uint32_t ls = wave_max(starts ? (uint32_t)base + (32u - __clz(starts)) : 0u);  // index + 1 of the highest set bit
    const uint32_t lc = wave_max(closes ? (uint32_t)base + (32u - __clz(closes)) : 0u);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        w_cnt[wave] = cnt;
        w_start[wave] = ls;
        w_close[wave] = lc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0, s = 0, e = 0;
        for (int w = 0; w < kThreads / 64; w++) {
            c += w_cnt[w];
            s = max(s, w_start[w]);
            e = max(e, w_close[w]);
        }
        block_agg[blockIdx.x] = make_uint4(c, s, e, 0);
    }
}

// (2a) many workgroups: exclusive scan of the block counts INSIDE each run of kSuper blocks, and the run's
//      (count, last start, last closing bracket); the single workgroup of (2) then only sees the runs (it
//      took 0.44 ms alone over the 1.6 x 10^5 blocks of 1 GiB of NDJSON: nothing hides its memory latency)
constexpr uint32_t kSuper = 1024;  // blocks per run: 256 threads x 4
__global__ __launch_bounds__(256) void doc_scan_super(const uint4 *__restrict__ block_agg, uint32_t nblocks, uint32_t *__restrict__ rel_off,
                                                      uint4 *__restrict__ super_agg) {
    __shared__ uint32_t w_cnt[4], w_start[4], w_close[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * kSuper + threadIdx.x * 4u;
    uint32_t own[4], c = 0, ls = 0, lc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        own[k] = 0;
        if (first + k < nblocks) {
            const uint4 q = block_agg[first + k];
            own[k] = q.x;
            ls = max(ls, q.y);
            lc = max(lc, q.z);
        }
        c += own[k];
    }
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t p = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += p;
    }
    ls = wave_max(ls);
    lc = wave_max(lc);
    if (lane == 63) w_cnt[wave] = inc;
    if (lane == 0) {
        w_start[wave] = ls;
        w_close[wave] = lc;
    }
    __syncthreads();
    uint32_t run = inc - c;
    for (int w = 0; w < wave; w++) run += w_cnt[w];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (first + k < nblocks) rel_off[first + k] = run;
        run += own[k];
    }
    if (threadIdx.x == 0)
        super_agg[blockIdx.x] = make_uint4(w_cnt[0] + w_cnt[1] + w_cnt[2] + w_cnt[3], max(max(w_start[0], w_start[1]), max(w_start[2], w_start[3])),
                                           max(max(w_close[0], w_close[1]), max(w_close[2], w_close[3])), 0);
}

// (2) one workgroup: exclusive scan of the block counts; settles the result
__global__ __launch_bounds__(1024) void doc_scan(const uint4 *__restrict__ block_agg, uint32_t nblocks, uint32_t *__restrict__ block_off,
                                                 const uint8_t *__restrict__ type, const uint32_t *__restrict__ idx, uint64_t n,
                                                 const uint8_t *__restrict__ buf, uint64_t len, int is_final,
                                                 const msj_carry *__restrict__ carry,
                                                 msj_documents_result *__restrict__ result) {
    __shared__ uint32_t s_cnt[1024], s_start[16], s_close[16];
    const uint32_t per = (nblocks + 1023u) / 1024u;
    const uint32_t b0 = threadIdx.x * per, b1 = min(nblocks, b0 + per);
    uint32_t c = 0, ls = 0, lc = 0;
    for (uint32_t b = b0; b < b1; b++) {
        const uint4 q = block_agg[b];
        c += q.x;
        ls = max(ls, q.y);
        lc = max(lc, q.z);
    }
    // inclusive scan of c over the 1 024 threads: within the wave, then across the 16 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t p = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += p;
    }
    ls = wave_max(ls);
    lc = wave_max(lc);
    if (lane == 63) s_cnt[wave] = inc;
    if (lane == 0) {
        s_start[wave] = ls;
        s_close[wave] = lc;
    }
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += s_cnt[w];
    uint32_t run = before + inc - c;
    for (uint32_t b = b0; b < b1; b++) {
        block_off[b] = run;
        run += block_agg[b].x;
    }
    if (threadIdx.x == 0) {
        uint32_t total = 0, last_start = 0, last_close = 0;
        for (int w = 0; w < 16; w++) {
            total += s_cnt[w];
            last_start = max(last_start, s_start[w]);
            last_close = max(last_close, s_close[w]);
        }
        msj_documents_result r;
        r.n_documents = total;
        if (total == 0) {
            r.n_complete = 0;
            r.tokens_complete = n;
            r.resume_offset = len;
        } else {
            const uint32_t s = last_start - 1u;
            const bool open_string = carry != nullptr && carry->in_string != 0;
            const uint32_t t = type[s];
            bool complete;
            if (is_open(t)) {
                complete = last_close > last_start;
            } else if ((uint64_t)s + 1u < n) {
                complete = true;  // something follows the scalar (only stray closing brackets can)
            } else if (open_string) {
                complete = false;  // the window ends inside the string this token opens (or is glued to)
            } else if (t == '"' || is_final) {
                complete = true;
            } else {
                const uint32_t e = buf[len - 1];
                complete = e == 0x20u || e == 0x0Au || e == 0x0Du || e == 0x09u;
            }
            r.n_complete = total - (complete ? 0u : 1u);
            r.tokens_complete = complete ? n : s;
            r.resume_offset = complete ? len : idx[s];
        }
        *result = r;
    }
}

// (3) ordered compaction of the starts
__global__ __launch_bounds__(kThreads) void doc_write(const uint8_t *__restrict__ type, const int32_t *__restrict__ depth, uint64_t n,
                                                      const uint32_t *__restrict__ block_off, const uint32_t *__restrict__ super_off,
                                                      uint32_t *__restrict__ doc_first, uint64_t capacity) {
    __shared__ uint32_t w_cnt[kThreads / 64];
    const uint64_t base = ((uint64_t)blockIdx.x * kThreads + threadIdx.x) * kPer;
    uint32_t starts, closes;
    classify8(type, depth, n, base, starts, closes);
    const uint32_t c = __popc(starts);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t p = (uint32_t)__shfl_up((int)inc, o);
        if (lane >= o) inc += p;
    }
    if (lane == 63) w_cnt[wave] = inc;
    __syncthreads();
    uint64_t slot = super_off[blockIdx.x / kSuper] + block_off[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; w++) slot += w_cnt[w];
    while (starts) {
        const uint32_t k = __ffs(starts) - 1u;
        starts &= starts - 1u;
        if (slot < capacity) doc_first[slot] = (uint32_t)base + k;
        slot++;
    }
}

}  // namespace msj_docs

extern "C" uint64_t msj_documents_workspace_bytes(uint64_t n) {
    const uint64_t nb = (n + msj_docs::kBlock - 1) / msj_docs::kBlock;
    const uint64_t ns = (nb + msj_docs::kSuper - 1) / msj_docs::kSuper;
    return ((nb ? nb : 1) + (ns ? ns : 1)) * (sizeof(uint4) + sizeof(uint32_t)) + 64;
}

extern "C" int msj_launch_documents(const uint8_t *d_buf, uint64_t len, int is_final, const uint32_t *d_idx, uint64_t n,
                                    const uint8_t *d_type, const int32_t *d_depth, const msj_carry *d_carry, uint32_t *d_doc_first, uint64_t capacity,
                                    msj_documents_result *d_result, void *d_ws, const void *d_block_agg, void *stream) {
    using namespace msj_docs;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t nb64 = (n + kBlock - 1) / kBlock;
    if (nb64 > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const uint32_t nb = (uint32_t)nb64;
    const uint32_t ns = (nb + kSuper - 1u) / kSuper;
    // per block; d_block_agg: already written by the token pre-pass for exactly these arrays
    const uint4 *agg = d_block_agg ? static_cast<const uint4 *>(d_block_agg) : static_cast<const uint4 *>(d_ws);
    uint4 *super_agg = static_cast<uint4 *>(d_ws) + (nb ? nb : 1);  // per run of kSuper blocks
    uint32_t *off = reinterpret_cast<uint32_t *>(super_agg + (ns ? ns : 1));
    uint32_t *super_off = off + (nb ? nb : 1);
    if (nb && !d_block_agg) hipLaunchKernelGGL(doc_count, dim3(nb), dim3(kThreads), 0, s, d_type, d_depth, n, static_cast<uint4 *>(d_ws));
    if (nb) hipLaunchKernelGGL(doc_scan_super, dim3(ns), dim3(256), 0, s, agg, nb, off, super_agg);
    hipLaunchKernelGGL(doc_scan, dim3(1), dim3(1024), 0, s, super_agg, ns, super_off, d_type, d_idx, n, d_buf, len, is_final, d_carry, d_result);
    if (nb) hipLaunchKernelGGL(doc_write, dim3(nb), dim3(kThreads), 0, s, d_type, d_depth, n, off, super_off, d_doc_first, capacity);
    return (int)hipGetLastError();
}
