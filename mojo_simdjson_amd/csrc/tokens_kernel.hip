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
// Kernels: (1) the type bytes and, per 512 structurals, the (sum, min prefix, max prefix) of their depth deltas, from
// the workgroup's stretch of the buffer staged in LDS (the span kernel further down, type-bytes-only instantiation),
// merged to blocks of 2 048; (2) a two-level scan of the block aggregates; (3) every block re-scans its type bytes
// from the exact depth at its start.  HBM-bound: the buffer + 4 B index in, 1 B type out, then 1 B type in + 4 B
// depth out per structural.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/msj_stage1.h"
#include "lane_math.h"
#include "token_math.h"
#include "tokens_launch.h"

#ifndef MSJ_SPAN_ABLATE
#define MSJ_SPAN_ABLATE 0  // diagnostic builds: 1 no token evaluation, 2 no depth aggregates, 3 no bit-plane transpose (wrong results)
#endif

namespace msj_tokens {

#ifdef MSJ_TILE_STAMPS
// diagnostic build only (scripts/tile_stamps.py, scripts/compact_stamps.py): 100 MHz real-time stamps, eight per wave of
// token_tiles / per wave and block of match_compact
__device__ unsigned long long *g_tile_stamps = nullptr;
#define MSJ_STAMP_AT(slot, k)                                                                             \
    do {                                                                                                  \
        if ((threadIdx.x & 63u) == 0 && g_tile_stamps) g_tile_stamps[(uint64_t)(slot) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define MSJ_STAMP_AT(slot, k) do {} while (0)
#endif
#define MSJ_TSTAMP(k) MSJ_STAMP_AT((uint64_t)blockIdx.x * kTgWaves + wave, k)

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

// minimum and maximum over the wave, by DPP: an inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8),
// then the last lane of a row into the next row (row_bcast:15, rows 1 and 3) and lane 31 into rows 2 and 3
// (row_bcast:31); lane 63 holds the result.  A lane without a source lane keeps its value (the instruction is
// off for it).  One instruction per step and value; a DPP read needs two wait states after the write of its
// source, which the other chain and an s_nop provide.
__device__ __forceinline__ void wave_min_max(int lo, int hi, int &mn_out, int &mx_out) {  // min over lo, max over hi
    int mn = lo, mx = hi;
    asm volatile(
        "s_nop 1\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(mn), "+v"(mx));
    mn_out = __builtin_amdgcn_readlane(mn, 63);
    mx_out = __builtin_amdgcn_readlane(mx, 63);
}

// inclusive prefix sum over the wave, the same six DPP steps
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
    return x;
}
// (1) type bytes + per-block aggregate: the span kernel's type-bytes-only instantiation (token_spans<true, false>
//     below: the workgroup's stretch of the buffer staged in LDS with coalesced loads, the bytes picked from there;
//     a kernel that gathered buf[idx[i]] per token through L2 took 0.6 ms per GiB minified where this takes 0.5) and
//     merge_chunk_counts.

// (2a) many workgroups: exclusive scan of the block aggregates INSIDE each run of kSuper blocks (relative
//      start depth and start slot per block) and the aggregate of the run.  One workgroup scanning all
//      the blocks alone took 0.1 - 0.26 ms for 10^5 blocks (every pass pays the full memory latency with
//      nothing else on the chip); this way the single workgroup of (2) only sees the runs.
constexpr uint32_t kSuper = 1024;  // blocks per run: 256 threads x 4
// The list of opening brackets left to the min tree is kept in kSurvivorShards separate lists, block b appending to
// list b mod kSurvivorShards with one returning atomic per wave that has any: ONE counter word sustains only ~85
// returning atomics per microsecond chip-wide (the stage-1 kernel's range tickets met the same wall), and 10^5 blocks
// on one word made the depth pass 1.26 ms per GiB minified instead of 0.3.  Counters 4 KiB apart; a list's capacity is
// what its blocks can hold at most.
constexpr uint32_t kSurvivorShards = 256;
constexpr uint32_t kSurvivorStride = 1024;  // words between two counters
__host__ __device__ inline uint64_t survivor_capacity(uint64_t nblocks) { return (uint64_t)(kThreads * kPer) * (nblocks / kSurvivorShards + 1u); }
__global__ __launch_bounds__(256) void scan_super(const int32_t *__restrict__ block_agg, uint32_t nblocks, int32_t *__restrict__ rel_start,
                                                  uint32_t *__restrict__ rel_open, int32_t *__restrict__ super_agg) {
    __shared__ Agg wave_agg[4];
    __shared__ uint32_t wave_opens[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * kSuper + threadIdx.x * 4u;
    Agg own[4];
    uint32_t own_opens[4];
    Agg a = {0, kNone, -kNone};
    uint32_t no = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        own[k] = Agg{0, kNone, -kNone};
        own_opens[k] = 0;
        if (first + k < nblocks) {
            const int4 q = *reinterpret_cast<const int4 *>(block_agg + 4 * (uint64_t)(first + k));
            own[k] = Agg{q.x, q.y, q.z};
            own_opens[k] = (uint32_t)q.w;
        }
        a = combine(a, own[k]);
        no += own_opens[k];
    }
    const Agg mine = a;
    const uint32_t mine_opens = no;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const Agg p = shfl_up(a, o);
        const uint32_t po = __shfl_up(no, o);
        if (lane >= o) {
            a = combine(p, a);
            no += po;
        }
    }
    if (lane == 63) {
        wave_agg[wave] = a;
        wave_opens[wave] = no;
    }
    __syncthreads();
    Agg before = {0, kNone, -kNone};
    uint32_t before_opens = 0;
    for (int w = 0; w < wave; w++) {
        before = combine(before, wave_agg[w]);
        before_opens += wave_opens[w];
    }
    const Agg incl = combine(before, a);
    const uint32_t incl_opens = before_opens + no;
    int32_t run = incl.sum - mine.sum;
    uint32_t ro = incl_opens - mine_opens;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (first + k < nblocks) {
            rel_start[first + k] = run;
            rel_open[first + k] = ro;
        }
        run += own[k].sum;
        ro += own_opens[k];
    }
    if (threadIdx.x == 255)
        *reinterpret_cast<int4 *>(super_agg + 4 * (uint64_t)blockIdx.x) = make_int4(incl.sum, incl.mn, incl.mx, (int)incl_opens);
}

// (2) one workgroup: exclusive scan of the block sums, global min / max / final depth.  Each
//     thread folds kScanPer consecutive block aggregates serially (so a pass covers 8 192 blocks).
constexpr int kScanPer = 8;
__global__ __launch_bounds__(1024) void scan_blocks(const int32_t *__restrict__ block_agg, uint32_t nblocks, int32_t *__restrict__ block_start,
                                                    uint32_t *__restrict__ open_start, msj_tokens_result *__restrict__ result, uint64_t n,
                                                    const msj_tokens_result *__restrict__ prev, uint32_t *__restrict__ survivors,
                                                    uint32_t *__restrict__ resid) {
    __shared__ Agg wave_agg[16];
    __shared__ uint32_t wave_opens[16];
    __shared__ Agg carry;
    __shared__ uint32_t carry_opens;
    if (threadIdx.x == 0) {
        carry = Agg{0, kNone, -kNone};
        carry_opens = 0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 1024 * kScanPer) {
        const uint32_t first = b0 + threadIdx.x * kScanPer;
        Agg own[kScanPer];
        uint32_t own_opens[kScanPer];
        Agg a = {0, kNone, -kNone};
        uint32_t no = 0;
#pragma unroll
        for (int k = 0; k < kScanPer; k++) {
            const uint32_t b = first + k;
            own[k] = Agg{0, kNone, -kNone};
            own_opens[k] = 0;
            if (b < nblocks) {
                const int4 q = *reinterpret_cast<const int4 *>(block_agg + 4 * (uint64_t)b);
                own[k] = Agg{q.x, q.y, q.z};
                own_opens[k] = (uint32_t)q.w;
            }
            a = combine(a, own[k]);
            no += own_opens[k];
        }
        const Agg mine = a;
        const uint32_t mine_opens = no;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const Agg p = shfl_up(a, o);
            const uint32_t po = __shfl_up(no, o);
            if (lane >= o) {
                a = combine(p, a);
                no += po;
            }
        }
        if (lane == 63) {
            wave_agg[wave] = a;
            wave_opens[wave] = no;
        }
        __syncthreads();
        Agg before = carry;  // everything in front of this wave
        uint32_t before_opens = carry_opens;
        for (int w = 0; w < wave; w++) {
            before = combine(before, wave_agg[w]);
            before_opens += wave_opens[w];
        }
        const Agg incl = combine(before, a);
        const uint32_t incl_opens = before_opens + no;
        int32_t run = incl.sum - mine.sum;  // depth at this thread's first block
        uint32_t ro = incl_opens - mine_opens;
#pragma unroll
        for (int k = 0; k < kScanPer; k++) {
            if (first + k < nblocks) {
                block_start[first + k] = run;
                open_start[first + k] = ro;
            }
            run += own[k].sum;
            ro += own_opens[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) {  // the last thread's inclusive values cover the whole pass
            carry = incl;
            carry_opens = incl_opens;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // the tokens in front of this call (msj_token_opts.d_prev): the running depth goes on from where they left it,
        // minimum and maximum are those of the stream so far
        const int32_t base = prev ? prev->final_depth : 0;
        const bool prev_has = prev && prev->n > 0;
        int32_t mn = (n && carry.mn != kNone) ? base + carry.mn : kNone;
        int32_t mx = (n && carry.mx != -kNone) ? base + carry.mx : -kNone;
        if (prev_has) {
            mn = min(mn, prev->min_depth);
            mx = max(mx, prev->max_depth);
        }
        // (kNone / -kNone left in place when the kernel in front only counted brackets: min_max_depth folds the exact
        // values in; an empty call at the start of a stream reports 0 / 0)
        const bool none = n == 0 && !prev_has;
        result->n = n;
        result->final_depth = base + carry.sum;
        result->min_depth = none ? 0 : mn;
        result->max_depth = none ? 0 : mx;
        result->reserved = carry_opens;  // number of opening brackets
    }
    // apply_depth / match_compact append the opening brackets they could not pair inside their block: one counter per list shard
    if (survivors && threadIdx.x < kSurvivorShards) survivors[threadIdx.x * kSurvivorStride] = 0u;
    if (resid && threadIdx.x < 4) resid[threadIdx.x] = 0u;  // the counts of this call's residual brackets (match_brackets, collect_closers)
}

// (3) depth of every token AND match[] -- a partner index per token -- for the calls that ask for it (the calls without,
// and the pairs form, take depth_rows below): the partner of every bracket whose container closes inside the block,
// the stack of start_container / end_container (generic/stage2/tape_builder.mojo:235-272) as a data-parallel step.
// The partner of a closing bracket at depth d is the MOST RECENT opening bracket at depth d in front of it (brackets
// of one depth alternate: between two closing ones the running depth must come back up through an opening one), so
// per depth level of the block a bitmap of its opening brackets (LDS, one bit per token, kMatchLevels levels from 4
// below the depth at the block's start) and a 64-bit summary of its non-empty words answer every closing bracket
// with at most three LDS reads and no loop.  Both ends are written into an LDS copy of the block's match[] slice
// (never initialised: one bit per token in s_paired says which of its words hold a partner),
// which leaves as one coalesced stream (no fill of match[] in front, no scattered writes); opening brackets nobody
// claimed (their container ends in a later block, or lies outside the levels) go on the survivors' list, which
// match_brackets resolves through the min tree.
constexpr int kMatchLevels = 16;
constexpr int kMatchBelow = 4;
// kFull: every token of the block exists (all blocks of a call but the last): no guards against n at all -- they were
// 64-bit compares, eight per loop (round 5: 661 -> ~450 vector instructions per wave together with block-relative
// 32-bit addressing, the DPP minima and the skipped document counts; profiles/r05/apply_depth_*.txt).
struct DepthShared {
    uint32_t bm[kMatchLevels][kBlock / 32];               // opening brackets per level, one bit per token
    unsigned long long bm_words[kMatchLevels];            // ... and which of a level's 64 words are not empty
    __attribute__((aligned(16))) uint32_t s_match[kBlock];
    uint32_t s_paired[kBlock / 32];                       // one bit per token: it has a partner
    int wave_sum[kThreads / 64];
    int wave_no[kThreads / 64];
    int wave_rmn[kThreads / 64], wave_rmx[kThreads / 64];
    uint32_t doc_cnt[kThreads / 64], doc_start[kThreads / 64], doc_close[kThreads / 64];
    __attribute__((aligned(16))) int s_out[kThreads / 64][512];          // the depths' way out (1 KiB contiguous per store instruction)
};
template <bool kFull>
__device__ __forceinline__ void apply_depth_block(DepthShared &sh, const uint8_t *__restrict__ type, const uint32_t nrem /* tokens of this block */,
                                                  const int block_depth0, int32_t *__restrict__ depth_blk, int32_t *__restrict__ min8,
                                                  int32_t *__restrict__ min64, int32_t *__restrict__ min512, uint32_t *__restrict__ opens,
                                                  uint4 *__restrict__ doc_agg, int32_t *__restrict__ block_mm,
                                                  uint32_t *__restrict__ match_blk, uint32_t *__restrict__ survivors, const uint32_t match_bias,
                                                  const uint32_t want_closers) {
    constexpr bool kMatch = true;
    // (the LDS lives in the kernel: two instantiations of this function must not own two copies of it)
    auto &bm = sh.bm;
    auto &bm_words = sh.bm_words;
    auto &s_match = sh.s_match;
    auto &s_paired = sh.s_paired;
    auto &wave_sum = sh.wave_sum;
    auto &wave_no = sh.wave_no;
    auto &wave_rmn = sh.wave_rmn;
    auto &wave_rmx = sh.wave_rmx;
    auto &doc_cnt = sh.doc_cnt;
    auto &doc_start = sh.doc_start;
    auto &doc_close = sh.doc_close;
    auto &s_out = sh.s_out;
    if (kMatch) {
        uint32_t *z = &bm[0][0];
#pragma unroll
        for (int k = 0; k < kMatchLevels * (int)(kBlock / 32) / kThreads; k++) z[threadIdx.x + k * kThreads] = 0u;
        if (threadIdx.x < kMatchLevels) bm_words[threadIdx.x] = 0ull;
        if (threadIdx.x < kBlock / 32) s_paired[threadIdx.x] = 0u;  // s_match itself is not initialised: s_paired says which words count
    }
    const uint32_t t0 = 8u * threadIdx.x;  // this thread's first token inside the block (everything below is block-relative)
    const uint32_t blk0 = blockIdx.x * kBlock;  // token indices are < 2^31
    // which of the thread's eight tokens exist (kFull: all)
    const uint32_t vm = kFull ? 0xFFu : (t0 >= nrem ? 0u : (nrem - t0 >= 8u ? 0xFFu : (1u << (nrem - t0)) - 1u));
    uint32_t c[kPer];
    if (kFull || vm == 0xFFu) {
        const uint2 t = *reinterpret_cast<const uint2 *>(type + t0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c[k] = (t.x >> (8 * k)) & 0xFFu;
            c[4 + k] = (t.y >> (8 * k)) & 0xFFu;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kPer; k++) c[k] = ((vm >> k) & 1u) ? type[t0 + k] : (uint32_t)' ';
    }
    int d[kPer], run = 0, no = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        d[k] = delta_of(c[k]);
        run += d[k];
        no += d[k] > 0;
    }
    // exclusive prefix of the thread sums (depth deltas, opening brackets) inside the block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = (int)wave_incl_sum((uint32_t)run), incl_no = (int)wave_incl_sum((uint32_t)no);
    if (lane == 63) {
        wave_sum[wave] = incl;
        wave_no[wave] = incl_no;
    }
    __syncthreads();
    int before = block_depth0 + incl - run;
    for (int w = 0; w < wave; w++) before += wave_sum[w];
    int out[kPer];
    int rmn = kNone, rmx = -kNone;  // minimum / maximum of the running depth AFTER each of this thread's tokens
#pragma unroll
    for (int k = 0; k < kPer; k++) {
        out[k] = before - (d[k] < 0 ? 1 : 0);  // a closing bracket sits at the depth of its container
        before += d[k];
        if (kFull || ((vm >> k) & 1u)) {
            rmn = min(rmn, before);
            rmx = max(rmx, before);
        }
    }
    bool surv_any = false;
    uint32_t surv_incl = 0, surv_slot = 0, surv_mine = 0, surv_mask = 0, cand_mask = 0;
    if (kMatch) {
        const int level0 = block_depth0 - kMatchBelow;
        // (a) the opening brackets' bits (the zeroing above is behind the barrier that published wave_sum).  Brackets are
        // few (6 % of the tokens of the minified workload are opening ones): a thread walks the SET BITS of its eight
        // tokens' bracket masks, so a wave runs max-over-lanes(brackets per thread) rounds, not eight; the depth of
        // token k follows from the masks (depth in front of the thread + opening - closing brackets below k).
        uint32_t om = 0, cm = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            om |= d[k] > 0 ? 1u << k : 0u;
            cm |= d[k] < 0 ? 1u << k : 0u;
        }
        if (!kFull) om &= vm, cm &= vm;  // (a blank stands in for a token that does not exist: no bracket anyway)
        const int before0 = out[0] + (d[0] < 0 ? 1 : 0);  // the running depth in front of this thread's first token
        // Round 5: the wave's brackets are COMPACTED first.  A thread holds 0 .. 8 of them (6 % of the minified workload's
        // tokens open a container, 6 % close one: ~60 per wave of 512 tokens), and a loop in which every thread walks its
        // own runs max-over-lanes rounds of the whole body -- 2 to 3 of the 22-instruction insertion and of the
        // 45-instruction look-up -- with a third of the lanes busy.  So a thread only works out WHERE its brackets are and at
        // which level (one short round per bracket), writes a 16-bit entry per bracket -- token inside the wave | level + 1
        // << 9, 0 = below the block's levels, 17 = above -- into the wave's list of opening or of closing brackets (the
        // depths' staging slice, free until the depths leave), and lane j then handles the wave's j-th bracket: one round of
        // each body for up to 64 brackets.  profiles/r05/apply_depth_*.txt.
        uint16_t *const olist = reinterpret_cast<uint16_t *>(&s_out[wave][0]);  // up to 512 entries each
        uint16_t *const clist = olist + 512;
        const uint32_t ncl = (uint32_t)__builtin_popcount(cm);
        const uint32_t incl_nc = wave_incl_sum(ncl);
        uint32_t oslot = (uint32_t)(incl_no - no), cslot = incl_nc - ncl;
        const uint32_t n_open_w = (uint32_t)__builtin_amdgcn_readlane(incl_no, 63), n_close_w = (uint32_t)__builtin_amdgcn_readlane((int)incl_nc, 63);
        for (uint32_t rem = om | cm; __ballot(rem != 0u) != 0ull;) {  // uniform
            if (rem != 0u) {
                const uint32_t k = (uint32_t)__builtin_ctz(rem), below = (1u << k) - 1u;
                rem &= rem - 1u;
                const uint32_t closes = (cm >> k) & 1u;
                // an opening bracket sits at the running depth in front of it, a closing one at the depth of its container
                const int lv = before0 + (int)__builtin_popcount(om & below) - (int)__builtin_popcount(cm & below) - (int)closes - level0;
                // a closing bracket BELOW the depth at the block's start: if nothing in the block pairs with it, its partner
                // is in an earlier block -- or in front of this CALL (the residuals of msj_stage2_prep_segments)
                if (closes && lv < kMatchBelow) cand_mask |= 1u << k;
                const uint32_t enc = (uint32_t)min(max(lv + 1, 0), kMatchLevels + 1);
                const uint16_t entry = (uint16_t)((8u * (uint32_t)lane + k) | (enc << 9));
                if (closes)
                    clist[cslot++] = entry;
                else
                    olist[oslot++] = entry;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (a) the opening brackets' bits, (b) the level's words that hold one
        for (uint32_t j0 = 0; j0 < n_open_w; j0 += 64u) {  // uniform: one round unless the wave holds more than 64
            const uint32_t j = j0 + (uint32_t)lane;
            if (j < n_open_w) {
                const uint32_t e = olist[j], lv = (e >> 9) - 1u, t = (uint32_t)wave * 512u + (e & 511u);
                if (lv < (uint32_t)kMatchLevels) {
                    atomicOr(&bm[lv][t >> 5], 1u << (t & 31u));
                    atomicOr(&bm_words[lv], 1ull << (t >> 5));
                }
            }
        }
        __syncthreads();
        // (c) every closing bracket looks for the most recent opening one of its level
        for (uint32_t j0 = 0; j0 < n_close_w; j0 += 64u) {  // uniform
            const uint32_t j = j0 + (uint32_t)lane;
            if (j < n_close_w) {
                const uint32_t e = clist[j], lv = (e >> 9) - 1u;
                if (lv < (uint32_t)kMatchLevels) {
                    const uint32_t t = (uint32_t)wave * 512u + (e & 511u), w = t >> 5;
                    uint32_t m = bm[lv][w] & ((1u << (t & 31u)) - 1u);
                    uint32_t wi = w;
                    if (m == 0u) {
                        const uint64_t nz = bm_words[lv] & ((1ull << w) - 1ull);
                        if (nz != 0ull) {
                            wi = 63u - (uint32_t)__clzll((long long)nz);
                            m = bm[lv][wi];
                        }
                    }
                    if (m != 0u) {
                        const uint32_t i = 32u * wi + 31u - (uint32_t)__clz((int)m);
                        const uint32_t b0 = blk0 + match_bias;  // + the call's place in the shard (msj_stage2_prep_segments)
                        s_match[t] = b0 + i;
                        s_match[i] = b0 + t;
                        atomicOr(&s_paired[w], 1u << (t & 31u));
                        atomicOr(&s_paired[wi], 1u << (i & 31u));
                    }
                }
            }
        }
        __syncthreads();
        // (d) the opening brackets nobody claimed: to the list match_brackets works through (any order)
        // (the thread's eight tokens are one byte of a word of s_paired)
        const uint32_t paired8 = s_paired[t0 >> 5] >> (t0 & 31u);
        surv_mask = om & ~paired8;
        // ... and (a shard call only) the closing brackets below the block's start depth that nothing in the block paired:
        // bits 8..15, listed with bit 31 set; match_brackets skips them, collect_closers keeps those still unpaired
        if (want_closers) surv_mask |= (cand_mask & ~paired8 & 0xFFu) << 8;
        const uint32_t mine = (uint32_t)__builtin_popcount(surv_mask);
        // the slot is DRAWN here (one returning atomic per wave that has any) and USED at the very end of the kernel: its
        // round trip overlaps the stores of match[] and depth[] and the aggregates below
        surv_any = __ballot(mine != 0u) != 0ull;  // uniform per wave
        if (surv_any) {
            surv_incl = wave_incl_sum(mine);
            if (lane == 63) surv_slot = atomicAdd(survivors + (blockIdx.x % kSurvivorShards) * kSurvivorStride, surv_incl);
        }
        surv_mine = mine;
        // (e) the block's slice of match[]: 1 KiB contiguous per store instruction, like the depths below
        const uint32_t wb = (uint32_t)wave * 512u;
        if (kFull || wb + 512u <= nrem) {  // uniform per wave
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            // tokens 4 lane .. 4 lane + 3 of the wave's first and second half: a nibble of s_paired each
            const uint32_t q0 = wb + 4u * lane, q1 = q0 + 256u;
            uint4 a = *reinterpret_cast<const uint4 *>(&s_match[q0]);
            uint4 b = *reinterpret_cast<const uint4 *>(&s_match[q1]);
            const uint32_t pa = s_paired[q0 >> 5] >> (q0 & 31u), pb = s_paired[q1 >> 5] >> (q1 & 31u);
            a.x = (pa & 1u) ? a.x : ~0u; a.y = (pa & 2u) ? a.y : ~0u; a.z = (pa & 4u) ? a.z : ~0u; a.w = (pa & 8u) ? a.w : ~0u;
            b.x = (pb & 1u) ? b.x : ~0u; b.y = (pb & 2u) ? b.y : ~0u; b.z = (pb & 4u) ? b.z : ~0u; b.w = (pb & 8u) ? b.w : ~0u;
            const u32x4 o0 = {a.x, a.y, a.z, a.w}, o1 = {b.x, b.y, b.z, b.w};
            __builtin_nontemporal_store(o0, reinterpret_cast<u32x4 *>(match_blk + q0));
            __builtin_nontemporal_store(o1, reinterpret_cast<u32x4 *>(match_blk + q1));
        } else {
#pragma unroll
            for (int k = 0; k < kPer; k++)
                if ((vm >> k) & 1u) match_blk[t0 + k] = ((paired8 >> k) & 1u) ? s_match[t0 + k] : ~0u;
        }
    }
    // The depths leave through LDS, so that each store instruction of a wave writes 1 KiB contiguous instead of 16 bytes
    // per lane at a 32-byte stride (a thread's eight tokens): msj_stage2_prep_device 0.865 -> 0.842 ms per GiB minified,
    // same box, alternating (plain instead of non-temporal stores: 0.90).  Write-once stream, far larger than L2 / MALL:
    // non-temporal stores.
    // (Round 4, measured with rocprofv3 on one box, alternating: letting the wave's quarter of s_match double as this
    // staging slice -- 13 KiB of LDS per workgroup instead of 21, eight resident workgroups per CU instead of seven --
    // changes nothing, 473 / 489 us either way; FEWER resident workgroups cost: six 533 us, four 616, three 752.)
    int *const stage = &s_out[wave][0];
    const uint32_t wave_base = (uint32_t)wave * 512u;
    if (kFull || wave_base + 512u <= nrem) {  // uniform per wave
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        *reinterpret_cast<int4 *>(&stage[8 * lane]) = make_int4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<int4 *>(&stage[8 * lane + 4]) = make_int4(out[4], out[5], out[6], out[7]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int4 a = *reinterpret_cast<const int4 *>(&stage[4 * lane]);
        const int4 b = *reinterpret_cast<const int4 *>(&stage[256 + 4 * lane]);
        const i32x4 o0 = {a.x, a.y, a.z, a.w}, o1 = {b.x, b.y, b.z, b.w};
        __builtin_nontemporal_store(o0, reinterpret_cast<i32x4 *>(depth_blk + wave_base + 4u * lane));
        __builtin_nontemporal_store(o1, reinterpret_cast<i32x4 *>(depth_blk + wave_base + 256u + 4u * lane));
    } else {  // the stream's last wave
#pragma unroll
        for (int k = 0; k < kPer; k++)
            if ((vm >> k) & 1u) depth_blk[t0 + k] = out[k];
    }
    {   // what the document split (documents_kernel.hip, doc_count) would recompute from type[] and depth[]:
        // per block the number of tokens that start a document (depth 0, not a closing bracket), the last of
        // them + 1 and the last closing bracket at depth 0 + 1
        uint32_t cnt = 0, ls = 0, lc = 0;
        // (round 5: a thread's eight tokens lie within 8 levels of its first one, and inside a large document no token of
        // a whole wave is anywhere near depth 0 -- one compare that IS the ballot skips ~80 vector instructions per wave;
        // streams of small documents take the counts as before)
        if (__ballot((uint32_t)(out[0] + 8) <= 16u) != 0ull) {  // uniform per wave
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                if ((kFull || ((vm >> k) & 1u)) && out[k] == 0) {
                    if (d[k] < 0) {
                        lc = blk0 + t0 + (uint32_t)k + 1u;
                    } else {
                        cnt++;
                        ls = blk0 + t0 + (uint32_t)k + 1u;
                    }
                }
            }
            // wave totals by DPP (token indices + 1 fit an int: n < 2^31); lane 63 holds the sum, every lane the maxima
            cnt = wave_incl_sum(cnt);
            int neg_lc, m_ls;  // one chain of minima, one of maxima: max(lc) = -min(-lc)
            wave_min_max(-(int)lc, (int)ls, neg_lc, m_ls);
            ls = (uint32_t)m_ls;
            lc = (uint32_t)(-neg_lc);
        }
        int w_rmn, w_rmx;
        wave_min_max(rmn, rmx, w_rmn, w_rmx);
        if (lane == 63) {
            doc_cnt[wave] = cnt;
            doc_start[wave] = ls;
            doc_close[wave] = lc;
            wave_rmn[wave] = w_rmn;
            wave_rmx[wave] = w_rmx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tc = 0, ts = 0, te = 0;
            int bmn = kNone, bmx = -kNone;
            for (int w = 0; w < kThreads / 64; w++) {
                tc += doc_cnt[w];
                ts = max(ts, doc_start[w]);
                te = max(te, doc_close[w]);
                bmn = min(bmn, wave_rmn[w]);
                bmx = max(bmx, wave_rmx[w]);
            }
            doc_agg[blockIdx.x] = make_uint4(tc, ts, te, 0);
            // the block's minimum / maximum running depth, into the words of its aggregate the scans are done with:
            // min_max_depth folds them into the result (one address for 10^5 blocks costs 80 us in atomics or in
            // the reads that would filter them: every request goes to the same L2 channel)
            block_mm[4 * (uint64_t)blockIdx.x + 1] = bmn;
            block_mm[4 * (uint64_t)blockIdx.x + 2] = bmx;
        }
    }
    if (min8) {  // the three lowest levels of the 8-ary min tree used for bracket matching (block-relative pointers)
        int m = kNone;
#pragma unroll
        for (int k = 0; k < kPer; k++)
            if (kFull || ((vm >> k) & 1u)) m = min(m, out[k]);
        if (kFull || vm) min8[threadIdx.x] = m;                      // 8 tokens = this thread
        // the minima over 8 threads and over the wave by DPP (round 5; six ds_bpermute butterflies before): an inclusive
        // scan to the right inside each row of 16 lanes -- after row_shr 1, 2, 4 lane 8g + 7 holds its group of eight --
        // then row_shr:8 and the two row broadcasts bring the wave's minimum to lane 63.  A lane without a source keeps
        // its value (a shift never crosses a row).  A group's / the wave's first token exists whenever any of it does.
        asm volatile(
            "s_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1"
            : "+v"(m));
        if ((threadIdx.x & 7) == 7 && (kFull || (t0 & ~63u) < nrem)) min64[threadIdx.x >> 3] = m;  // 64 tokens = 8 threads
        asm volatile(
            "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
            "s_nop 1"
            : "+v"(m));
        if (lane == 63 && (kFull || wave_base < nrem)) min512[wave] = m;  // 512 tokens = this wave
    }
    if (kMatch && surv_any) {  // uniform per wave: the opening brackets left to match_brackets, at the slot drawn above
        const uint32_t shard = blockIdx.x % kSurvivorShards;
        uint32_t *list = opens + (uint64_t)shard * survivor_capacity(gridDim.x);
        uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)surv_slot, 63) + surv_incl - surv_mine;
#pragma unroll
        for (int k = 0; k < kPer; k++)
            if ((surv_mask >> k) & 1u) list[slot++] = blk0 + t0 + (uint32_t)k;
        if (surv_mask >> 8) {
#pragma unroll
            for (int k = 0; k < kPer; k++)
                if ((surv_mask >> (8 + k)) & 1u) list[slot++] = (blk0 + t0 + (uint32_t)k) | 0x80000000u;
        }
    }
}

__global__ __launch_bounds__(kThreads) void apply_depth(const uint8_t *__restrict__ type, uint64_t n,
                                                        const int32_t *__restrict__ block_start, const int32_t *__restrict__ super_start,
                                                        int32_t *__restrict__ depth, int32_t *__restrict__ min8, int32_t *__restrict__ min64,
                                                        int32_t *__restrict__ min512, uint32_t *__restrict__ opens, uint4 *__restrict__ doc_agg,
                                                        int32_t *__restrict__ block_mm, const msj_tokens_result *__restrict__ prev,
                                                        uint32_t *__restrict__ match, uint32_t *__restrict__ survivors, uint32_t match_bias,
                                                        uint32_t want_closers) {
    // everything the block touches, as uniform (scalar) base pointers: the lanes add 32-bit offsets inside the block
    const uint64_t b0 = (uint64_t)blockIdx.x * kBlock;
    // tokens of this block: n < 2^31 (the entry points check), so 32-bit arithmetic and ONE s_min_u32.  (Written as a
    // 64-bit compare + select, hipcc 7.2 lowered the select to an s_cselect on an SCC that another select's condition had
    // set -- the not-full path then ran with nrem = 2 048 on the stream's last block and folded the stale type bytes
    // behind the stream's end into the minimum / maximum depth: found by test_stage2_prep_1gib_replicated.)
    const uint32_t nrem = min((uint32_t)n - blockIdx.x * kBlock, kBlock);
    const int block_depth0 = (prev ? prev->final_depth : 0) + super_start[blockIdx.x / kSuper] + block_start[blockIdx.x];  // uniform
    int32_t *m8 = min8 + (b0 >> 3), *m64 = min64 + (b0 >> 6), *m512 = min512 + (b0 >> 9);
    __shared__ DepthShared sh;
    if (nrem == kBlock)
        apply_depth_block<true>(sh, type + b0, nrem, block_depth0, depth + b0, m8, m64, m512, opens, doc_agg, block_mm, match + b0, survivors,
                                match_bias, want_closers);
    else
        apply_depth_block<false>(sh, type + b0, nrem, block_depth0, depth + b0, m8, m64, m512, opens, doc_agg, block_mm, match + b0, survivors,
                                 match_bias, want_closers);
}

// (3b) the same pass ORGANISED BY ROWS (round 5) for the calls without match[]: 64 tokens a row, lane l of row r holding
// token 64 r + l.  A row's opening and closing brackets are two 64-bit masks -- the compares ARE the ballots -- and
// everything the pass derives is a count of mask bits below the lane: the running depth in front of a token = depth in
// front of the row (scalar) + opening - closing brackets below the lane (v_mbcnt pairs; no per-thread loop, no DPP scan),
// the document counts are popcounts and find-last-bits of masks (scalar), and -- kCompact, the pairs form -- a bracket's
// place in the COMPACT list of the call's brackets is the row's base (scalar) + the brackets below the lane.
// A WAVE TAKES A WHOLE BLOCK of 2 048 tokens (32 rows, four chunks of eight), so a block's start depth from the scans is
// all it needs: no prefix over the waves of a workgroup, no LDS, no barrier -- the waves of a workgroup are independent,
// all 32 loads of a wave go out at once and their latency is paid once per 2 048 tokens.  (With a block per WORKGROUP,
// 512 tokens a wave, the kernel was bound by the workgroup's lifetime -- load latency, barrier, ~1 us of arithmetic, the
// list's round, barrier: 5 us per block at the 8 workgroups a CU holds, every dependent step added paid in full -- 267
// us for the pairs form against 175 for the depths; the per-thread form of apply_depth, a loop over the set bits of eight tokens'
// bracket masks, was bound by instruction issue: 356 vector instructions per wave of 512 tokens.)
// The compact list (kCompact): {token | closing << 31, depth} per bracket in token order, which match_compact pairs.
// A bracket's slot needs no scan of its own: in front of any token, opening + closing brackets = its slot and opening -
// closing = the running depth, so the brackets in front of a block = 2 x the opening ones - the depth they leave behind.
// profiles/r05/depth_pass_forms.txt.
__device__ __forceinline__ uint32_t bits_below(uint64_t m, uint32_t init) {  // init + bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, init));
}
constexpr uint32_t kRowsWaves = kThreads / 64;  // blocks per workgroup: a wave each
template <bool kCompact, bool kFull>
__device__ __forceinline__ void depth_rows_wave(uint32_t *__restrict__ blist /* kCompact: 512 words of LDS, this wave's */,
                                                const uint8_t *__restrict__ type_blk, const uint32_t blk /* block number */, const uint32_t nrem,
                                                const int block_depth0, int32_t *__restrict__ depth_blk, uint4 *__restrict__ doc_agg,
                                                int32_t *__restrict__ block_mm, uint32_t *__restrict__ brk_tok,
                                                int32_t *__restrict__ brk_depth, const uint32_t brk_base) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t blk0 = blk * kBlock;
    // the block's type bytes: 64 contiguous bytes per row and load, all of them asked for at once; a token that does not
    // exist reads as a blank (no bracket)
    uint32_t x[4][8];
#pragma unroll
    for (uint32_t c = 0; c < 4; c++)
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            const uint32_t t = 512u * c + 64u * r + lane;
            x[c][r] = (kFull || t < nrem) ? (uint32_t)type_blk[t] : 0x20u;
        }
    int D = block_depth0;  // running depth in front of the row (scalar)
    uint32_t G = brk_base;  // (compact) the call's brackets in front of the chunk
    int rmn = kNone, rmx = -kNone;  // minimum / maximum of the running depth AFTER each of this lane's tokens
    uint32_t cnt = 0, ls = 0, lc = 0;  // scalar: document starts of the block, the last of them + 1, the last closing bracket at depth 0 + 1
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
        if (!kFull && 512u * c >= nrem) break;  // uniform
        uint64_t up[8], dn[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t v = x[c][r] & 0xDFu;  // '[' '{' -> 5B, ']' '}' -> 5D
            up[r] = __ballot(v == 0x5Bu);
            dn[r] = __ballot(v == 0x5Du);
        }
        uint32_t S = 0;  // (compact) brackets of the chunk in front of the row
#pragma unroll
        for (uint32_t r = 0; r < 8; r++) {
            const uint32_t t = 512u * c + 64u * r + lane;
            // the masks as lane predicates (no second compare: the mask IS the execution mask / the select's condition)
            const bool isup = __builtin_amdgcn_inverse_ballot_w64(up[r]);
            // a closing bracket sits at the depth of its container, one below the running depth in front of it: the closing
            // brackets AT or below the lane = those below the lane of the mask shifted down by one + its bit 0 (scalar)
            const int out = (int)bits_below(up[r], (uint32_t)D) - (int)bits_below(dn[r] >> 1, (uint32_t)(dn[r] & 1ull));
            const int after = out + (isup ? 1 : 0);
            const bool exists = kFull || t < nrem;
            if (exists) {
                rmn = min(rmn, after);
                rmx = max(rmx, after);
                __builtin_nontemporal_store(out, depth_blk + t);  // 256 contiguous bytes per row: a write-once stream
            }
            // what the document split (documents_kernel.hip) would recompute from type[] and depth[]: tokens at depth 0 --
            // no row of a large document holds one, the compare IS the ballot
            const uint64_t zero = __ballot(exists && out == 0);
            if (zero != 0ull) {  // uniform
                const uint64_t zs = zero & ~dn[r], zc = zero & dn[r];
                cnt += (uint32_t)__popcll(zs);
                if (zs) ls = blk0 + 512u * c + 64u * r + 64u - (uint32_t)__clzll((long long)zs);
                if (zc) lc = blk0 + 512u * c + 64u * r + 64u - (uint32_t)__clzll((long long)zc);
            }
            if (kCompact) {
                const uint64_t m = up[r] | dn[r];
                if (m != 0ull) {  // uniform
                    const uint32_t pos = bits_below(m, S);
                    // one word per bracket: token inside the chunk | closing << 9 | depth relative to the block's start << 10
                    const uint32_t e = ((uint32_t)(out - block_depth0) << 10) | (__builtin_amdgcn_inverse_ballot_w64(dn[r]) ? 512u : 0u) | (64u * r + lane);
                    if (__builtin_amdgcn_inverse_ballot_w64(m)) blist[pos] = e;
                    S += (uint32_t)__popcll(m);
                }
            }
            D += (int)__popcll(up[r]) - (int)__popcll(dn[r]);
        }
        if (kCompact && S != 0u) {  // uniform: lane j takes the chunk's j-th bracket, two coalesced stores per 64 of them
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (scalar bases, the lane's 32-bit offset behind them: no 64-bit vector arithmetic)
            uint32_t *const bt = brk_tok + G;
            int32_t *const bd = brk_depth + G;
            const uint32_t tok0 = blk0 + 512u * c;
#pragma unroll 1
            for (uint32_t j0 = 0; j0 < S; j0 += 64u) {  // uniform: one round unless the chunk holds more than 64 brackets
                const uint32_t j = j0 + lane;
                if (j < S) {
                    const uint32_t e = blist[j];
                    bt[j] = (tok0 + (e & 511u)) | ((e >> 9) << 31);
                    bd[j] = block_depth0 + ((int)e >> 10);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the next chunk writes the list again)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            G += S;
        }
    }
    int w_rmn, w_rmx;
    wave_min_max(rmn, rmx, w_rmn, w_rmx);
    if (lane == 0u) {
        doc_agg[blk] = make_uint4(cnt, ls, lc, 0);
        // (words 1 and 2 of the block's aggregate: min_max_depth folds them into the result, as behind apply_depth)
        block_mm[4 * (uint64_t)blk + 1] = w_rmn;
        block_mm[4 * (uint64_t)blk + 2] = w_rmx;
    }
}
template <bool kCompact>
__global__ __launch_bounds__(kThreads) void depth_rows(const uint8_t *__restrict__ type, uint64_t n, uint32_t nblocks,
                                                       const int32_t *__restrict__ block_start,
                                                       const int32_t *__restrict__ super_start, const uint32_t *__restrict__ super_open,
                                                       const uint32_t *__restrict__ open_start, int32_t *__restrict__ depth,
                                                       uint4 *__restrict__ doc_agg, int32_t *__restrict__ block_mm,
                                                       const msj_tokens_result *__restrict__ prev, uint32_t *__restrict__ brk_tok,
                                                       int32_t *__restrict__ brk_depth) {
    __shared__ uint32_t s_list[kCompact ? kRowsWaves : 1][kCompact ? 512 : 1];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t blk = blockIdx.x * kRowsWaves + wave;
    if (blk >= nblocks) return;  // uniform per wave (no barrier anywhere in the kernel)
    const uint64_t b0 = (uint64_t)blk * kBlock;
    const uint32_t nrem = min((uint32_t)n - blk * kBlock, kBlock);  // (32-bit: see apply_depth)
    const int rel0 = super_start[blk / kSuper] + block_start[blk];  // depth in front of the block, relative to the call's start
    const int block_depth0 = (prev ? prev->final_depth : 0) + rel0;
    const uint32_t brk_base = kCompact ? 2u * (super_open[blk / kSuper] + open_start[blk]) - (uint32_t)rel0 : 0u;
    uint32_t *blist = &s_list[kCompact ? wave : 0][0];
    if (nrem == kBlock)
        depth_rows_wave<kCompact, true>(blist, type + b0, blk, nrem, block_depth0, depth + b0, doc_agg, block_mm, brk_tok, brk_depth, brk_base);
    else
        depth_rows_wave<kCompact, false>(blist, type + b0, blk, nrem, block_depth0, depth + b0, doc_agg, block_mm, brk_tok, brk_depth, brk_base);
}

// (4) minimum / maximum of the running depth over the stream, from the per-block values apply_depth left in the block
//     aggregates (words 1 and 2): the scan set the result's fields to what ITS aggregates gave -- nothing (kNone /
//     -kNone) when the kernel in front only counted brackets (token_tiles) -- and this folds the exact values in.
constexpr uint32_t kMinMaxGroups = 64;
__global__ __launch_bounds__(256) void min_max_depth(const int32_t *__restrict__ block_mm, uint32_t nblocks, msj_tokens_result *__restrict__ result) {
    __shared__ int w_mn[4], w_mx[4];
    int mn = kNone, mx = -kNone;
    for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < nblocks; b += gridDim.x * 256u) {
        const int4 q = *reinterpret_cast<const int4 *>(block_mm + 4 * (uint64_t)b);
        mn = min(mn, q.y);
        mx = max(mx, q.z);
    }
    int wmn, wmx;
    wave_min_max(mn, mx, wmn, wmx);
    if ((threadIdx.x & 63) == 63) {
        w_mn[threadIdx.x >> 6] = wmn;
        w_mx[threadIdx.x >> 6] = wmx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = min(min(w_mn[0], w_mn[1]), min(w_mn[2], w_mn[3]));
        mx = max(max(w_mx[0], w_mx[1]), max(w_mx[2], w_mx[3]));
        if (mn != kNone) atomicMin(&result->min_depth, mn);
        if (mx != -kNone) atomicMax(&result->max_depth, mx);
    }
}

// ---- bracket matching: match[i] = index of the other end of the container a bracket opens or
// closes (what start_container / end_container keep on a stack, generic/stage2/tape_builder.mojo:
// 235-272), 0xFFFFFFFF for every other token and for brackets without a partner.
// The partner of an opening bracket at token i (depth d) is the first j > i with depth[j] <= d:
// everything inside the container is deeper.  An 8-ary min tree over depth[] answers that in
// O(8 * levels) with levels ~ log8(container size): scan to the end of the group, climb while
// nothing qualifies, descend into the first node whose minimum does.
constexpr int kFanShift = 3;
constexpr uint32_t kFanMask = (1u << kFanShift) - 1u;
constexpr int kMaxLevels = 12;  // 8^11 > 2^32
struct MinTree {
    const int32_t *lv[kMaxLevels];  // lv[0] = depth, lv[k][g] = min of lv[k-1][8g .. 8g+7]
    uint32_t cnt[kMaxLevels];
    int nlev;
};

// level k from level k-1 (the levels above the three that apply_depth writes are tiny)
__global__ __launch_bounds__(256) void build_level(const int32_t *__restrict__ in, uint32_t n_in, int32_t *__restrict__ out, uint32_t n_out) {
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    if (o >= n_out) return;
    int m = kNone;
    for (uint32_t k = 0; k <= kFanMask; k++) {
        const uint32_t i = (o << kFanShift) + k;
        if (i < n_in) m = min(m, in[i]);
    }
    out[o] = m;
}

// levels 6 .. of the tree in ONE single-workgroup launch (they hold n / 262 144 entries and less; one launch of
// build_level each cost 28 us of launch latency per call, one workgroup from level 5 on 30 us: level 5 is too long for it)
struct UpperLevels {
    int32_t *lv[kMaxLevels];
    uint32_t cnt[kMaxLevels];
    int first, nlev;  // builds lv[first .. nlev) from lv[first - 1]
};
__global__ __launch_bounds__(1024) void build_upper_levels(const UpperLevels u) {
    for (int k = u.first; k < u.nlev; k++) {
        const int32_t *in = u.lv[k - 1];
        const uint32_t n_in = u.cnt[k - 1], n_out = u.cnt[k];
        for (uint32_t o = threadIdx.x; o < n_out; o += 1024u) {
            int m = kNone;
            for (uint32_t j = 0; j <= kFanMask; j++) {
                const uint32_t i = (o << kFanShift) + j;
                if (i < n_in) m = min(m, in[i]);
            }
            u.lv[k][o] = m;
        }
        __threadfence_block();
        __syncthreads();
    }
}

// the 8 entries of group g at one level; a level's last group is read with a guard at its end (the padding of a
// level is never written: only the brackets a block could not pair walk the tree, so the guard costs nothing that
// shows and saves a fill of the tree in front of every call)
__device__ __forceinline__ void load_group(const MinTree &t, int lev, uint32_t g, int v[8], const uint32_t cnt) {  // cnt: entries of the level
    const int32_t *p = t.lv[lev] + ((uint64_t)g << kFanShift);
    if (((uint64_t)g << kFanShift) + 8u <= cnt) {
        const int4 a = *reinterpret_cast<const int4 *>(p);
        const int4 b = *reinterpret_cast<const int4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = (((uint64_t)g << kFanShift) + k < cnt) ? p[k] : kNone;
    }
}
// first k >= from with v[k] <= target, 8 if none
__device__ __forceinline__ uint32_t first_le(const int v[8], uint32_t from, int target) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) m |= (uint32_t)(v[k] <= target) << k;
    m &= 0xFFu << from;
    return m ? (uint32_t)__builtin_ctz(m) : 8u;
}

// EIGHT LANES per opening bracket the depth pass could not pair inside its block (1.4 % of the containers of the
// BASELINE workloads: those that span a block border; their partner is a median 12, at the 99th percentile 146 tokens
// away).  The eight first look at the next kLinear tokens, 32 coalesced bytes of depth[] per step and a ballot; only a
// container longer than that climbs the min tree (the group's first lane), from where the scan stopped.  Measured per
// GiB minified, 180 000 such brackets: one THREAD per bracket walking the tree 93 us (every load instruction of such a
// wave is 64 scattered 32-byte reads), sixteen lanes per bracket with one step of the scan and one level of the climb
// per round trip 84 us -- the longest chain of one bracket, whatever the grid -- and with the rounds below ...
#ifndef MSJ_MATCH_GRID
#define MSJ_MATCH_GRID 32
#endif
#ifndef MSJ_MATCH_LINEAR
#define MSJ_MATCH_LINEAR 256
#endif
#ifndef MSJ_COMPACT_GRID
#define MSJ_COMPACT_GRID 4096  // workgroups of match_compact at most (each strides over the blocks of 2 048 brackets)
#endif
#ifndef MSJ_MATCH_LINEAR_COMPACT
#define MSJ_MATCH_LINEAR_COMPACT 64  // on the compact list: one round (the partners of the brackets a block of 2 048 brackets left over are mostly further away: 30.0 us against 33.9 with four rounds, 57 with none)
#endif
constexpr uint32_t kLinear = MSJ_MATCH_LINEAR, kLinearCompact = MSJ_MATCH_LINEAR_COMPACT, kGroup = 8, kSteps = 8;  // 64 tokens per group and round
// kCompact (round 5, the pairs form): the same walk over the COMPACT list of the call's brackets -- t_in.lv[0] is that
// list's depth word per bracket, brk_tok its token word (bit 31: a closing bracket), an entry of the lists the bracket's
// place in the compact list, and the number of brackets -- hence every level's count -- is only known on the device:
// 2 x the opening brackets - the depth they leave behind (scan_blocks put both into the result).
__device__ __forceinline__ uint32_t bracket_count(const msj_tokens_result *__restrict__ result, const msj_tokens_result *__restrict__ prev) {
    return 2u * result->reserved - (uint32_t)(result->final_depth - (prev ? prev->final_depth : 0));
}
// entries of level `lev` of a tree over n0 entries: ceil(n0 / 8^lev) (a ceiling of ceilings is the ceiling of the product)
__device__ __forceinline__ uint32_t level_entries(uint32_t n0, int lev) {
    return (uint32_t)(((uint64_t)n0 + (1ull << (kFanShift * lev)) - 1ull) >> (kFanShift * lev));
}
__device__ __forceinline__ int level_number(uint32_t n0) {  // as the host lays the levels out: one more while the last holds more than 8
    int nlev = 1;
    while (nlev < kMaxLevels && level_entries(n0, nlev - 1) > 8u) nlev++;
    return nlev;
}
template <bool kCompact>
__global__ __launch_bounds__(256) void match_brackets(const uint8_t *__restrict__ type, const uint32_t *__restrict__ opens,
                                                      const uint32_t *__restrict__ n_opens, const MinTree t_in,
                                                      uint32_t *__restrict__ match, uint64_t list_capacity, uint32_t match_bias,
                                                      const msj_tokens_result *__restrict__ result, uint32_t *__restrict__ resid,
                                                      uint2 *__restrict__ pairs, const uint32_t *__restrict__ brk_tok,
                                                      const msj_tokens_result *__restrict__ prev) {
    const MinTree &t = t_in;
    // (compact: the counts follow from the call's number of brackets; computed, not kept in an array a loop would index)
    const uint32_t n = kCompact ? bracket_count(result, prev) : t.cnt[0];
    const int nlev = kCompact ? level_number(n) : t.nlev;
    const auto count_of = [&](int lev) -> uint32_t { return kCompact ? level_entries(n, lev) : t.cnt[lev]; };
    // one list per blockIdx.y (kSurvivorShards of them), the lane groups of its workgroups stride over it
    const uint32_t total = n_opens[blockIdx.y * kSurvivorStride];
    opens += (uint64_t)blockIdx.y * list_capacity;
    const uint32_t lane = threadIdx.x & 63u, sub = lane & (kGroup - 1u), grp = lane / kGroup;
    constexpr uint32_t per_wave = 64u / kGroup;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * per_wave, stride = (uint64_t)gridDim.x * 4u * per_wave;
    for (uint64_t w0 = wave0; w0 < total; w0 += stride) {  // uniform per wave
        const uint64_t w = w0 + grp;
        const uint32_t entry = w < total ? opens[w] : 0x80000000u;
        const bool have = (entry >> 31) == 0u;  // (bit 31: a closing bracket listed for collect_closers)
        const uint32_t i = have ? entry : 0u;
        const int target = have ? t.lv[0][i] : kNone;
        uint32_t pos = i + 1u;
        bool found = false;
        // What bounds this kernel is the LONGEST dependent chain of loads of any one bracket (the root array of a 64 MiB
        // document: 84 us, whatever the grid), so loads whose addresses do not depend on loaded data go out together:
        // four steps of the linear scan per round, and every level of the climb at once.
        for (uint32_t round = 0; round < (kCompact ? kLinearCompact : kLinear) / (kSteps * kGroup); round++) {  // uniform
            int v4[kSteps];
#pragma unroll
            for (uint32_t q = 0; q < kSteps; q++) {
                const uint32_t j = pos + q * kGroup + sub;
                v4[q] = (have && !found && j < n) ? t.lv[0][j] : kNone;
            }
            uint32_t first = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t q = 0; q < kSteps; q++) {
                const uint64_t hit = __ballot(v4[q] <= target && v4[q] != kNone);
                const uint32_t mine = (uint32_t)(hit >> (kGroup * grp)) & ((1u << kGroup) - 1u);
                if (mine != 0u && first == 0xFFFFFFFFu) first = q * kGroup + (uint32_t)__builtin_ctz(mine);
            }
            if (have && !found) {
                if (first != 0xFFFFFFFFu) {
                    pos += first;
                    found = true;
                } else {
                    pos += kSteps * kGroup;
                }
            }
            if (__ballot(have && !found && pos < n) == 0ull) break;  // uniform: every group of the wave is done
        }
        if (have && !found && pos < n && sub == 0u) {  // a long container: the tree, from where the scan stopped
            // climb: the node visited at level k + 1 is (node at level k >> 3) + 1 whatever level k holds -- all levels'
            // groups are requested at once, the first level with an entry <= target decides
            uint32_t at[kMaxLevels];
            int4 ga[kMaxLevels], gb[kMaxLevels];
            uint32_t p = pos;
#pragma unroll
            for (int lev = 0; lev < kMaxLevels; lev++) {
                at[lev] = p;
                const uint32_t g = p >> kFanShift;
                const bool in = lev < nlev && ((uint64_t)g << kFanShift) < count_of(lev);
                ga[lev] = gb[lev] = make_int4(kNone, kNone, kNone, kNone);
                if (in) {
                    int v[8];
                    load_group(t, lev, g, v, count_of(lev));
                    ga[lev] = make_int4(v[0], v[1], v[2], v[3]);
                    gb[lev] = make_int4(v[4], v[5], v[6], v[7]);
                }
                p = g + 1u;
            }
            int lev_hit = -1;
#pragma unroll
            for (int lev = kMaxLevels - 1; lev >= 0; lev--) {
                const int v[8] = {ga[lev].x, ga[lev].y, ga[lev].z, ga[lev].w, gb[lev].x, gb[lev].y, gb[lev].z, gb[lev].w};
                const uint32_t k = first_le(v, at[lev] & kFanMask, target);
                if (k < 8u) {  // the LOWEST level that qualifies wins (the loop runs downwards)
                    lev_hit = lev;
                    pos = ((at[lev] >> kFanShift) << kFanShift) + k;
                }
            }
            if (lev_hit >= 0) {
                found = true;
                int v[8];
                for (int lev = lev_hit; lev > 0;) {  // descend: the first child that qualifies
                    lev--;
                    load_group(t, lev, pos, v, count_of(lev));
                    pos = (pos << kFanShift) + first_le(v, 0u, target);
                }
            }
        }
        if (kCompact) {
            if (have && sub == 0u) {
                // the record's place: the opening brackets in front of bracket i = (i + the depth in front of it) / 2
                const uint32_t r = (i + (uint32_t)(target - (prev ? prev->final_depth : 0))) >> 1;
                const uint32_t cj = found ? brk_tok[pos] : 0u;
                pairs[r] = make_uint2(brk_tok[i], (cj >> 31) ? (cj & 0x7FFFFFFFu) : 0xFFFFFFFFu);
            }
        } else if (have && found && sub == 0u) {
            const uint32_t cj = type[pos];
            if (cj == '}' || cj == ']') {
                match[i] = pos + match_bias;
                match[pos] = i + match_bias;
            }
        } else if (have && sub == 0u && resid) {
            // never closed inside this call: the unclosed opening brackets nest, so the one at depth `target` is entry
            // final_depth - 1 - target of the call's residual list (msj_stage2_prep_segments stitches the segments)
            const uint32_t j = (uint32_t)(result->final_depth - 1 - target);
            atomicAdd(&resid[0], 1u);
            if (j < MSJ_RESID_CAP) resid[4u + j] = i;
        }
    }
}

// ---- residuals: bracket partners over a whole shard (msj_stage2_prep_segments; the stack of start_container /
// end_container, generic/stage2/tape_builder.mojo:235-272, has no segment border).  A call pairs what closes inside it.
// What is left are (a) opening brackets never closed in the call -- match_brackets finds them: the walk of a survivor
// that reaches the call's end -- and (b) closing brackets whose partner is in front of the call: among the listed
// candidates (apply_depth: below the block's start depth, unpaired in the block) those that match_brackets, which
// writes both ends, has not touched either.  Both sets nest, so the bracket's DEPTH is its place in the list: no
// sorting, no counting pass.  stitch_partners then pairs closing bracket k of segment s (depth c) with the unclosed
// opening bracket at depth c of the latest segment in front that holds one.
__global__ __launch_bounds__(256) void collect_closers(const uint32_t *__restrict__ opens, const uint32_t *__restrict__ n_opens,
                                                       uint64_t list_capacity, const int32_t *__restrict__ depth,
                                                       const uint32_t *__restrict__ match, const msj_tokens_result *__restrict__ prev,
                                                       uint32_t *__restrict__ resid) {
    const uint32_t total = n_opens[blockIdx.y * kSurvivorStride];
    opens += (uint64_t)blockIdx.y * list_capacity;
    const int d0 = prev ? prev->final_depth : 0;
    for (uint64_t w = (uint64_t)blockIdx.x * 256u + threadIdx.x; w < total; w += (uint64_t)gridDim.x * 256u) {
        const uint32_t entry = opens[w];
        if (!(entry >> 31)) continue;
        const uint32_t i = entry & 0x7FFFFFFFu;
        if (match[i] != 0xFFFFFFFFu) continue;  // an opening bracket of an earlier block of this call claimed it
        const uint32_t k = (uint32_t)(d0 - 1 - depth[i]);
        atomicAdd(&resid[1], 1u);
        if (k < MSJ_RESID_CAP) resid[4u + MSJ_RESID_CAP + k] = i;
    }
}

__global__ __launch_bounds__(256) void stitch_partners(const msj_stitch_args a, uint32_t *__restrict__ match,
                                                       msj_tokens_result *__restrict__ results, const msj_tokens_result *__restrict__ prev) {
    const uint32_t s = blockIdx.x + 1u;  // segment 0 has nothing in front of it inside the shard
    if (s >= a.n_segments) return;
    const uint32_t *rs = a.resid[s];
    const uint32_t n_close = rs[1];
    bool clipped = n_close > MSJ_RESID_CAP;
    const int d0 = results[s - 1].final_depth;  // the depth this segment starts at
    for (uint32_t k = threadIdx.x; k < n_close && k < MSJ_RESID_CAP; k += 256u) {
        const int c = d0 - 1 - (int)k;  // the closing bracket's depth = its container's
        const uint32_t ci = rs[4u + MSJ_RESID_CAP + k];
        for (int q = (int)s - 1; q >= 0; q--) {
            const uint32_t *rq = a.resid[q];
            const int df = results[q].final_depth, nu = (int)rq[0];
            if (c < df && c >= df - nu) {  // segment q left an opening bracket at this depth open
                const uint32_t j = (uint32_t)(df - 1 - c);
                if (j < MSJ_RESID_CAP) {
                    const uint32_t oi = rq[4u + j];
                    match[a.offsets[s] + ci] = a.offsets[q] + oi;
                    match[a.offsets[q] + oi] = a.offsets[s] + ci;
                } else {
                    clipped = true;
                }
                break;
            }
            // (else: segment q closed that level again or never reached it: look further in front; below every
            //  segment's levels the partner lies in front of the shard and the bracket keeps 0xFFFFFFFF)
        }
    }
    (void)prev;
    if (clipped) atomicOr(&results[a.n_segments - 1u].reserved, 0x80000000u);  // nesting deeper than MSJ_RESID_CAP at a border
}


// ---- the pairs form (round 5): bracket partners on the COMPACT list depth_rows<true> leaves -- brk_tok[j] = token |
// closing << 31, brk_depth[j] = the bracket's depth, j in token order.  A workgroup takes 2 048 brackets (the containers of
// ~17 000 tokens of the minified workload): the same level bitmaps as apply_depth -- the partner of a closing bracket
// at depth d is the most recent opening one at depth d in front of it -- but every lane's every slot is a bracket, the
// depths are given (no scan), and far fewer containers span a border.  A container's record goes to the place of its
// opening bracket among the call's opening ones: (slot + depth in front) / 2.  Opening brackets nobody claimed go on the
// lists match_brackets<true> walks, with the three lowest levels of the min tree over brk_depth[] written on the way.
// (Measured and dropped, profiles/r05/match_compact_history.txt: a WAVE per 2 048 brackets, 32 rows of 64, pairing through
// the fact that the brackets of one level alternate -- a closing bracket's partner is the level's bracket in front of it:
// the next lower bit of the level's ballot, or the one carried from the rows before in lane `level` of a register; no LDS,
// no barrier -- is bit-exact and costs a loop over the levels present in every row: ~8 000 instructions per block against
// 2 400 here, 183 us per GiB minified against 81; a wave per 2 048 brackets WITH these bitmaps, kept per piece of 512 in the
// wave's own LDS, the level's last opening bracket carried from piece to piece: 117 - 128 us -- 12 200 waves of ~40 us each
// do not fill the chip, and the wait for a piece's words is a wait for the stores of the piece in front.)
constexpr uint32_t kCompactBlock = 2048;
__global__ __launch_bounds__(256) void match_compact(const uint32_t *__restrict__ brk_tok, const int32_t *__restrict__ brk_depth,
                                                     const msj_tokens_result *__restrict__ result, const msj_tokens_result *__restrict__ prev,
                                                     uint2 *__restrict__ pairs, int32_t *__restrict__ min8, int32_t *__restrict__ min64,
                                                     int32_t *__restrict__ min512, uint32_t *__restrict__ opens,
                                                     uint32_t *__restrict__ survivors, uint64_t list_capacity) {
    __shared__ uint32_t bm[kMatchLevels][kCompactBlock / 32];
    __shared__ unsigned long long bm_words[kMatchLevels];
    __shared__ uint32_t s_paired[kCompactBlock / 32];
    __shared__ __attribute__((aligned(16))) uint32_t s_tok[kCompactBlock];
    const int d_call = prev ? prev->final_depth : 0;
    const uint32_t nbrk = bracket_count(result, prev);
    const uint32_t ncb = (nbrk + kCompactBlock - 1u) / kCompactBlock;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t p0 = 8u * threadIdx.x;  // this thread's first bracket inside the block
    for (uint32_t cb = blockIdx.x; cb < ncb; cb += gridDim.x) {  // uniform
        const uint32_t base = cb * kCompactBlock, nrem = min(nbrk - base, kCompactBlock);
#define MSJ_CSTAMP(k) MSJ_STAMP_AT((uint64_t)cb * 4u + wave, k)
        MSJ_CSTAMP(0);
        {
            uint32_t *z = &bm[0][0];
#pragma unroll
            for (int k = 0; k < kMatchLevels * (int)(kCompactBlock / 32) / 256; k++) z[threadIdx.x + k * 256] = 0u;
            if (threadIdx.x < kCompactBlock / 32) s_paired[threadIdx.x] = 0u;
        }
        uint32_t tok[8];
        int dep[8];
        const uint32_t vm = p0 >= nrem ? 0u : (nrem - p0 >= 8u ? 0xFFu : (1u << (nrem - p0)) - 1u);
        if (vm == 0xFFu) {
            const uint4 a = *reinterpret_cast<const uint4 *>(brk_tok + base + p0), b = *reinterpret_cast<const uint4 *>(brk_tok + base + p0 + 4u);
            const int4 c = *reinterpret_cast<const int4 *>(brk_depth + base + p0), e = *reinterpret_cast<const int4 *>(brk_depth + base + p0 + 4u);
            tok[0] = a.x, tok[1] = a.y, tok[2] = a.z, tok[3] = a.w, tok[4] = b.x, tok[5] = b.y, tok[6] = b.z, tok[7] = b.w;
            dep[0] = c.x, dep[1] = c.y, dep[2] = c.z, dep[3] = c.w, dep[4] = e.x, dep[5] = e.y, dep[6] = e.z, dep[7] = e.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const bool in = (vm >> k) & 1u;
                tok[k] = in ? brk_tok[base + p0 + k] : 0u;
                dep[k] = in ? brk_depth[base + p0 + k] : kNone;
            }
        }
        *reinterpret_cast<uint4 *>(&s_tok[p0]) = make_uint4(tok[0], tok[1], tok[2], tok[3]);
        *reinterpret_cast<uint4 *>(&s_tok[p0 + 4u]) = make_uint4(tok[4], tok[5], tok[6], tok[7]);
        uint32_t cm = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) cm |= (tok[k] >> 31) << k;
        cm &= vm;
        const uint32_t om = ~cm & vm;
        // the levels kept: from kMatchBelow below the running depth in front of the block's first bracket (uniform, scalar loads)
        const int level0 = brk_depth[base] + (int)(brk_tok[base] >> 31) - kMatchBelow;
        MSJ_CSTAMP(1);
        __syncthreads();
        MSJ_CSTAMP(2);
        // (a) the opening brackets' bits (a thread's eight brackets share a word)
        const uint32_t word = p0 >> 5, sh0 = p0 & 31u;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t lv = (uint32_t)(dep[k] - level0);
            if (((om >> k) & 1u) && lv < (uint32_t)kMatchLevels) atomicOr(&bm[lv][word], 1u << (sh0 + k));
        }
        __syncthreads();
        MSJ_CSTAMP(3);
        // (b) the level's words that hold one (every lane here holds brackets: 64 lanes OR-ing one summary word would
        // serialise; a wave takes four levels, the compare of a level's 64 words IS the ballot)
#pragma unroll
        for (uint32_t q = 0; q < kMatchLevels / 4u; q++) {
            const uint32_t lv = 4u * wave + q;
            const uint64_t nz = __ballot(bm[lv][lane] != 0u);
            if (lane == 0u) bm_words[lv] = nz;
        }
        __syncthreads();
        MSJ_CSTAMP(4);
        // (c) every closing bracket looks for the most recent opening one of its level
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t lv = (uint32_t)(dep[k] - level0);
            if (((cm >> k) & 1u) && lv < (uint32_t)kMatchLevels) {
                uint32_t m = bm[lv][word] & ((1u << (sh0 + k)) - 1u);
                uint32_t wi = word;
                if (m == 0u) {
                    const uint64_t nz = bm_words[lv] & ((1ull << word) - 1ull);
                    if (nz != 0ull) {
                        wi = 63u - (uint32_t)__clzll((long long)nz);
                        m = bm[lv][wi];
                    }
                }
                if (m != 0u) {
                    const uint32_t i = 32u * wi + 31u - (uint32_t)__clz((int)m);
                    const uint32_t r = (base + i + (uint32_t)(dep[k] - d_call)) >> 1;
                    pairs[r] = make_uint2(s_tok[i], tok[k] & 0x7FFFFFFFu);
                    atomicOr(&s_paired[wi], 1u << (i & 31u));
                }
            }
        }
        __syncthreads();
        MSJ_CSTAMP(5);
        // (d) the opening brackets nobody claimed: to the lists match_brackets<true> works through (any order)
        {
            const uint32_t surv = om & ~(s_paired[word] >> sh0);
            const uint32_t mine = (uint32_t)__builtin_popcount(surv);
            if (__ballot(mine != 0u) != 0ull) {  // uniform per wave
                const uint32_t incl = wave_incl_sum(mine);
                const uint32_t shard = cb % kSurvivorShards;
                uint32_t slot0 = 0;
                if (lane == 63u) slot0 = atomicAdd(survivors + shard * kSurvivorStride, incl);
                uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)slot0, 63) + incl - mine;
                uint32_t *list = opens + (uint64_t)shard * list_capacity;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((surv >> k) & 1u) list[slot++] = base + p0 + (uint32_t)k;
            }
        }
        MSJ_CSTAMP(6);
        // (e) the three lowest levels of the 8-ary min tree over brk_depth[] (as apply_depth over depth[])
        {
            int m = kNone;
#pragma unroll
            for (int k = 0; k < 8; k++) m = min(m, dep[k]);  // (a bracket that does not exist: kNone)
            if (vm) min8[(base >> 3) + threadIdx.x] = m;
            asm volatile(
                "s_nop 1\n\t"
                "v_min_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1"
                : "+v"(m));
            if ((threadIdx.x & 7u) == 7u && (p0 & ~63u) < nrem) min64[(base >> 6) + (threadIdx.x >> 3)] = m;
            asm volatile(
                "v_min_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                "s_nop 1\n\t"
                "v_min_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                "s_nop 1"
                : "+v"(m));
            if (lane == 63u && wave * 512u < nrem) min512[(base >> 9) + wave] = m;
        }
        __syncthreads();  // (the next round zeroes what (d) read)
        MSJ_CSTAMP(7);
#undef MSJ_CSTAMP
    }
}

// levels 4 and 5 of the min tree over the compact list: the counts come from the device (the grid is laid out for n)
__global__ __launch_bounds__(256) void build_level_compact(const int32_t *__restrict__ in, int32_t *__restrict__ out, int lev,
                                                           const msj_tokens_result *__restrict__ result, const msj_tokens_result *__restrict__ prev) {
    const uint32_t nbrk = bracket_count(result, prev);
    if (lev >= level_number(nbrk)) return;
    const uint32_t o = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n_in = level_entries(nbrk, lev - 1), n_out = level_entries(nbrk, lev);
    if (o >= n_out) return;
    int m = kNone;
    for (uint32_t k = 0; k <= kFanMask; k++) {
        const uint32_t i = (o << kFanShift) + k;
        if (i < n_in) m = min(m, in[i]);
    }
    out[o] = m;
}
// ... and the levels from 6 on in one workgroup, as build_upper_levels
__global__ __launch_bounds__(1024) void build_upper_levels_compact(const MinTree t, const msj_tokens_result *__restrict__ result,
                                                                   const msj_tokens_result *__restrict__ prev) {
    const uint32_t nbrk = bracket_count(result, prev);
    const int nlev = level_number(nbrk);
    for (int k = 6; k < nlev; k++) {  // uniform
        const int32_t *in = t.lv[k - 1];
        int32_t *out = const_cast<int32_t *>(t.lv[k]);
        const uint32_t n_in = level_entries(nbrk, k - 1), n_out = level_entries(nbrk, k);
        for (uint32_t o = threadIdx.x; o < n_out; o += 1024u) {
            int m = kNone;
            for (uint32_t j = 0; j <= kFanMask; j++) {
                const uint32_t i = (o << kFanShift) + j;
                if (i < n_in) m = min(m, in[i]);
            }
            out[o] = m;
        }
        __threadfence_block();
        __syncthreads();
    }
}

}  // namespace msj_tokens

// workspace: 4 int32 per block (aggregates) + start depth + start slot in the list of opening
// brackets per block, then (matching only) the min tree and that list (one uint32 per token at most)
static uint64_t tree_words(uint64_t n) {  // every level padded to a multiple of 8 entries
    uint64_t w = 0, c = n;
    while (c > 8) {
        c = (c + 7) / 8;
        w += (c + 7) & ~7ull;
    }
    return w;
}
static uint64_t super_count(uint64_t n) {
    const uint64_t nb = (n + msj_tokens::kBlock - 1) / msj_tokens::kBlock;
    return nb ? (nb + msj_tokens::kSuper - 1) / msj_tokens::kSuper : 1;
}
// per block: 4 words (depth aggregate) + 4 (document aggregate, for msj_documents_device) + relative start
// depth + relative start slot
static uint64_t block_words(uint64_t n) {
    const uint64_t nb = (n + msj_tokens::kBlock - 1) / msj_tokens::kBlock;
    return (10 * (nb ? nb : 1) + 7u) & ~7ull;
}
static_assert(msj_tokens::kBlock == 2048, "documents_kernel.hip reads these aggregates with its own block size");
extern "C" void *msj_tokens_doc_aggregates(int32_t *d_ws, uint64_t n) {
    const uint64_t nb = (n + msj_tokens::kBlock - 1) / msj_tokens::kBlock;
    return d_ws + 4 * (nb ? nb : 1);
}
static uint64_t head_words(uint64_t n) {  // ... + the same six words per run of kSuper blocks
    return block_words(n) + ((6 * super_count(n) + 7u) & ~7ull);
}
// with_match: 0 = depths only, 1 = + match[] (the min tree over depth[], 64 scratch words, the lists of opening brackets
// left to the tree and their counters), 2 = + pairs[] (round 5: the same over the call's brackets as a compact list -- a
// token word and a depth word each, n of them at most -- which comes on top)
static uint64_t compact_words(uint64_t n) { return (n + 7u) & ~7ull; }
extern "C" uint64_t msj_tokens_workspace_bytes(uint64_t n, int with_match) {
    const uint64_t nb = (n + msj_tokens::kBlock - 1) / msj_tokens::kBlock;
    uint64_t w = head_words(n);
    if (with_match) w += tree_words(n) + 64 + (uint64_t)msj_tokens::kSurvivorShards * msj_tokens::kSurvivorStride;
    if (with_match) w += msj_tokens::kSurvivorShards * msj_tokens::survivor_capacity(nb);
    if (with_match == 2) w += 2 * compact_words(n);
    return w * sizeof(int32_t);
}

// scan of the block aggregates (already in d_ws), depth of every token, bracket partners
static int launch_depth_passes(const uint32_t *d_idx, uint64_t n, uint8_t *d_type, int32_t *d_depth, uint32_t *d_match,
                               msj_tokens_result *d_result, int32_t *d_ws, hipStream_t s, const msj_token_opts &o) {
    using namespace msj_tokens;
    (void)d_idx;
    const uint32_t nb = (uint32_t)((n + kBlock - 1) / kBlock);
    const uint64_t nbs = nb ? nb : 1;
    int32_t *agg = d_ws, *start = d_ws + 8 * nbs;
    uint32_t *open_start = reinterpret_cast<uint32_t *>(d_ws + 9 * nbs);
    uint4 *doc_agg = reinterpret_cast<uint4 *>(d_ws + 4 * nbs);
    int32_t *tree = d_ws + head_words(n);  // 32-byte aligned inside the workspace
    uint2 *d_pairs = reinterpret_cast<uint2 *>(o.d_pairs);
    const bool want_match = (d_match != nullptr || d_pairs != nullptr) && n > 0;
    const bool compact = want_match && d_pairs != nullptr;
    // the workspace behind the tree (msj_tokens_workspace_bytes): [compact: the bracket list, a token and a depth word
    // each,] kSurvivorShards lists, then their counters (zeroed by scan_blocks)
    uint32_t *behind = want_match ? reinterpret_cast<uint32_t *>(tree + tree_words(n) + 64) : nullptr;
    uint32_t *brk_tok = compact ? behind : nullptr;
    int32_t *brk_depth = compact ? reinterpret_cast<int32_t *>(behind + compact_words(n)) : nullptr;
    uint32_t *opens = compact ? behind + 2 * compact_words(n) : behind;
    uint32_t *survivors = want_match ? opens + kSurvivorShards * survivor_capacity(nb) : nullptr;
    const uint32_t nsuper = (uint32_t)super_count(n);
    int32_t *super_agg = d_ws + block_words(n), *super_start = super_agg + 4 * (uint64_t)nsuper;
    uint32_t *super_open = reinterpret_cast<uint32_t *>(super_start + nsuper);
    if (nb) hipLaunchKernelGGL(scan_super, dim3(nsuper), dim3(256), 0, s, agg, nb, start, open_start, super_agg);
    uint32_t *resid = (want_match && !d_pairs) ? o.d_resid : nullptr;
    hipLaunchKernelGGL(scan_blocks, dim3(1), dim3(1024), 0, s, super_agg, nb ? nsuper : 0u, super_start, super_open, d_result, n, o.d_prev,
                       survivors, resid);
    // levels of the min tree (over depth[]; compact: over the bracket list's depth words, laid out for n brackets, the
    // counts of a call known on the device only): 1..3 come out of apply_depth / match_compact, the rest from build_level
    MinTree t;
    t.lv[0] = compact ? brk_depth : d_depth;
    t.cnt[0] = (uint32_t)n;
    t.nlev = 1;
    int32_t *lvl[kMaxLevels] = {nullptr};
    for (int k = 1; k < kMaxLevels; k++) t.lv[k] = nullptr, t.cnt[k] = 0;
    if (want_match) {
        int32_t *p = tree;
        while (t.cnt[t.nlev - 1] > 8 && t.nlev < kMaxLevels) {
            const uint32_t c_out = (t.cnt[t.nlev - 1] + 7u) / 8u;
            lvl[t.nlev] = p;
            t.lv[t.nlev] = p;
            t.cnt[t.nlev] = c_out;
            p += (c_out + 7u) & ~7u;
            t.nlev++;
        }
    }
    // apply_depth writes levels 1..3 unconditionally when asked to: give it scratch for the ones a short input lacks
    int32_t *l1 = want_match ? (t.nlev > 1 ? lvl[1] : tree) : nullptr;
    int32_t *l2 = want_match ? (t.nlev > 2 ? lvl[2] : tree + tree_words(n) + 8) : nullptr;
    int32_t *l3 = want_match ? (t.nlev > 3 ? lvl[3] : tree + tree_words(n) + 40) : nullptr;
    int32_t *const no_level = nullptr;
    uint32_t *const no_brk = nullptr;
    if (nb && compact)
        hipLaunchKernelGGL(depth_rows<true>, dim3((nb + kRowsWaves - 1u) / kRowsWaves), dim3(kThreads), 0, s, d_type, n, nb, start, super_start, super_open,
                           open_start, d_depth, doc_agg, agg, o.d_prev, brk_tok, brk_depth);
    else if (nb && want_match)
        hipLaunchKernelGGL(apply_depth, dim3(nb), dim3(kThreads), 0, s, d_type, n, start, super_start, d_depth, l1, l2, l3, opens, doc_agg, agg, o.d_prev,
                           d_match, survivors, o.match_bias, resid ? 1u : 0u);
    else if (nb)
        hipLaunchKernelGGL(depth_rows<false>, dim3((nb + kRowsWaves - 1u) / kRowsWaves), dim3(kThreads), 0, s, d_type, n, nb, start, super_start, super_open,
                           open_start, d_depth, doc_agg, agg, o.d_prev, no_brk, no_level);
    if (nb) hipLaunchKernelGGL(min_max_depth, dim3(nb < kMinMaxGroups * 256u ? (nb + 255u) / 256u : kMinMaxGroups), dim3(256), 0, s, agg, nb, d_result);
    const uint32_t lists = nb < kSurvivorShards ? nb : kSurvivorShards;
    const uint32_t per_list = nb / kSurvivorShards / 8u + 1u;  // ~2 survivors per block, 16 brackets per workgroup and round
    if (compact) {
        // the brackets' partners on the compact list: in-block pairs, the tree's levels above (grids laid out for n
        // brackets, the kernels take the call's count from the result), the containers that span a block of brackets
        const uint32_t ncb_max = (uint32_t)((n + kCompactBlock - 1) / kCompactBlock);
        hipLaunchKernelGGL(match_compact, dim3(ncb_max < MSJ_COMPACT_GRID ? ncb_max : MSJ_COMPACT_GRID), dim3(256), 0, s, brk_tok, brk_depth, d_result,
                           o.d_prev, d_pairs, l1, l2, l3, opens, survivors, survivor_capacity(nb));
        for (int k = 4; k < t.nlev && k < 6; k++)
            hipLaunchKernelGGL(build_level_compact, dim3((t.cnt[k] + 255u) / 256u), dim3(256), 0, s, t.lv[k - 1], lvl[k], k, d_result, o.d_prev);
        if (t.nlev > 6) hipLaunchKernelGGL(build_upper_levels_compact, dim3(1), dim3(1024), 0, s, t, d_result, o.d_prev);
        hipLaunchKernelGGL(match_brackets<true>, dim3(per_list < MSJ_MATCH_GRID ? per_list : MSJ_MATCH_GRID, lists), dim3(256), 0, s, d_type, opens,
                           survivors, t, d_match, survivor_capacity(nb), 0u, d_result, resid, d_pairs, brk_tok, o.d_prev);
    } else if (want_match) {
        for (int k = 4; k < t.nlev && k < 6; k++)
            hipLaunchKernelGGL(build_level, dim3((t.cnt[k] + 255u) / 256u), dim3(256), 0, s, t.lv[k - 1], t.cnt[k - 1], lvl[k], t.cnt[k]);
        if (t.nlev > 6) {
            UpperLevels u;
            for (int k = 0; k < kMaxLevels; k++) {
                u.lv[k] = k < t.nlev ? const_cast<int32_t *>(t.lv[k]) : nullptr;
                u.cnt[k] = k < t.nlev ? t.cnt[k] : 0u;
            }
            u.first = 6;
            u.nlev = t.nlev;
            hipLaunchKernelGGL(build_upper_levels, dim3(1), dim3(1024), 0, s, u);
        }
        // match[] is complete for every container that closes inside a block (apply_depth wrote the whole array);
        // what is left -- a container that spans a block border, or lies outside the levels a block keeps -- walks the tree
        // (the lane groups of 32 workgroups stride over each of the lists)
        // (block b appends to list b mod kSurvivorShards: a short call uses the first nb lists only, and a list then holds
        // the survivors of nb / kSurvivorShards blocks -- the grid follows, instead of 8 192 workgroups for a handful of tokens)
        hipLaunchKernelGGL(match_brackets<false>, dim3(per_list < MSJ_MATCH_GRID ? per_list : MSJ_MATCH_GRID, lists), dim3(256), 0, s, d_type, opens,
                           survivors, t, d_match, survivor_capacity(nb), o.match_bias, d_result, resid, d_pairs, no_brk, o.d_prev);
        if (resid)  // the closing brackets whose partner lies in front of this call (behind match_brackets: it writes both ends)
            hipLaunchKernelGGL(collect_closers, dim3(per_list < 8u ? per_list : 8u, lists), dim3(256), 0, s, opens, survivors, survivor_capacity(nb),
                               d_depth, d_match, o.d_prev, resid);
    }
    return (int)hipGetLastError();
}

// ---- token spans --------------------------------------------
// For a string token the offset of its closing quote and whether the body holds a backslash -- the
// scan parse_string does byte by byte before it can copy (generic/stage2/string_parsing.mojo:
// 334-386); for a number token the offset one past its last character and whether it is written as a
// float ('.', 'e' or 'E' present) -- what parse_number finds out while it accumulates digits
// (include/generic/number_parsing.mojo:22-80).  One thread per structural.
// The closing quote needs no scan of the body: in stage 1's output the next structural after an
// opening quote is the first byte after the closing quote that is not whitespace (an operator, a
// quote and a scalar that follows a quote are all structural), so the closing quote is the last
// non-blank byte in front of the next structural.  The backslash flag is a scan of the body with
// independent 8-byte reads (bodies over kSpanCap bytes: MSJ_SPAN_LONG, flag left to stage 2).
// DERIVED quantities, like the token stream above: defined by the CPU statement the tests use.
namespace msj_tokens {
constexpr uint32_t kSpanCap = 1024;
constexpr uint64_t k7f = 0x7F7F7F7F7F7F7F7Full;
// bit 7 of every byte of x that is zero; exact (no carries between bytes)
__device__ __forceinline__ uint64_t zero_bytes(uint64_t x) { return ~(((x & k7f) + k7f) | x | k7f); }
__device__ __forceinline__ uint64_t load8(const uint8_t *p) {
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    return ((uint64_t)w.y << 32) | w.x;
}
__device__ __forceinline__ bool is_blank(uint32_t b) { return b == 0x20u || b == 0x09u || b == 0x0Au || b == 0x0Du; }
__device__ __forceinline__ bool is_digit(uint32_t b) { return b - 0x30u < 10u; }
// the reference's structural_or_whitespace table (internal/jsoncharutils_tables.mojo:5-16): 09 0A 0D 20 , : [ ] { }
__device__ __forceinline__ bool is_sow(uint32_t b) {
    return is_blank(b) || b == ',' || b == ':' || b == '[' || b == ']' || b == '{' || b == '}';
}

// where the bytes come from: global memory, or the workgroup's stretch of the buffer staged in LDS
struct FromGlobal {
    const uint8_t *buf;
    uint64_t len;
    __device__ __forceinline__ uint32_t byte(uint64_t pos) const { return buf[pos]; }
    // aligned 8-byte word; bytes past the buffer read as blanks
    __device__ __forceinline__ uint64_t word(uint64_t wa) const {
        if (wa + 8 <= len) return load8(buf + wa);
        uint64_t w = 0x2020202020202020ull;
        for (uint64_t b = wa; b < len; b++) w = (w & ~(0xFFull << (8 * (b - wa)))) | ((uint64_t)buf[b] << (8 * (b - wa)));
        return w;
    }
};
struct FromLds {
    const uint8_t *lds;  // lds[0] = byte `lo` of the buffer; staged [lo, hi_al), blanks past the buffer
    uint64_t lo, hi_al;
    __device__ __forceinline__ uint32_t byte(uint64_t pos) const { return lds[pos - lo]; }
    __device__ __forceinline__ uint64_t word(uint64_t wa) const {
        if (wa + 8 > hi_al) return 0x2020202020202020ull;  // past the next structural: never part of this token
        const uint2 w = *reinterpret_cast<const uint2 *>(lds + (wa - lo));
        return ((uint64_t)w.y << 32) | w.x;
    }
};

// token [start, next): next = offset of the next structural (len for the last token)
template <class Src>
__device__ __forceinline__ void span_of(const Src &src, uint64_t start, uint64_t next, uint64_t len, uint32_t &e, uint32_t &f) {
    const uint32_t c = src.byte(start);
    e = 0;
    f = 0;
    if (c == '"') {
        f = MSJ_SPAN_STRING;
        uint64_t q = next;  // exclusive end of the candidate region
        while (q > start + 1 && is_blank(src.byte(q - 1))) q--;
        bool closed = false;
        if (q > start + 1 && src.byte(q - 1) == '"') {
            uint64_t k = q - 1;  // unescaped iff an even number of backslashes stands right in front of it
            while (k > start + 1 && src.byte(k - 1) == '\\') k--;
            closed = (((q - 1) - k) & 1u) == 0;
        }
        if (closed) {
            const uint64_t close = q - 1;
            const uint64_t b0 = start + 1;  // body = [b0, close)
            if (close - b0 > kSpanCap) {
                f |= MSJ_SPAN_LONG;
            } else if (close > b0) {
                uint64_t acc = 0;
                const uint64_t w0 = b0 & ~7ull, w1 = (close - 1) & ~7ull;  // first and last aligned word
                for (uint64_t w = w0; w <= w1; w += 8) {
                    uint64_t z = zero_bytes(src.word(w) ^ 0x5C5C5C5C5C5C5C5Cull);
                    if (w == w0) z &= ~0ull << (8 * (b0 - w0));
                    if (w == w1) z &= ~0ull >> (8 * (7 - ((close - 1) - w1)));
                    acc |= z;
                }
                if (acc) f |= MSJ_SPAN_ESCAPED;
            }
        } else {
            f |= MSJ_SPAN_OPEN;  // only the last token can be like this (stage 1 reports UNCLOSED_STRING)
        }
        e = closed ? (uint32_t)(q - 1) : (uint32_t)len;
    } else if (c == '-' || (c >= '0' && c <= '9')) {
        // parse_number's scan, include/generic/number_parsing.mojo:41-59: an optional '-', digits, then either
        // one of . e E (a float: it ends at the first structural or blank byte), a structural or blank byte (an
        // integer ends here), or anything else (the reference returns NUMBER_ERROR).  Bytes past the buffer read
        // as blanks.  (The rare path: stretches too long for LDS; byte by byte.)
        f = MSJ_SPAN_NUMBER;
        const uint64_t stop = (start + 1 + kSpanCap < len) ? start + 1 + kSpanCap : len;
        uint64_t j = start + (c == '-' ? 1u : 0u);
        while (j < stop && is_digit(src.byte(j))) j++;
        const uint32_t ch = j < len ? src.byte(j) : 0x20u;
        if (j < stop || j == len) {
            if (ch == '.' || ch == 'e' || ch == 'E') {
                f |= MSJ_SPAN_FLOAT;
                while (j < stop && !is_sow(src.byte(j))) j++;
            } else if (!is_sow(ch)) {
                f |= MSJ_SPAN_BAD;
            }
        }
        if (j == stop && stop < len) {  // kSpanCap characters and still no end
            f = (f & ~MSJ_SPAN_BAD) | MSJ_SPAN_LONG;
            j = 0;
        }
        e = (uint32_t)j;
    }
}

// One thread per structural, kSpanTokens per workgroup.  The workgroup's tokens cover one contiguous
// stretch of the buffer, [idx[first], idx[first of the next workgroup]].  It is staged in LDS, and the
// lane that brings in a 64-byte block classifies it ONCE the way stage 1 does (bit-plane transpose, then
// every class is a few three-input operations on the planes: lane_math.h span_classes): one bit per byte
// for "digit", "structural or blank" (the reference's table, where a float ends), "backslash" and "not
// blank", ~3 operations per byte.  A token then needs a handful of LDS reads and no loop in the common
// case: the digits of a number end at the first clear "digit" bit of a 32-bit window, a float at the
// next set "structural or blank" bit, the closing quote is the top set bit of a "not blank" window that
// ends at the next structural, a string's escape flag is two masked 64-bit words plus a bit per block in
// between ("this block holds a backslash", one ballot per wave).  Work per byte instead of work per
// token times its length.  A stretch over kSpanLds bytes (long strings) takes the per-token path from
// global memory (span_of) instead.
constexpr uint32_t kSpanThreads = 256;
constexpr int kSpanPer = 2;  // tokens per thread
constexpr uint32_t kSpanTokens = kSpanThreads * kSpanPer;  // per workgroup
constexpr uint32_t kSpanLds = 12288;  // + four bitmaps of 1.5 KiB: 8 workgroups (32 waves) per CU
constexpr uint32_t kSpanBlocks = kSpanLds / 64;
// a bitmap in LDS: two zero words, one bit per staged byte, zero words behind (windows reach one word
// in front of position 0 and two words past the last staged byte)
constexpr uint32_t kSpanMapFront = 2;
constexpr uint32_t kSpanMapWords = kSpanMapFront + 2 * kSpanBlocks + 4;
static_assert(kSpanBlocks <= kSpanThreads, "one lane per staged block");

// Tokens the span kernel could not finish from its staged bytes (staged_token returns true: see there) go on a
// work list in global memory -- fix[0] counts them, fix[2 ..] holds their token indices, fix[1] counts the
// workgroups of span_fixup that are done -- and get the sentinel 0xFF as their flags; span_fixup, launched behind
// the span kernel, works them out from global memory.  Appending is one atomic and one store in a block that
// almost no wave enters (a call or an inlined scan from global memory at that place cost the kernel 5 %).
constexpr uint32_t kFixCap = 16382;        // entries; more than that: span_fixup looks for the sentinel itself (MSJ_SPANS_FIX_CAP: tests)
constexpr uint32_t kFixWords = kFixCap + 2;
constexpr uint32_t kFixSentinel = 0xFFu;   // no token has all eight flag bits
constexpr uint32_t kFixGroups = 256;       // (x 4 waves: a wave per listed long string)
__device__ __forceinline__ void fix_later(uint32_t *fix, uint32_t cap, uint32_t token, uint32_t &e, uint32_t &f) {
    const uint32_t slot = atomicAdd(&fix[0], 1u);
    if (slot < cap) fix[2 + slot] = token;
    e = 0;
    f = kFixSentinel;
}

// Strings whose body is longer than kSpanCap (round 5: the escape flag is exact at ANY length; MSJ_SPAN_LONG is left
// to numbers).  The span kernels know such a string's closing quote -- the last non-blank byte in front of the next
// structural -- but not whether its body holds a backslash: that is what parse_string rescans the most bytes for
// (generic/stage2/string_parsing.mojo:334-386, include/haswell/stringparsing_defs.mojo:27-48).  They give it the flags
// MSJ_SPAN_STRING | MSJ_SPAN_LONG as a MARKER and put it on a second work list behind the fix-up list (same buffer);
// span_fixup scans the body of every listed string with one WAVE (1 KiB contiguous per load instruction), strings
// over kBigBody bytes go on to long_strings_big, which scans each with the whole grid.  A list that overflows:
// span_fixup finds the marker in flags[] itself.  Work per byte of long strings, whatever their length.
constexpr uint32_t kLongString = MSJ_SPAN_STRING | MSJ_SPAN_LONG;
constexpr uint32_t kLngHdr = 4;             // [0] entries, [2] strings over kBigBody, [3] workgroups of long_strings_big that are done
constexpr uint32_t kBigBody = 1u << 20;     // bytes one wave scans at most
constexpr uint32_t kBigCap = 4096;          // a segment (< 4 GiB) holds fewer strings of more than 1 MiB
constexpr uint32_t kBigPiece = 1u << 16;    // bytes per wave and step of long_strings_big
constexpr uint32_t kLngWords = kLngHdr + kFixCap + 2 * kBigCap;  // header, token indices, the big ones' tokens, their results
__device__ __forceinline__ void long_later(uint32_t *fix, uint32_t cap, uint32_t token) {
    uint32_t *lng = fix + kFixWords;
    const uint32_t slot = atomicAdd(&lng[0], 1u);
    if (slot < cap) lng[kLngHdr + slot] = token;
}
// 0x80 in every byte of x that is zero; exact (no carries between bytes)
__device__ __forceinline__ uint32_t zero_bytes32(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }
// any backslash in buf[b0, close)?  One wave; 1 KiB contiguous per load instruction, four of them per round trip;
// the 16-byte pieces that straddle b0 or close are looked at byte by byte.  close <= len; buf 16-byte aligned.
__device__ __forceinline__ bool wave_has_backslash(const uint8_t *__restrict__ buf, uint64_t b0, uint64_t close, uint32_t lane) {
    uint32_t any = 0;
    constexpr uint32_t kBs = 0x5C5C5C5Cu;
    for (uint64_t base = b0 & ~15ull; base < close; base += 4096u) {  // uniform
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint64_t off = base + 1024u * k + 16u * lane;
            if (off < close) {
                if (off >= b0 && off + 16u <= close) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(buf + off);
                    any |= zero_bytes32(v.x ^ kBs) | zero_bytes32(v.y ^ kBs) | zero_bytes32(v.z ^ kBs) | zero_bytes32(v.w ^ kBs);
                } else {
                    for (uint32_t j = 0; j < 16u; j++) {
                        const uint64_t p = off + j;
                        if (p >= b0 && p < close && buf[p] == 0x5Cu) any = 1u;
                    }
                }
            }
        }
        if (__ballot(any != 0u) != 0ull) return true;  // uniform
    }
    return false;
}

// 32 bits of a bitmap starting at bit `pos` (bit 0 of the result = position pos)
__device__ __forceinline__ uint32_t bits_at(const uint32_t *map, uint32_t pos) {
    const uint32_t w = (pos >> 5) + kSpanMapFront;
    return __funnelshift_r(map[w], map[w + 1], pos & 31u);
}
// the 32 bits in front of position t (bit 31 of the result = position t - 1); positions below 0 read as 0
__device__ __forceinline__ uint32_t bits_before(const uint32_t *map, uint32_t t) {
    const uint32_t w = (t >> 5) + kSpanMapFront - 1u;
    return __funnelshift_r(map[w], map[w + 1], t & 31u);
}
__device__ __forceinline__ uint64_t block_bits(const uint32_t *map, uint32_t blk) {
    const uint2 v = *reinterpret_cast<const uint2 *>(map + kSpanMapFront + 2u * blk);
    return ((uint64_t)v.y << 32) | v.x;
}

// one token from the staged stretch; offsets relative to lo (< kSpanLds).  Returns true when the staged bytes were not
// enough: a float's scan runs `while not structural-or-blank`, and where the next structural is a scalar that follows
// a quote (`1.5"a"b`: no blank or operator in front of it) that scan goes on past the next structural and may leave
// the stretch -- the caller then puts the token on the work list of span_fixup.
__device__ __forceinline__ bool staged_token(const uint8_t *stage, const uint32_t *m_num, const uint32_t *m_flt, const uint32_t *m_bs,
                                             const uint32_t *m_ink, const uint32_t *bs_blocks, uint64_t lo, uint32_t span, uint64_t len,
                                             uint32_t c, uint64_t start, uint64_t next, uint32_t &e_out, uint32_t &f_out) {
    const uint32_t rs = (uint32_t)(start - lo);
    const uint32_t rn = (uint32_t)(next - lo);
    const uint32_t rlen = (len - lo < (uint64_t)span) ? (uint32_t)(len - lo) : span;  // end of the buffer within the stretch
    uint32_t e = 0, f = 0;
    if (c == '"') {
        f = MSJ_SPAN_STRING;
        // exclusive end of the candidate region: behind the last non-blank byte in (rs, rn) -- pretty-printed
        // input has a line break and its indentation between a value and the closing bracket
        uint32_t q = rs + 1;
        if (rn > rs + 1) {
            uint32_t t = rn;
            for (;;) {  // one round unless more than 32 blanks stand in front of the next structural
                const uint32_t d = t - (rs + 1);
                uint32_t v = bits_before(m_ink, t);
                if (d < 32) v &= ~0u << (32u - d);
                if (v) {
                    q = t - __clz(v);
                    break;
                }
                if (d <= 32) break;
                t -= 32;
            }
        }
        bool closed = false;
        if (q > rs + 1 && stage[q - 1] == '"') {
            // unescaped iff an even number of backslashes stands right in front of it (the byte at rs is the
            // opening quote, so the run ends there at the latest)
            uint32_t run = 0, t = q - 1;
            for (;;) {  // one round unless 32 backslashes in a row
                const uint32_t v = ~bits_before(m_bs, t);
                const uint32_t r = v ? (uint32_t)__clz(v) : 32u;
                run += r;
                if (r < 32) break;
                t -= 32;
            }
            closed = (run & 1u) == 0;
        }
        if (closed) {
            const uint32_t b0 = rs + 1, close = q - 1;  // body = [b0, close)
            if (close - b0 > kSpanCap) {
                f |= MSJ_SPAN_LONG;
            } else if (close > b0) {
                const uint32_t last = close - 1, k0 = b0 >> 6, k1 = last >> 6;
                const uint64_t head = block_bits(m_bs, k0) & (~0ull << (b0 & 63u));
                const uint64_t tail = block_bits(m_bs, k1) & (~0ull >> (63u - (last & 63u)));
                bool any;
                if (k0 == k1) {
                    any = (head & tail) != 0;
                } else {  // blocks strictly between: at most kSpanCap / 64 + 1 of them, one bit each
                    const uint32_t between = k1 - k0 - 1u;
                    const uint32_t w = (k0 + 1u) >> 5;
                    const uint32_t mid = __funnelshift_r(bs_blocks[w], bs_blocks[w + 1], (k0 + 1u) & 31u) & ((1u << between) - 1u);
                    any = (head | tail | mid) != 0;
                }
                if (any) f |= MSJ_SPAN_ESCAPED;
            }
        } else {
            f |= MSJ_SPAN_OPEN;  // only the last token can be like this (stage 1 reports UNCLOSED_STRING)
        }
        e = closed ? (uint32_t)(lo + (q - 1)) : (uint32_t)len;
    } else if (c == '-' || (c >= '0' && c <= '9')) {
        // parse_number's scan (include/generic/number_parsing.mojo:41-59) on the bitmaps: optional '-', digits
        // up to the first clear "digit" bit; then . e E -> float, ends at the next "structural or blank" bit;
        // a structural or blank byte -> integer, ends here; anything else -> the reference's NUMBER_ERROR.
        // Everything up to the next structural is staged; past the end of the buffer (rlen) reads as blank.
        f = MSJ_SPAN_NUMBER;
        const uint32_t stop = (rs + 1 + kSpanCap < rlen) ? rs + 1 + kSpanCap : rlen;
        uint32_t p = rs + (c == '-' ? 1u : 0u);
        while (p < stop) {
            const uint32_t run = ~bits_at(m_num, p);
            if (run) {
                p += __ffs(run) - 1u;
                break;
            }
            p += 32;
        }
        if (p > stop) p = stop;
        if (p < stop || lo + p == len) {
            const uint32_t ch = p < rlen ? stage[p] : 0x20u;
            if (ch == '.' || ch == 'e' || ch == 'E') {
                f |= MSJ_SPAN_FLOAT;
                while (p < stop) {
                    const uint32_t hit = bits_at(m_flt, p);
                    if (hit) {
                        p += __ffs(hit) - 1u;
                        break;
                    }
                    p += 32;
                }
                if (p > stop) p = stop;
            } else if (p < rlen && !((bits_at(m_flt, p)) & 1u)) {
                f |= MSJ_SPAN_BAD;
            }
        }
        if (p == stop && lo + stop < len) {
            if (stop < rs + 1 + kSpanCap) return true;  // not the cap: the stretch ended
            f = (f & ~MSJ_SPAN_BAD) | MSJ_SPAN_LONG;      // kSpanCap characters and still no end
        } else {
            e = (uint32_t)(lo + p);
        }
    }
    e_out = e;
    f_out = f;
    return false;
}

// bits of m_bs in front of position pos, counted from the start of the 4 KiB the wave that staged pos covers
// (bs_cnt: set bits in front of each 32-bit word, per wave)
__device__ __forceinline__ uint32_t bs_before(const uint32_t *m_bs, const uint16_t *bs_cnt, uint32_t pos) {
    const uint32_t w = (pos >> 5) + kSpanMapFront;
    return (uint32_t)bs_cnt[w] + (uint32_t)__popc(m_bs[w] & ((1u << (pos & 31u)) - 1u));
}

// staged_token without a loop or a branch: every window is read once and both kinds of token are worked out
// for every lane (a wave holds strings and numbers anyway).  Returns false where one round was not enough -- a
// window came up empty, or a string body crosses the 4 KiB one wave counted backslashes over: such a lane
// takes staged_token.  Positions relative to lo, 32 bits throughout (a segment is < 4 GiB).
__device__ __forceinline__ bool staged_token_fast(const uint8_t *stage, const uint32_t *m_num, const uint32_t *m_flt, const uint32_t *m_bs,
                                                  const uint16_t *bs_cnt, const uint32_t *m_ink, uint32_t lo, uint32_t span, uint32_t len,
                                                  uint32_t c, uint32_t rs, uint32_t rn, uint32_t &e_out, uint32_t &f_out) {
    const uint32_t rlen = min(len - lo, span);  // end of the buffer within the stretch
    const bool is_str = c == '"';
    const bool is_num = c == '-' || c - '0' < 10u;
    // ---- string: behind the last non-blank byte in front of the next structural (the byte at rs is not blank, so
    // a window that reaches rs is never empty; one that does not and is empty needs another round)
    const uint32_t vi = bits_before(m_ink, rn);
    const uint32_t q = (uint32_t)max((int)rn - __clz(vi), (int)rs + 1);
    const bool more_ink = vi == 0 && rn - rs > 33u;
    const bool isq = q > rs + 1u && stage[q - 1u] == '"';
    const uint32_t vb = ~bits_before(m_bs, q - 1u);  // backslashes right in front of that quote
    const bool closed = isq && ((uint32_t)__clz(vb) & 1u) == 0;
    const bool more_bs = isq && vb == 0;
    const uint32_t b0 = rs + 1u, close = q - 1u;  // body = [b0, close)
    const bool esc = bs_before(m_bs, bs_cnt, close) != bs_before(m_bs, bs_cnt, b0);
    const bool far = closed && (b0 >> 12) != (close >> 12);
    const bool lng = close - b0 > kSpanCap;
    const uint32_t f_str = MSJ_SPAN_STRING | (closed ? (lng ? MSJ_SPAN_LONG : (esc ? MSJ_SPAN_ESCAPED : 0u)) : MSJ_SPAN_OPEN);
    const uint32_t e_str = closed ? lo + close : len;
    // ---- number: parse_number's scan (include/generic/number_parsing.mojo:41-59), see staged_token
    const uint32_t stop = min(rs + 1u + kSpanCap, rlen);
    const uint32_t p0 = rs + (c == '-' ? 1u : 0u);
    const uint32_t wn = ~bits_at(m_num, p0);
    const uint32_t p = min(p0 + min((uint32_t)(__ffs(wn) - 1), 32u), stop);
    const bool more_num = wn == 0 && p0 + 32u < stop;
    const bool ends = p < stop || lo + p == len;
    const uint32_t ch = p < rlen ? (uint32_t)stage[p] : 0x20u;
    const bool flt = ends && (ch == '.' || (ch | 0x20u) == 'e');
    const uint32_t wf = bits_at(m_flt, p);
    const bool bad = ends && !flt && p < rlen && !(wf & 1u);
    const uint32_t pe = flt ? min(p + min((uint32_t)(__ffs(wf) - 1), 32u), stop) : p;
    const bool more_flt = flt && wf == 0 && p + 32u < stop;
    const bool no_end = pe == stop && lo + stop < len;  // kSpanCap characters and still no end, or the stretch ended: staged_token
    const uint32_t f_num = MSJ_SPAN_NUMBER | (flt ? MSJ_SPAN_FLOAT : 0u) | (bad ? MSJ_SPAN_BAD : 0u);
    const uint32_t e_num = lo + pe;
    e_out = is_str ? e_str : (is_num ? e_num : 0u);
    f_out = is_str ? f_str : (is_num ? f_num : 0u);
    return !(is_str ? (more_ink || more_bs || far || (closed && lng)) : (is_num && (more_num || more_flt || no_end)));
}

// kFused: the kernel also writes the type byte of every token and the (sum, min, max, opening brackets)
// aggregate of its kSpanTokens tokens (what the token pre-pass scans); kSpans = false: those only -- no classes, no
// bitmaps, no token evaluation, no ends and flags (msj_tokens_device).
template <bool kFused, bool kSpans = true>
__global__ __launch_bounds__(kSpanThreads) void token_spans(const uint8_t *__restrict__ buf, uint64_t len, const uint32_t *__restrict__ idx,
                                                           uint64_t n, uint32_t *__restrict__ end, uint8_t *__restrict__ flags, uint32_t lds_limit,
                                                           uint8_t *__restrict__ type, int4 *__restrict__ sub_agg, uint32_t *__restrict__ fix,
                                                           uint32_t fix_cap) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kSpanLds + 16];  // a number may be asked for the byte behind the stretch
    __shared__ __attribute__((aligned(8))) uint32_t m_num[kSpanMapWords], m_flt[kSpanMapWords], m_bs[kSpanMapWords], m_ink[kSpanMapWords];
    __shared__ __attribute__((aligned(4))) uint16_t bs_cnt[kSpanMapWords];  // set bits of m_bs in front of each word, from the wave's first block
    __shared__ uint32_t bs_blocks[2 * (kSpanThreads / 64) + 2];  // bit b: block b of the stretch holds a backslash
    const uint32_t nt = (uint32_t)n;  // n < 2^31 (msj_token_spans_device)
    const uint32_t first = blockIdx.x * kSpanTokens;
    // this thread's tokens: two neighbours.  In a valid document a scalar is followed by an operator, so at most one
    // of the two is a string or a number and ONE evaluation serves the pair (two scalars in a row: a second one).
    const uint32_t tok0 = first + 2u * threadIdx.x;
    // uniform: every lane has both tokens and a structural behind them, and the arrays take two tokens per access
    const bool full = first + kSpanTokens < nt && (reinterpret_cast<uintptr_t>(end) & 7u) == 0 &&
                      ((reinterpret_cast<uintptr_t>(flags) | reinterpret_cast<uintptr_t>(type)) & 1u) == 0;
    const bool have0 = full || tok0 < nt, have1 = full || tok0 + 1 < nt;
    // Every load below is unconditional (indices clamped to the last token, the value dropped afterwards): the
    // requests for the stretch bounds, for this thread's tokens and -- next -- for the bytes are in flight together
    // instead of one round trip after the other.
    const uint32_t last = nt - 1u;  // nt >= 1: the kernel is not launched for an empty index
    const bool more = first + kSpanTokens < nt;  // a workgroup follows
    // uniform: the stretch [lo, hi) -- through the byte at the next workgroup's first structural
    const uint32_t i_lo = idx[first], i_hi = idx[min(first + kSpanTokens, last)];
    uint32_t i0 = idx[min(tok0, last)], i1 = idx[min(tok0 + 1u, last)], i2 = idx[min(tok0 + 2u, last)];
    const uint64_t lo = (uint64_t)i_lo & ~63ull;
    const uint64_t hi = more ? (uint64_t)i_hi + 1u : len;
    const uint64_t hi_al = (hi + 63u) & ~63ull;
    // (the token offsets are first looked at behind MSJ_SPAN_ARRIVED: nothing waits for them before the bytes are asked for)
#define MSJ_SPAN_ARRIVED()                                     \
    asm volatile("" : "+v"(i0), "+v"(i1), "+v"(i2));           \
    const uint64_t start0 = have0 ? i0 : 0u, start1 = have1 ? i1 : 0u; \
    const uint64_t next1 = (full || tok0 + 2u < nt) ? (uint64_t)i2 : len; \
    const uint64_t next0 = have1 ? start1 : len
    const bool staged = hi_al - lo <= lds_limit;  // lds_limit <= kSpanLds; uniform
    uint32_t e0 = 0, f0 = 0, c0 = 0, e1 = 0, f1 = 0, c1 = 0;
    if (!staged) {
        MSJ_SPAN_ARRIVED();
        if (have0) {
            if (kSpans) {
                span_of(FromGlobal{buf, len}, start0, next0, len, e0, f0);
                if (f0 == kLongString) long_later(fix, fix_cap, tok0);
            }
            if (kFused) c0 = buf[start0];
        }
        if (have1) {
            if (kSpans) {
                span_of(FromGlobal{buf, len}, start1, next1, len, e1, f1);
                if (f1 == kLongString) long_later(fix, fix_cap, tok0 + 1u);
            }
            if (kFused) c1 = buf[start1];
        }
    } else {
        const uint32_t span = (uint32_t)(hi_al - lo);
        const uint32_t nblk = span >> 6;  // <= kSpanBlocks: lane j stages and classifies block j
        const uint32_t j = threadIdx.x;
        uint64_t has_bs = 0;
        uint32_t bs_lo = 0, bs_all = 0;  // backslashes in the low word / in all of this lane's block
        if (j < nblk) {
            const uint64_t g = lo + 64u * j;
            uint32_t x[16];
            if (g + 64 <= len) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(buf + g + 16 * q);
                    x[4 * q] = v.x, x[4 * q + 1] = v.y, x[4 * q + 2] = v.z, x[4 * q + 3] = v.w;
                }
            } else {  // the buffer ends inside this block: blanks behind it
#pragma unroll
                for (int d = 0; d < 16; d++) {
                    uint32_t w = 0x20202020u;
#pragma unroll
                    for (int bb = 0; bb < 4; bb++) {
                        const uint64_t pos = g + 4u * d + bb;
                        if (pos < len) w = (w & ~(0xFFu << (8 * bb))) | ((uint32_t)buf[pos] << (8 * bb));
                    }
                    x[d] = w;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                *reinterpret_cast<uint4 *>(stage + 64u * j + 16 * q) = make_uint4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
            if (kSpans) {
            uint64_t pl[8];
#if MSJ_SPAN_ABLATE == 3
            for (int q = 0; q < 8; q++) pl[q] = x[q] | ((uint64_t)x[q + 8] << 32);
#else
            msj::bitplanes(x, pl);
#endif
            const msj::SpanClasses cl = msj::span_classes(pl);
            const uint32_t w = kSpanMapFront + 2u * j;
            *reinterpret_cast<uint2 *>(m_num + w) = make_uint2((uint32_t)cl.digit, (uint32_t)(cl.digit >> 32));
            *reinterpret_cast<uint2 *>(m_flt + w) = make_uint2((uint32_t)cl.sow, (uint32_t)(cl.sow >> 32));
            *reinterpret_cast<uint2 *>(m_bs + w) = make_uint2((uint32_t)cl.backslash, (uint32_t)(cl.backslash >> 32));
            *reinterpret_cast<uint2 *>(m_ink + w) = make_uint2((uint32_t)~cl.blank, (uint32_t)(~cl.blank >> 32));
            has_bs = cl.backslash;
            bs_lo = __popc((uint32_t)cl.backslash);
            bs_all = bs_lo + __popc((uint32_t)(cl.backslash >> 32));
            }
        } else if (kSpans && j < nblk + 2u) {  // zero words behind the maps
            const uint32_t w = kSpanMapFront + 2u * j;
            *reinterpret_cast<uint2 *>(m_num + w) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_flt + w) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_bs + w) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_ink + w) = make_uint2(0, 0);
        }
        if (kSpans && j == kSpanThreads - 1) {  // ... and in front of them
            *reinterpret_cast<uint2 *>(m_num) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_flt) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_bs) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_ink) = make_uint2(0, 0);
            bs_blocks[2 * (kSpanThreads / 64)] = 0;
            bs_blocks[2 * (kSpanThreads / 64) + 1] = 0;
        }
        if (kSpans) {  // all lanes: blocks in front of this lane's, within the wave
            const uint32_t upto = wave_incl_sum(bs_all) - bs_all;
            if (j < nblk + 2u)
                *reinterpret_cast<uint32_t *>(bs_cnt + kSpanMapFront + 2u * j) = upto | ((upto + bs_lo) << 16);
            if (j == kSpanThreads - 1) *reinterpret_cast<uint32_t *>(bs_cnt) = 0;
        }
        const uint64_t bsb = __ballot(has_bs != 0);
        if (kSpans && (j & 63u) == 0) {
            bs_blocks[2 * (j >> 6)] = (uint32_t)bsb;
            bs_blocks[2 * (j >> 6) + 1] = (uint32_t)(bsb >> 32);
        }
        MSJ_SPAN_ARRIVED();
        __syncthreads();
        if (!kSpans || MSJ_SPAN_ABLATE == 1) {  // the type bytes only
            if (have0) {
                c0 = stage[(uint32_t)(start0 - lo)];
                c1 = have1 ? (uint32_t)stage[(uint32_t)(start1 - lo)] : 0u;
            }
        } else if (have0) {
            const uint32_t lo32 = (uint32_t)lo, len32 = (uint32_t)len;
            const uint32_t rs0 = (uint32_t)(start0 - lo), rn0 = (uint32_t)(next0 - lo);
            const uint32_t rs1 = have1 ? (uint32_t)(start1 - lo) : rs0, rn1 = have1 ? (uint32_t)(next1 - lo) : rn0;
            c0 = stage[rs0];
            c1 = have1 ? (uint32_t)stage[rs1] : 0u;
            const bool s0 = c0 == '"' || c0 == '-' || c0 - '0' < 10u;            // a string or a number: something to work out
            const bool s1 = have1 && (c1 == '"' || c1 == '-' || c1 - '0' < 10u);
            const uint32_t cs = s0 ? c0 : c1, rs = s0 ? rs0 : rs1, rn = s0 ? rn0 : rn1;
            uint32_t e, f;
            if (!staged_token_fast(stage, m_num, m_flt, m_bs, bs_cnt, m_ink, lo32, span, len32, cs, rs, rn, e, f)) {
                if (staged_token(stage, m_num, m_flt, m_bs, m_ink, bs_blocks, lo, span, len, cs, lo + rs, lo + rn, e, f))
                    fix_later(fix, fix_cap, tok0 + (s0 ? 0u : 1u), e, f);
                else if (f == kLongString)
                    long_later(fix, fix_cap, tok0 + (s0 ? 0u : 1u));
            }
            e0 = s0 ? e : 0u, f0 = s0 ? f : 0u;
            e1 = s0 ? 0u : e, f1 = s0 ? 0u : f;
            if (s0 && s1) {  // two scalars in a row (not a valid document)
                if (!staged_token_fast(stage, m_num, m_flt, m_bs, bs_cnt, m_ink, lo32, span, len32, c1, rs1, rn1, e1, f1)) {
                    if (staged_token(stage, m_num, m_flt, m_bs, m_ink, bs_blocks, lo, span, len, c1, start1, next1, e1, f1))
                        fix_later(fix, fix_cap, tok0 + 1u, e1, f1);
                    else if (f1 == kLongString)
                        long_later(fix, fix_cap, tok0 + 1u);
                }
            }
        }
    }
    if (full) {
        if (kSpans) {  // write-once streams: non-temporal stores
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 ee = {e0, e1};
            __builtin_nontemporal_store(ee, reinterpret_cast<u32x2 *>(end + tok0));
            __builtin_nontemporal_store((uint16_t)(f0 | (f1 << 8)), reinterpret_cast<uint16_t *>(flags + tok0));
        }
        if (kFused) *reinterpret_cast<uint16_t *>(type + tok0) = (uint16_t)(c0 | (c1 << 8));
    } else if (have0) {
        if (kSpans) {
            end[tok0] = e0;
            flags[tok0] = (uint8_t)f0;
        }
        if (kFused) type[tok0] = (uint8_t)c0;
        if (have1) {
            if (kSpans) {
                end[tok0 + 1] = e1;
                flags[tok0 + 1] = (uint8_t)f1;
            }
            if (kFused) type[tok0 + 1] = (uint8_t)c1;
        }
    }
    if (kFused && MSJ_SPAN_ABLATE != 2) {
        // the bracket counts of this wave's 128 tokens = chunk 4 * blockIdx.x + wave of the grid token_tiles uses (same
        // downstream: merge_chunk_counts, the scans, apply_depth -- which supplies the minimum / maximum of the running
        // depth that this kernel reduced with two DPP chains, an LDS hand-over and a second barrier per workgroup until
        // round 3).  '[' '{' are 5B 7B, ']' '}' are 5D 7D: one masked compare each, the compare IS the ballot
        const uint32_t k0 = have0 ? (c0 & 0xDFu) : 0u, k1 = have1 ? (c1 & 0xDFu) : 0u;
        const uint64_t up0 = __ballot(k0 == 0x5Bu), dn0 = __ballot(k0 == 0x5Du), up1 = __ballot(k1 == 0x5Bu), dn1 = __ballot(k1 == 0x5Du);
        const uint32_t chunk = 4u * blockIdx.x + (threadIdx.x >> 6);
        if ((threadIdx.x & 63u) == 0 && chunk * 128u < nt) {
            const int ups = (int)__popcll(up0) + (int)__popcll(up1), downs = (int)__popcll(dn0) + (int)__popcll(dn1);
            sub_agg[chunk] = make_int4(ups - downs, kNone, -kNone, ups);
        }
    }
}

#undef MSJ_SPAN_ARRIVED

// Test hooks (msj_token_opts, kept per context by msj_debug_set_span_limits; 0xFFFFFFFF = the built-in value): stretches
// over lds_limit bytes take the global-memory path, the fix-up list holds fix_cap entries.  No environment variable is read.
static uint32_t fix_cap(const msj_token_opts &o) {  // host: the list's capacity for this launch
    return o.fix_cap < kFixCap ? o.fix_cap : kFixCap;
}
static uint32_t span_lds_limit(const msj_token_opts &o) { return o.lds_limit < kSpanLds ? o.lds_limit : kSpanLds; }

// ---- the same results from TILES of the buffer (round 3) -------------------------------------------------------
// token_spans above is organised by tokens: a workgroup first loads idx[first], idx[first + 512] to learn WHICH bytes
// it needs and only then asks for them -- two memory round trips one after the other in a workgroup that lives for
// 4-5 us, with the classification running on the ~42 lanes the stretch has blocks for.  Here the unit is a GROUP of
// kTgTiles 4 KiB tiles of the buffer: which bytes a workgroup stages is known from blockIdx alone, so the bytes, the
// group's entry of a small table (`tbl[g]` = number of structurals in front of byte g * kTgBytes: group_table below)
// and then the first chunk's indices are all requested before anything is waited for; one wave per tile stages and
// classifies it with all 64 lanes (1 KiB-contiguous wave loads, linear in LDS, the lane's own 64-byte block read back
// from there), one more wave does the same for a halo of kTgHaloBlocks blocks behind the group.  Tokens are handled in
// CHUNKS of 128 on a global grid (chunk c = tokens [128 c, 128 c + 128), two per lane, the pair evaluated once --
// pair_fast and its fallbacks); a group owns the chunks whose FIRST token lies in its kTgBytes,
// its waves take them round robin, the next chunk's indices are requested before the present one is worked on.
// A chunk whose tokens end inside the staged range (the halo is what the group's last chunk usually needs) reads
// LDS; one that does not (sparse input, long strings) takes the per-token path from global memory, like a long
// stretch above.  The depth aggregates leave per chunk (merge_chunk_counts folds 16 of them into a block).
// Three tiles and the halo = FOUR waves per workgroup, one per SIMD.  Measured on 1 GiB minified, same box, alternating
// (profiles/r03/prep_tiles_history.txt): 2 / 3 / 4 / 6 tiles per group = 0.90 / 0.87 / 1.04 / 1.00 ms per call of
// msj_stage2_prep_device -- five waves (4 tiles) load one SIMD twice as much as the others, and the smaller group's
// LDS (25 KB) lets six workgroups = 24 waves onto a CU instead of four = 20.  Halo 1 KiB against 2 KiB: no difference on minified; the larger one keeps
// more last chunks of a group on the LDS path where tokens are sparse.
#ifndef MSJ_TG_TILES
#define MSJ_TG_TILES 3
#endif
#ifndef MSJ_TG_HALO_BLOCKS
#define MSJ_TG_HALO_BLOCKS 32
#endif
constexpr uint32_t kTgTiles = MSJ_TG_TILES;
constexpr uint32_t kTgHaloBlocks = MSJ_TG_HALO_BLOCKS;  // 2 KiB behind them (<= 64: one wave)
#ifdef MSJ_TG_OVERLAP
// EXPERIMENT (round 5, measured slower, profiles/r05/token_tiles_parts.txt): no wave of its own for the halo -- every wave
// stages and classifies a full tile, the groups advance by the staged range less the halo, which the next group classifies
// again as the start of its first tile (redundant work halo / advance instead of one wave in kTgTiles + 1)
constexpr uint32_t kTgWaves = kTgTiles;
constexpr uint32_t kTgBytes = kTgTiles * 4096u - kTgHaloBlocks * 64u;
constexpr uint32_t kTgBlocks = kTgTiles * 64;
#else
constexpr uint32_t kTgWaves = kTgTiles + 1;            // one wave per tile + one for the halo
constexpr uint32_t kTgBytes = kTgTiles * 4096u;        // bytes of the buffer per workgroup
constexpr uint32_t kTgBlocks = kTgTiles * 64 + kTgHaloBlocks;
#endif
constexpr uint32_t kTgThreads = 64 * kTgWaves;
constexpr uint32_t kTgStage = kTgBlocks * 64;          // bytes staged
constexpr uint32_t kTgMapWords = kSpanMapFront + 2 * kTgBlocks + 4;
constexpr uint32_t kChunk = 128;                       // tokens per wave iteration
static_assert(kTgHaloBlocks <= 64 && (kTgHaloBlocks * 64) % 1024 == 0, "the halo is staged by one wave, 1 KiB per instruction");
static_assert(kBlock % kChunk == 0, "a block of the depth pass is a whole number of chunks");

// '"', '-' or a digit, as a 0 / 1 word without a compare chain: bit (c - 0x22) of a 24-bit table
__device__ __forceinline__ uint32_t starts_scalar(uint32_t c) {
    const uint32_t t = c - 0x22u;
    return (uint32_t)(t < 24u) & (0x00FFC801u >> (t & 31u));
}

// ---- token evaluation of token_tiles.  Its bitmaps (one bit per staged byte, layout as above):
//   m_num  digit                      m_flt  structural or blank (the reference's table: where a float ends)
//   m_dot  . e E                      m_ink  not blank
//   m_q    UNESCAPED quote: the escape scanner of stage 1 (json_escape_scanner.mojo:18-45) run per block with the
//          carries resolved across the lanes of the tile's wave, the carry into a tile from the bytes in front of it
//   m_e    unescaped quote with a backslash between it and the unescaped quote in front of it (or the start of its
//          tile): (backslash + ~quote) & quote -- the carry of the addition runs from a backslash through everything
//          that is not a quote and is absorbed by the next quote -- again with the carries resolved across the lanes;
//          co[t] = that carry at the end of tile t ("a backslash since the tile's last quote").
// With them a string token needs three windows that all END at the next structural (known before anything is read:
// one LDS round trip): the closing quote is the last non-blank byte in front of the next structural (top set bit of
// the "ink" window), the string is closed iff the same bit of the m_q window is set, and its body holds a backslash
// iff the same bit of the m_e window is set (or, where the body crosses into the next tile, co[] of the opening
// quote's tile is).  A number needs three windows that all START at its first digit.  Both are evaluated for every
// lane (a wave holds strings and numbers anyway), without a branch.  Nonzero return: a window came up short (32
// blanks in front of the next structural, 32 digits, no end of a float within the window, the last bytes of the
// staged range): the lane takes tile_token_slow.
struct TileMaps {
    const uint8_t *stage;
    const uint32_t *m_num, *m_flt, *m_dot, *m_ink, *m_q, *m_e, *co;
};
__device__ __forceinline__ uint32_t pair_fast(const TileMaps &t, uint32_t lo, uint32_t span, uint32_t len, uint32_t c, uint32_t is_num,
                                              uint32_t rs, uint32_t rn, uint32_t &e_out, uint32_t &f_out) {
    const uint32_t rlen = min(len - lo, span);  // end of the buffer within the staged range
    const uint32_t is_str = c == '"';
    // every read is issued here, before anything is looked at, and pinned: left to itself the compiler moves the
    // reads whose results only one kind of token needs behind a branch on that kind -- a second LDS round trip
    const uint32_t p0 = rs + (uint32_t)(c == '-');
    const uint32_t wa = (rn >> 5) + kSpanMapFront - 1u, wb = (p0 >> 5) + kSpanMapFront;
    uint32_t i_lo = t.m_ink[wa], i_hi = t.m_ink[wa + 1u], q_lo = t.m_q[wa], q_hi = t.m_q[wa + 1u], e_lo = t.m_e[wa], e_hi = t.m_e[wa + 1u];
    uint32_t n_lo = t.m_num[wb], n_hi = t.m_num[wb + 1u], f_lo = t.m_flt[wb], f_hi = t.m_flt[wb + 1u], d_lo = t.m_dot[wb], d_hi = t.m_dot[wb + 1u];
    uint32_t cob = t.co[rs >> 12];
    asm volatile("" : "+v"(i_lo), "+v"(i_hi), "+v"(q_lo), "+v"(q_hi), "+v"(e_lo), "+v"(e_hi), "+v"(cob));
    asm volatile("" : "+v"(n_lo), "+v"(n_hi), "+v"(f_lo), "+v"(f_hi), "+v"(d_lo), "+v"(d_hi));
    // ---- string: the windows end at the next structural (bit 31 = the byte in front of it)
    const uint32_t vi = __funnelshift_r(i_lo, i_hi, rn & 31u), vq = __funnelshift_r(q_lo, q_hi, rn & 31u), ve = __funnelshift_r(e_lo, e_hi, rn & 31u);
    const uint32_t z = (uint32_t)__clz(vi);                                       // 32 for an empty window
    const uint32_t q = (uint32_t)max((int)rn - (int)z, (int)rs + 1);              // behind the last non-blank byte
    const uint32_t more_ink = (uint32_t)(vi == 0u) & (uint32_t)(rn - rs > 33u);
    const uint32_t top = z & 31u;                                                 // (an empty window: closed = 0 below)
    const uint32_t closed = (uint32_t)(rn - z > rs + 1u) & (uint32_t)(z < 32u) & ((vq << top) >> 31);
    const uint32_t close = q - 1u, b0 = rs + 1u;                                  // body = [b0, close)
    const uint32_t far = (rs >> 12) != (close >> 12);
    const uint32_t esc = ((ve << top) >> 31) | (far & cob);
    const uint32_t lng = close - b0 > kSpanCap;
    const uint32_t f_str = MSJ_SPAN_STRING | (closed ? (lng ? MSJ_SPAN_LONG : (esc ? MSJ_SPAN_ESCAPED : 0u)) : MSJ_SPAN_OPEN);
    const uint32_t e_str = closed ? lo + close : len;
    // ---- number: parse_number's scan (include/generic/number_parsing.mojo:41-59), see staged_token; the windows start
    //      at its first digit
    const uint32_t wn = ~__funnelshift_r(n_lo, n_hi, p0 & 31u), wf = __funnelshift_r(f_lo, f_hi, p0 & 31u), wd = __funnelshift_r(d_lo, d_hi, p0 & 31u);
    const uint32_t nd = min((uint32_t)(__ffs(wn) - 1), 32u);                      // digits
    const uint32_t sh = nd & 31u;
    const uint32_t flt = (wd >> sh) & 1u;                                         // the byte behind them is . e E
    const uint32_t wf2 = wf >> sh;                                                // structural or blank, from that byte on
    const uint32_t bad = ((flt | wf2) & 1u) ^ 1u;                                   // neither: the reference's NUMBER_ERROR
    const uint32_t fe = min((uint32_t)(__ffs(wf2) - 1), 32u);
    const uint32_t pe = p0 + nd + (flt ? fe : 0u);
    const uint32_t short_num = (uint32_t)(nd == 32u) | (flt & (uint32_t)(wf2 == 0u)) | (uint32_t)(rs + 80u > rlen);
    const uint32_t f_num = MSJ_SPAN_NUMBER | (flt ? MSJ_SPAN_FLOAT : 0u) | (bad ? MSJ_SPAN_BAD : 0u);
    e_out = is_str ? e_str : (is_num ? lo + pe : 0u);
    f_out = is_str ? f_str : (is_num ? f_num : 0u);
    return is_str ? (more_ink | (closed & lng)) : (is_num & short_num);  // (a long string: the slow path puts it on the long-string list)
}

// One token, any length, from the same maps (rare lanes; the chunks at the end of the input): staged_token with the
// string branch on m_q / m_e.  Returns true when the staged bytes were not enough (see staged_token).
__device__ __forceinline__ bool tile_token_slow(const TileMaps &t, uint32_t lo, uint32_t span, uint32_t len, uint32_t c, uint32_t rs,
                                             uint32_t rn, uint32_t &e_out, uint32_t &f_out) {
    const uint32_t rlen = min(len - lo, span);
    uint32_t e = 0, f = 0;
    if (c == '"') {
        f = MSJ_SPAN_STRING;
        uint32_t q = rs + 1u;
        if (rn > rs + 1u) {
            uint32_t tt = rn;
            for (;;) {  // one round unless more than 32 blanks stand in front of the next structural
                const uint32_t d = tt - (rs + 1u);
                uint32_t v = bits_before(t.m_ink, tt);
                if (d < 32u) v &= ~0u << (32u - d);
                if (v) {
                    q = tt - __clz(v);
                    break;
                }
                if (d <= 32u) break;
                tt -= 32u;
            }
        }
        const uint32_t close = q - 1u, b0 = rs + 1u;
        const bool closed = q > rs + 1u && (bits_at(t.m_q, close) & 1u);
        if (closed) {
            if (close - b0 > kSpanCap) {
                f |= MSJ_SPAN_LONG;
            } else if ((bits_at(t.m_e, close) & 1u) | (((rs >> 12) != (close >> 12)) ? t.co[rs >> 12] : 0u)) {
                f |= MSJ_SPAN_ESCAPED;
            }
            e = lo + close;
        } else {
            f |= MSJ_SPAN_OPEN;  // only the last token can be like this (stage 1 reports UNCLOSED_STRING)
            e = len;
        }
    } else if (c == '-' || c - '0' < 10u) {
        f = MSJ_SPAN_NUMBER;
        const uint32_t stop = (rs + 1u + kSpanCap < rlen) ? rs + 1u + kSpanCap : rlen;
        uint32_t p = rs + (c == '-' ? 1u : 0u);
        while (p < stop) {
            const uint32_t run = ~bits_at(t.m_num, p);
            if (run) {
                p += __ffs(run) - 1u;
                break;
            }
            p += 32u;
        }
        if (p > stop) p = stop;
        if (p < stop || lo + p == len) {
            if (p < rlen && (bits_at(t.m_dot, p) & 1u)) {
                f |= MSJ_SPAN_FLOAT;
                while (p < stop) {
                    const uint32_t hit = bits_at(t.m_flt, p);
                    if (hit) {
                        p += __ffs(hit) - 1u;
                        break;
                    }
                    p += 32u;
                }
                if (p > stop) p = stop;
            } else if (p < rlen && !(bits_at(t.m_flt, p) & 1u)) {
                f |= MSJ_SPAN_BAD;
            }
        }
        if (p == stop && lo + stop < len) {
            if (stop < rs + 1u + kSpanCap) return true;   // not the cap: the staged range ended
            f = (f & ~MSJ_SPAN_BAD) | MSJ_SPAN_LONG;      // kSpanCap characters and still no end
        } else {
            e = lo + p;
        }
    }
    e_out = e;
    f_out = f;
    return false;
}

// tbl[g] = number of structurals in front of byte g * kTgBytes (g = 0 .. ngroups; tbl[ngroups] = n): one search per
// group in the index array.  A plain binary search is 28 dependent loads (24 us for a 1 GiB buffer: nothing else runs
// meanwhile); the index of a JSON text is close to linear in the byte offset, so the search first closes in by
// interpolation -- a probe at the position the density so far predicts, pushed a little PAST the prediction so that
// the bracket shrinks from both sides -- and bisects what is left, a few dozen entries in one or two cache lines.
// The bracket invariants (everything in front of lo is < target, everything from hi on is >= target) hold after every
// probe whatever the prediction was worth, so the result is the exact lower bound for any sorted index.
// (Stage 1 could hand the same numbers over for nothing -- every tile's emission knows its first slot -- if the two
// calls shared more than the arrays.)
__global__ __launch_bounds__(256) void group_table(const uint32_t *__restrict__ idx, uint32_t n, uint32_t ngroups, uint64_t len,
                                                   uint32_t *__restrict__ tbl) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g > ngroups) return;
    uint32_t lo = 0, hi = n;
    if (g == ngroups) lo = n;
    const uint32_t target = g * kTgBytes;  // < len < 2^32
    const double density = (double)n / (double)len;
    int64_t pos = (int64_t)((double)target * density);
    for (int probe = 0; probe < 6 && hi - lo > 32u; probe++) {
        pos = pos < (int64_t)lo ? (int64_t)lo : (pos >= (int64_t)hi ? (int64_t)hi - 1 : pos);
        const uint32_t v = idx[pos];
        const int64_t est = (int64_t)(((double)target - (double)v) * density);  // entries between here and the target
        if (v < target) {
            lo = (uint32_t)pos + 1u;
            pos += est + (est >> 3) + 4;   // a little past it: the next probe should land behind the target
        } else {
            hi = (uint32_t)pos;
            pos += est + (est >> 3) - 4;   // (est <= 0) a little in front of it
        }
    }
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (idx[mid] < target) lo = mid + 1u; else hi = mid;
    }
    tbl[g] = lo;
}

// 16 bytes at `pos`, blanks past the end of the buffer (the last group only)
__device__ __noinline__ uint4 chunk16_or_blanks(const uint8_t *buf, uint64_t pos, uint64_t len) {
    uint32_t w[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t x = 0x20202020u;
#pragma unroll
        for (int bb = 0; bb < 4; bb++) {
            const uint64_t p = pos + 4u * d + bb;
            if (p < len) x = (x & ~(0xFFu << (8 * bb))) | ((uint32_t)buf[p] << (8 * bb));
        }
        w[d] = x;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// carries into the lanes of a wave from per-lane generate / propagate ballots and the carry into lane 0: bit l of the
// result = carry into lane l; *out = carry out of lane 63  (the 64-bit addition does the look-ahead)
__device__ __forceinline__ uint64_t lane_carries(uint64_t gen, uint64_t prop, uint32_t carry_in, uint32_t *out) {
    const uint64_t a = gen | prop, b = gen;
    const uint64_t s = a + b + carry_in;
    *out = (uint32_t)(((a & b) | ((a | b) & ~s)) >> 63);
    return s ^ a ^ b;
}


template <bool kFused, bool kSpans>
__global__ __launch_bounds__(kTgThreads) void token_tiles(const uint8_t *__restrict__ buf, uint64_t len, const uint32_t *__restrict__ idx,
                                                         uint64_t n, uint32_t *__restrict__ end, uint8_t *__restrict__ flags, uint32_t lds_limit,
                                                         uint8_t *__restrict__ type, int4 *__restrict__ chunk_agg, uint32_t *__restrict__ fix,
                                                         uint32_t fix_cap, const uint32_t *__restrict__ tbl) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[kTgStage + 16];
    __shared__ __attribute__((aligned(8))) uint32_t m_num[kTgMapWords], m_flt[kTgMapWords], m_dot[kTgMapWords], m_ink[kTgMapWords],
        m_q[kTgMapWords], m_e[kTgMapWords];
    __shared__ uint32_t co[kTgWaves + 1];
    const uint32_t nt = (uint32_t)n, len32 = (uint32_t)len;  // n < 2^31, len < 2^32 (the entry points check)
    // the wave index as a SCALAR: everything derived from it (the chunk, its addresses, the loop) is then scalar code
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t base = blockIdx.x * kTgBytes;  // < len
    MSJ_TSTAMP(0);
    // the chunks this group owns: those whose first token lies in [base, base + kTgBytes).  (Asking for the bytes before
    // the table entry has arrived -- the entry as a vector load, looked at behind the byte loads -- was measured: the
    // workgroup's first 2.5 us shrink, the kernel as a whole got 13 % slower, scripts/tile_stamps.py and
    // profiles/r03/prep_tiles_history.txt.)
    const uint32_t t_lo = tbl[blockIdx.x], t_hi = tbl[blockIdx.x + 1u];
    const uint32_t c_lo = (t_lo + kChunk - 1u) / kChunk, c_hi = (t_hi + kChunk - 1u) / kChunk;
    if (c_lo >= c_hi) return;  // uniform: none (sparse input: nothing to stage for)

    // ---- the wave's part of the range: 1 KiB contiguous per wave instruction, linear in LDS
    const uint32_t region = wave * 4096u + 16u * lane;
    const bool tail = (uint64_t)base + kTgStage > len;  // uniform: the range reaches past the buffer (blanks there)
    constexpr uint32_t kHaloInsts = kTgHaloBlocks * 64u / 1024u;
    uint4 v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (wave == kTgTiles && k >= kHaloInsts) break;  // uniform
        const uint64_t pos = (uint64_t)base + region + 1024u * k;
        v[k] = tail ? chunk16_or_blanks(buf, pos, len) : *reinterpret_cast<const uint4 *>(buf + pos);
    }
    // the byte in front of the wave's part, 64 of them: what its first byte's escape state depends on
    uint32_t wb = 0;
    if (kSpans && wave > 0) {
        const uint64_t pos = (uint64_t)base + wave * 4096u - 64u + lane;
        wb = buf[pos < len ? pos : len - 1u];  // past the buffer: whatever, no tile behind it holds anything but blanks
    }
    // this wave's first chunk: its indices are on their way while the bytes are classified
    const uint32_t last = nt - 1u;
    const auto request = [&](uint32_t c, uint32_t &i0, uint32_t &i1, uint32_t &nxt) {
        const uint32_t tok0 = c * kChunk + 2u * lane;
        if ((c + 1u) * kChunk < nt) {  // uniform: all 128 tokens and one behind them exist
            // scalar base, the lane's pair at a constant offset; three words per lane: the pair and the structural behind
            // it (the last lane's is the next chunk's first).  One vector load -- a scalar load of idx[128 (c + 1)] would
            // sit in the same counter as the LDS reads of the evaluation, and every wait for those would wait for it
            typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
            typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
            const uint32_t *cp = idx + (uint64_t)c * kChunk;
            const u32x3 p = *reinterpret_cast<const u32x3_a4 *>(cp + 2u * lane);
            i0 = p.x, i1 = p.y, nxt = p.z;
        } else {
            i0 = idx[min(tok0, last)], i1 = idx[min(tok0 + 1u, last)];
            nxt = len32;
        }
    };
    uint32_t c = c_lo + wave;
    uint32_t i0 = 0, i1 = 0, nxt = 0;
    if (c < c_hi) request(c, i0, i1, nxt);
    MSJ_TSTAMP(1);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        if (wave == kTgTiles && k >= kHaloInsts) break;
        *reinterpret_cast<uint4 *>(stage + region + 1024u * k) = v[k];
    }
    const uint32_t j = threadIdx.x;  // lane j of the workgroup classifies block j of the range
    MSJ_TSTAMP(2);
#ifdef MSJ_TILE_PRIO_CLASS
    __builtin_amdgcn_s_setprio(MSJ_TILE_PRIO_CLASS);
#endif
    if (kSpans) {
        // LDS written by this wave, read by this wave
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // the carry into the wave's first byte: the parity of the run of backslashes that ends in front of it.  A run
        // that fills the 64-byte window goes on in the window in front -- down to the start of the range at most, and the
        // carry into the range itself is taken as 0: a run that comes in from the bytes in front of the range belongs to a
        // token of an earlier group -- inside a string to that string, outside one (a byte soup) to the scalar its first
        // backslash starts (stage 1 never makes the quote behind such a run a structural) -- and the quotes this
        // group's tokens look at lie behind an opening quote inside the range (test_prep_around_the_tile_groups, g)
        uint32_t e_in = 0;
        if (wave > 0) {
            uint64_t WB = __ballot(wb == 0x5Cu);
            uint32_t back = 64u, run = 0;
            while (WB == ~0ull && back < wave * 4096u) {  // uniform, practically never
                run += 64u;
                back += 64u;
                const uint64_t pos = (uint64_t)base + wave * 4096u - back + lane;
                WB = __ballot(buf[pos < len ? pos : len - 1u] == 0x5Cu);
            }
            e_in = (run + msj::top_run(WB)) & 1u;
        }
        uint64_t q_mask = 0, bs = 0;
        msj::SpanClasses cl = {0, 0, 0, ~0ull};  // a lane without a block: nothing set in any map
        msj::TileClasses tc = {0, 0};
        const bool own = j < kTgBlocks;
        if (own) {
            uint32_t x[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 w = *reinterpret_cast<const uint4 *>(stage + 64u * j + 16 * q);
                x[4 * q] = w.x, x[4 * q + 1] = w.y, x[4 * q + 2] = w.z, x[4 * q + 3] = w.w;
            }
            uint64_t pl[8];
            msj::bitplanes(x, pl);
            cl = msj::span_classes(pl);
            tc = msj::tile_classes(pl);
            bs = cl.backslash;
        }
        {   // unescaped quotes: a block's last run of backslashes decides the carry into the next block (odd: escaped),
            // a block of 64 backslashes hands its own carry on
            const uint32_t r = msj::top_run(bs);
            uint32_t unused;
            const uint64_t carries = lane_carries(__ballot(r < 64u && (r & 1u)), __ballot(r == 64u), e_in, &unused);
            uint32_t lane_out;
            const uint64_t escaped = msj::escaped_mask(bs, (uint32_t)(carries >> lane) & 1u, &lane_out);
            q_mask = tc.quote & ~escaped;
        }
        uint64_t e_mask;
        {   // quotes with a backslash since the quote in front of them: (backslash + ~quote) & quote, carries across lanes
            const uint64_t m = ~q_mask;
            const uint64_t s0 = bs + m;
            uint32_t tile_out;
            const uint64_t carries = lane_carries(__ballot(s0 < bs), __ballot(q_mask == 0ull), 0u, &tile_out);
            e_mask = (s0 + ((carries >> lane) & 1ull)) & q_mask;
            if (lane == 0) co[wave] = tile_out;
        }
        if (j < kTgBlocks + 2u) {  // the lanes behind the last block write the zero words behind the maps
            const uint32_t w = kSpanMapFront + 2u * j;
            const uint64_t ink = ~cl.blank;
            *reinterpret_cast<uint2 *>(m_num + w) = make_uint2((uint32_t)cl.digit, (uint32_t)(cl.digit >> 32));
            *reinterpret_cast<uint2 *>(m_flt + w) = make_uint2((uint32_t)cl.sow, (uint32_t)(cl.sow >> 32));
            *reinterpret_cast<uint2 *>(m_dot + w) = make_uint2((uint32_t)tc.dote, (uint32_t)(tc.dote >> 32));
            *reinterpret_cast<uint2 *>(m_ink + w) = make_uint2((uint32_t)ink, (uint32_t)(ink >> 32));
            *reinterpret_cast<uint2 *>(m_q + w) = make_uint2((uint32_t)q_mask, (uint32_t)(q_mask >> 32));
            *reinterpret_cast<uint2 *>(m_e + w) = make_uint2((uint32_t)e_mask, (uint32_t)(e_mask >> 32));
        }
        if (j == kTgThreads - 1u) {  // ... and the ones in front of them
            *reinterpret_cast<uint2 *>(m_num) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_flt) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_dot) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_ink) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_q) = make_uint2(0, 0);
            *reinterpret_cast<uint2 *>(m_e) = make_uint2(0, 0);
            co[kTgWaves] = 0;
        }
    }
    MSJ_TSTAMP(3);
    __syncthreads();
    MSJ_TSTAMP(4);
#ifdef MSJ_TILE_PRIO_LOOP
    __builtin_amdgcn_s_setprio(MSJ_TILE_PRIO_LOOP);
#endif

    const TileMaps maps = {stage, m_num, m_flt, m_dot, m_ink, m_q, m_e, co};
    const bool wide = (reinterpret_cast<uintptr_t>(end) & 7u) == 0 &&
                      ((reinterpret_cast<uintptr_t>(flags) | reinterpret_cast<uintptr_t>(type)) & 1u) == 0;  // uniform
    for (; c < c_hi; c += kTgWaves) {
        // the next chunk's indices: requested now, looked at in the next iteration
        uint32_t n_i0 = 0, n_i1 = 0, n_nxt = 0;
        if (c + kTgWaves < c_hi) request(c + kTgWaves, n_i0, n_i1, n_nxt);
        const uint32_t tok0 = c * kChunk + 2u * lane;
        const bool allhere = (c + 1u) * kChunk < nt;  // uniform: 128 tokens and one behind them
        const bool have0 = allhere || tok0 < nt, have1 = allhere || tok0 + 1u < nt;
        // the structural behind this lane's pair: loaded with it where all tokens exist, else the next lane's first token
        const uint32_t i2 = allhere ? nxt : (uint32_t)__builtin_amdgcn_update_dpp((int)nxt, (int)i0, 0x130, 0xF, 0xF, false);  // wave_shl:1
        const uint32_t start0 = have0 ? i0 : 0u, start1 = have1 ? i1 : 0u;
        const uint32_t next1 = (allhere || tok0 + 2u < nt) ? i2 : len32;
        const uint32_t next0 = have1 ? start1 : len32;
        // uniform: the chunk's bytes -- through the byte at the next chunk's first structural -- are staged
        const uint32_t first_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)i0);
        const uint32_t hi = allhere ? (uint32_t)__builtin_amdgcn_readlane((int)nxt, 63) + 1u : len32;
        const bool staged = hi <= base + kTgStage && hi - first_pos <= lds_limit && first_pos >= base;
        uint32_t e0 = 0, f0 = 0, c0 = 0, e1 = 0, f1 = 0, c1 = 0;
        if (kSpans && staged && allhere) {
            // the common case, straight-line: all 128 tokens exist, their bytes are staged; one evaluation per pair
            // (in a valid document at most one of two neighbouring tokens is a string or a number)
            const uint32_t rs0 = i0 - base, rs1 = i1 - base, rn1 = i2 - base;
            c0 = stage[rs0];
            c1 = stage[rs1];
            const uint32_t s0 = starts_scalar(c0), s1 = starts_scalar(c1);
            const uint32_t cs = s0 ? c0 : c1, rs = s0 ? rs0 : rs1, rn = s0 ? rs1 : rn1;
            uint32_t e, f;
            const uint32_t again = pair_fast(maps, base, kTgStage, len32, cs, s0 | s1, rs, rn, e, f);
            if (__ballot((again | (s0 & s1)) != 0u) != 0ull) {  // rare: a window came up short, a long string, or two scalars in a row
                if (again) {
                    if (tile_token_slow(maps, base, kTgStage, len32, cs, rs, rn, e, f))
                        fix_later(fix, fix_cap, tok0 + (s0 ? 0u : 1u), e, f);
                    else if (f == kLongString)
                        long_later(fix, fix_cap, tok0 + (s0 ? 0u : 1u));
                }
                if (s0 & s1) {
                    if (tile_token_slow(maps, base, kTgStage, len32, c1, rs1, rn1, e1, f1))
                        fix_later(fix, fix_cap, tok0 + 1u, e1, f1);
                    else if (f1 == kLongString)
                        long_later(fix, fix_cap, tok0 + 1u);
                }
            }
            e0 = s0 ? e : 0u, f0 = s0 ? f : 0u;
            if (!(s0 & s1)) e1 = s0 ? 0u : e, f1 = s0 ? 0u : f;
        } else if (!staged) {
            if (have0) {
                if (kSpans) {
                    span_of(FromGlobal{buf, len}, (uint64_t)start0, (uint64_t)next0, len, e0, f0);
                    if (f0 == kLongString) long_later(fix, fix_cap, tok0);
                }
                if (kFused) c0 = buf[start0];
            }
            if (have1) {
                if (kSpans) {
                    span_of(FromGlobal{buf, len}, (uint64_t)start1, (uint64_t)next1, len, e1, f1);
                    if (f1 == kLongString) long_later(fix, fix_cap, tok0 + 1u);
                }
                if (kFused) c1 = buf[start1];
            }
        } else if (have0) {  // staged: the type bytes only (kSpans = false), or the chunk at the end of the index
            c0 = stage[start0 - base];
            c1 = have1 ? (uint32_t)stage[start1 - base] : 0u;
            if (kSpans) {
                if (tile_token_slow(maps, base, kTgStage, len32, c0, start0 - base, next0 - base, e0, f0))
                    fix_later(fix, fix_cap, tok0, e0, f0);
                else if (f0 == kLongString)
                    long_later(fix, fix_cap, tok0);
                if (have1) {
                    if (tile_token_slow(maps, base, kTgStage, len32, c1, start1 - base, next1 - base, e1, f1))
                        fix_later(fix, fix_cap, tok0 + 1u, e1, f1);
                    else if (f1 == kLongString)
                        long_later(fix, fix_cap, tok0 + 1u);
                }
            }
        }
        if (allhere && wide) {
            const uint64_t cbase = (uint64_t)c * kChunk;  // scalar; the lane's pair at a constant offset
            if (kSpans) {  // write-once streams: non-temporal stores
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 ee = {e0, e1};
                __builtin_nontemporal_store(ee, reinterpret_cast<u32x2 *>(end + cbase) + lane);
                __builtin_nontemporal_store((uint16_t)(f0 | (f1 << 8)), reinterpret_cast<uint16_t *>(flags + cbase) + lane);
            }
            if (kFused) reinterpret_cast<uint16_t *>(type + cbase)[lane] = (uint16_t)(c0 | (c1 << 8));
        } else if (have0) {
            if (kSpans) {
                end[tok0] = e0;
                flags[tok0] = (uint8_t)f0;
            }
            if (kFused) type[tok0] = (uint8_t)c0;
            if (have1) {
                if (kSpans) {
                    end[tok0 + 1] = e1;
                    flags[tok0 + 1] = (uint8_t)f1;
                }
                if (kFused) type[tok0 + 1] = (uint8_t)c1;
            }
        }
        if (kFused) {
            // the chunk's bracket counts: '[' '{' are 5B 7B, ']' '}' are 5D 7D -- one masked compare each, the compare IS the
            // ballot, the counts are scalar.  (The minimum and maximum of the running depth, which token_spans reduces
            // per workgroup with two DPP chains, are folded in by apply_depth, which has every running value in a
            // register anyway and time to spare: kNone / -kNone = "none from here".)
            const uint32_t k0 = have0 ? (c0 & 0xDFu) : 0u, k1 = have1 ? (c1 & 0xDFu) : 0u;
            const uint64_t up0 = __ballot(k0 == 0x5Bu), dn0 = __ballot(k0 == 0x5Du), up1 = __ballot(k1 == 0x5Bu), dn1 = __ballot(k1 == 0x5Du);
            if (lane == 0) {
                const int ups = (int)__popcll(up0) + (int)__popcll(up1), downs = (int)__popcll(dn0) + (int)__popcll(dn1);
                chunk_agg[c] = make_int4(ups - downs, kNone, -kNone, ups);
            }
        }
        i0 = n_i0, i1 = n_i1, nxt = n_nxt;
    }
    MSJ_TSTAMP(5);
}

// the tokens on the fix-up list, from global memory, then the strings on the long-string list (a wave each).  Neither
// list is cleared here: long_strings_big, always launched behind this kernel, does that (stream order: every workgroup of
// this kernel has read the counts by then -- the "last workgroup clears" protocol of round 2 cost one returning atomic
// per workgroup on one word, ~85 per microsecond chip-wide).
__global__ __launch_bounds__(256) void span_fixup(const uint8_t *__restrict__ buf, uint64_t len, const uint32_t *__restrict__ idx, uint64_t n,
                                                  uint32_t *__restrict__ end, uint8_t *__restrict__ flags, uint32_t *__restrict__ fix, uint32_t cap) {
    const uint32_t count = __hip_atomic_load(&fix[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (count != 0) {
        const uint32_t stride = gridDim.x * blockDim.x, me = blockIdx.x * blockDim.x + threadIdx.x;
        const auto redo = [&](uint64_t tok) {
            uint32_t e, f;
            span_of(FromGlobal{buf, len}, (uint64_t)idx[tok], tok + 1 < n ? (uint64_t)idx[tok + 1] : len, len, e, f);
            end[tok] = e;
            flags[tok] = (uint8_t)f;
        };
        if (count <= cap) {
            for (uint32_t k = me; k < count; k += stride) redo(fix[2 + k]);
        } else {  // the list overflowed: every token that carries the sentinel
            for (uint64_t tok = me; tok < n; tok += stride)
                if (flags[tok] == kFixSentinel) redo(tok);
        }
    }
    // ---- strings over kSpanCap bytes: end[] holds their closing quote, the body is scanned here (one wave per string)
    uint32_t *lng = fix + kFixWords;
    const uint32_t lcount = __hip_atomic_load(&lng[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lcount != 0) {
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
        const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
        const auto one = [&](uint64_t tok) {  // uniform per wave
            const uint64_t b0 = (uint64_t)idx[tok] + 1u, close = end[tok];
            if (close - b0 > kBigBody) {
                if (lane == 0) {
                    const uint32_t slot = atomicAdd(&lng[2], 1u);
                    if (slot < kBigCap) lng[kLngHdr + kFixCap + slot] = (uint32_t)tok;
                }
            } else {
                const bool any = wave_has_backslash(buf, b0, close, lane);
                if (lane == 0) flags[tok] = (uint8_t)(MSJ_SPAN_STRING | (any ? MSJ_SPAN_ESCAPED : 0u));
            }
        };
        if (lcount <= cap) {
            for (uint32_t k = wv; k < lcount; k += nw) one(lng[kLngHdr + k]);
        } else {  // the list overflowed: every token that carries the marker
            for (uint64_t t0 = (uint64_t)wv * 64u; t0 < n; t0 += (uint64_t)nw * 64u) {  // uniform
                const uint64_t t = t0 + lane;
                uint64_t m = __ballot(t < n && flags[t] == kLongString);
                while (m) {  // uniform
                    const uint32_t b = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1ull;
                    one(t0 + b);
                }
            }
        }
    }
}

// Strings of more than kBigBody bytes (an embedded blob): every workgroup takes pieces of every such string, a wave per
// piece; the last workgroup to finish writes the flags.  Workgroup 0 also clears the two lists span_fixup worked through
// (this kernel runs behind it on the stream).  Launched with every span call: ~2 us when there is nothing to do.
constexpr uint32_t kBigGroups = 64;
__global__ __launch_bounds__(256) void long_strings_big(const uint8_t *__restrict__ buf, const uint32_t *__restrict__ idx,
                                                        const uint32_t *__restrict__ end, uint8_t *__restrict__ flags, uint32_t *__restrict__ fix) {
    uint32_t *lng = fix + kFixWords, *big = lng + kLngHdr + kFixCap, *res = big + kBigCap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        __hip_atomic_store(&fix[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&lng[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t nbig = __hip_atomic_load(&lng[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nbig == 0u) return;  // (nobody has touched lng[3])
    if (nbig > kBigCap) nbig = kBigCap;  // cannot happen within one segment
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    const uint32_t nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t k = 0; k < nbig; k++) {  // uniform
        const uint64_t tok = big[k];
        const uint64_t b0 = (uint64_t)idx[tok] + 1u, close = end[tok];
        const uint64_t pieces = (close - b0 + kBigPiece - 1u) / kBigPiece;
        bool any = false;
        for (uint64_t p = wv; p < pieces && !any; p += nw) {  // uniform
            const uint64_t lo = b0 + p * kBigPiece, hi = lo + kBigPiece < close ? lo + kBigPiece : close;
            any = wave_has_backslash(buf, lo, hi, lane);
        }
        if (any && lane == 0) atomicOr(&res[k], 1u);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&lng[3], 1u) == gridDim.x - 1u) {  // every workgroup is through
        __threadfence();
        for (uint32_t k = 0; k < nbig; k++) {
            const uint32_t any = __hip_atomic_load(&res[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flags[big[k]] = (uint8_t)(MSJ_SPAN_STRING | (any ? MSJ_SPAN_ESCAPED : 0u));
            res[k] = 0u;
        }
        lng[3] = 0u;
        __hip_atomic_store(&lng[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// block aggregates of the token pre-pass (kBlock tokens) from the fused kernels' chunk aggregates (kChunk tokens each),
// which carry bracket counts only (kNone / -kNone as minimum / maximum: "none from here", apply_depth supplies them):
// sixteen chunks make a block, sixteen lanes a DPP row -- one coalesced load per lane, the row's sums by four row_shr
// additions, the row's last lane writes the block
static_assert(kBlock / kChunk == 16, "one DPP row per block");
__global__ __launch_bounds__(256) void merge_chunk_counts(const int4 *__restrict__ sub, uint32_t nsub, int32_t *__restrict__ block_agg,
                                                          uint32_t nblocks) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const int4 q = i < nsub ? sub[i] : make_int4(0, 0, 0, 0);
    uint32_t sum = (uint32_t)q.x, no = (uint32_t)q.w;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(sum), "+v"(no));
    const uint32_t b = i >> 4;
    if ((threadIdx.x & 15u) == 15u && b < nblocks) *reinterpret_cast<int4 *>(block_agg + 4 * (uint64_t)b) = make_int4((int)sum, kNone, -kNone, (int)no);
}
// Which of the two kernels a token call runs.  0 (default): by the density of the index -- token_tiles pays for every
// BYTE it stages (classification, escape and carry chains: ~1.6 vector instructions per byte) and little per token,
// token_spans the other way round; measured on 1 GiB, same box, alternating (profiles/r03/stage2_prep_r03_tiles_ab.txt,
// prep_density_ab.txt), tiles against tokens: minified (one structural per 5.2 bytes) 0.87 against 1.09 ms, tab + CRLF
// (1 / 7.2) 0.72 / 0.79, indent 2 (1 / 8.0) 0.64 / 0.73, UTF-8-heavy (1 / 9.6) 0.63 / 0.64, indent 4 (1 / 10.3)
// 0.575 / 0.592, indent 8 (1 / 14.9) 0.51 / 0.43 -- the tiles from one structural per 11 bytes on.  1: token_spans,
// 2: token_tiles whatever the density (msj_debug_set_span_mode: the tests run both, A/B runs).
static bool by_tiles(const msj_token_opts &o, uint64_t n, uint64_t len) { return o.span_mode == 2u || (o.span_mode == 0u && n * 11u >= len); }
}  // namespace msj_tokens

// what the tests place their tokens around: bytes of the buffer per workgroup of token_tiles (0) and its halo (1)
extern "C" uint32_t msj_debug_tile_group(int32_t which) { return which == 0 ? msj_tokens::kTgBytes : msj_tokens::kTgHaloBlocks * 64u; }
#ifdef MSJ_TILE_STAMPS
extern "C" int msj_debug_set_tile_stamps(void *d_stamps) {
    unsigned long long *p = static_cast<unsigned long long *>(d_stamps);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(msj_tokens::g_tile_stamps), &p, sizeof(p));
}
#endif

// the work list of span_fixup: zeroed once by the owner (the kernels leave it zeroed)
extern "C" uint64_t msj_span_fix_bytes(void) { return (msj_tokens::kFixWords + msj_tokens::kLngWords) * sizeof(uint32_t); }  // + the long-string list

// ---- workspace of the span / prep calls: the token pre-pass's own words, then (16-byte aligned) one int4 per chunk
// of kChunk tokens (the fused kernels' depth aggregates), then the group table of token_tiles
static uint64_t chunk_count(uint64_t n) { return n ? (n + msj_tokens::kChunk - 1) / msj_tokens::kChunk : 1; }
static uint64_t group_count(uint64_t len) { return (len + msj_tokens::kTgBytes - 1) / msj_tokens::kTgBytes; }
static uint64_t sub_offset_bytes(uint64_t n, int with_match) { return (msj_tokens_workspace_bytes(n, with_match) + 15u) & ~15ull; }
extern "C" uint64_t msj_stage2_prep_workspace_bytes(uint64_t n, uint64_t len, int with_match) {
    return sub_offset_bytes(n, with_match) + chunk_count(n) * sizeof(int4) + (group_count(len) + 2) * sizeof(uint32_t);
}
static int4 *sub_of(int32_t *d_ws, uint64_t n, int with_match) {
    return reinterpret_cast<int4 *>(reinterpret_cast<uint8_t *>(d_ws) + sub_offset_bytes(n, with_match));
}
static uint32_t *table_of(int32_t *d_ws, uint64_t n, int with_match) {
    return reinterpret_cast<uint32_t *>(sub_of(d_ws, n, with_match) + chunk_count(n));
}

// the tile-organised kernel with its table in front of it and the fix-up pass behind it
template <bool kFused, bool kSpans>
static void launch_token_tiles(const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint32_t *d_end, uint8_t *d_flags,
                               uint8_t *d_type, int4 *sub, uint32_t *tbl, uint32_t *d_fix, hipStream_t s, const msj_token_opts &o) {
    using namespace msj_tokens;
    const uint32_t ngroups = (uint32_t)group_count(len);
    hipLaunchKernelGGL(group_table, dim3((ngroups + 1u + 255u) / 256u), dim3(256), 0, s, d_idx, (uint32_t)n, ngroups, len, tbl);
    hipLaunchKernelGGL((token_tiles<kFused, kSpans>), dim3(ngroups), dim3(kTgThreads), 0, s, d_buf, len, d_idx, n, d_end, d_flags,
                       o.lds_limit < kTgStage ? o.lds_limit : 0xFFFFFFFFu, d_type, sub, d_fix, fix_cap(o), tbl);
    if (kSpans) {
        hipLaunchKernelGGL(span_fixup, dim3(kFixGroups), dim3(256), 0, s, d_buf, len, d_idx, n, d_end, d_flags, d_fix, fix_cap(o));
        hipLaunchKernelGGL(long_strings_big, dim3(kBigGroups), dim3(256), 0, s, d_buf, d_idx, d_end, d_flags, d_fix);
    }
}

int msj_launch_token_spans(const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint32_t *d_end, uint8_t *d_flags,
                           int32_t *d_ws, uint32_t *d_fix, void *stream, const msj_token_opts &o) {
    using namespace msj_tokens;
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the spans alone: the tiles only where the index is dense (1 GiB minified 0.70 against 0.76 ms, UTF-8-heavy 0.53
    // against 0.49, pretty-printed 0.48 against 0.45: without the type bytes and bracket counts the fused call gets for
    // nothing, the tile kernel's per-byte work pays off later) -- from one structural per 7 bytes on
    if ((o.span_mode == 2u || (o.span_mode == 0u && n * 7u >= len)) && (reinterpret_cast<uintptr_t>(d_idx) & 7u) == 0) {
        launch_token_tiles<false, true>(d_buf, len, d_idx, n, d_end, d_flags, nullptr, nullptr, table_of(d_ws, n, 0), d_fix, s, o);
        return (int)hipGetLastError();
    }
    const uint32_t lds_limit = span_lds_limit(o);  // stretches over this many bytes take the global-memory path
    hipLaunchKernelGGL(token_spans<false>, dim3((uint32_t)((n + kSpanTokens - 1) / kSpanTokens)), dim3(kSpanThreads), 0, s, d_buf, len, d_idx, n,
                       d_end, d_flags, lds_limit, static_cast<uint8_t *>(nullptr), static_cast<int4 *>(nullptr), d_fix, fix_cap(o));
    hipLaunchKernelGGL(span_fixup, dim3(kFixGroups), dim3(256), 0, s, d_buf, len, d_idx, n, d_end, d_flags, d_fix, fix_cap(o));
    hipLaunchKernelGGL(long_strings_big, dim3(kBigGroups), dim3(256), 0, s, d_buf, d_idx, d_end, d_flags, d_fix);
    return (int)hipGetLastError();
}

// ---- everything stage 2 reads first, in one go (rows f1 + f2 + f4): the span kernel has every token's
// first byte in LDS anyway, so it writes the type bytes and the depth aggregates as well and the token
// pre-pass starts at its scan -- one pass over the buffer instead of two.
int msj_launch_stage2_prep(const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type, int32_t *d_depth,
                           uint32_t *d_match, uint32_t *d_end, uint8_t *d_flags, msj_tokens_result *d_result, int32_t *d_ws,
                           uint32_t *d_fix, void *stream, const msj_token_opts &o) {
    using namespace msj_tokens;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t nb64 = (n + kBlock - 1) / kBlock;
    if (nb64 > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const uint32_t nb = (uint32_t)nb64;
    // the fused kernel's aggregates live behind the token pre-pass's own workspace (16-byte aligned)
    const int wm = o.d_pairs ? 2 : (d_match != nullptr ? 1 : 0);
    int4 *sub = sub_of(d_ws, n, wm);
    if (n && by_tiles(o, n, len)) {
        launch_token_tiles<true, true>(d_buf, len, d_idx, n, d_end, d_flags, d_type, sub, table_of(d_ws, n, wm), d_fix, s, o);
        hipLaunchKernelGGL(merge_chunk_counts, dim3((nb * 16u + 255u) / 256u), dim3(256), 0, s, sub, (uint32_t)chunk_count(n), d_ws, nb);
    } else if (n) {
        const uint32_t nsub = (uint32_t)((n + kSpanTokens - 1) / kSpanTokens);
        const uint32_t lds_limit = span_lds_limit(o);
        hipLaunchKernelGGL(token_spans<true>, dim3(nsub), dim3(kSpanThreads), 0, s, d_buf, len, d_idx, n, d_end, d_flags, lds_limit, d_type, sub, d_fix, fix_cap(o));
        hipLaunchKernelGGL(span_fixup, dim3(kFixGroups), dim3(256), 0, s, d_buf, len, d_idx, n, d_end, d_flags, d_fix, fix_cap(o));
        hipLaunchKernelGGL(long_strings_big, dim3(kBigGroups), dim3(256), 0, s, d_buf, d_idx, d_end, d_flags, d_fix);
        hipLaunchKernelGGL(merge_chunk_counts, dim3((nb * 16u + 255u) / 256u), dim3(256), 0, s, sub, (uint32_t)chunk_count(n), d_ws, nb);
    }
    return launch_depth_passes(d_idx, n, d_type, d_depth, d_match, d_result, d_ws, s, o);
}

// ---- PROTOTYPE (round 5): depth (and partners) from type bytes that stage 1 wrote itself
// (msj_stage1_types_device) -- the bracket counts per block from one pass over type[] (1 byte per token), then the scans
// and apply_depth as above.  One thread per 16 tokens, one wave per 1 024, two waves per block of 2 048.
namespace msj_tokens {
__global__ __launch_bounds__(256) void count_brackets(const uint8_t *__restrict__ type, uint32_t n, int32_t *__restrict__ block_agg) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, first = t * 16u;
    uint32_t ups = 0, downs = 0;
    if (first + 16u <= n) {
        const uint4 q = *reinterpret_cast<const uint4 *>(type + first);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t x = w[k] & 0xDFDFDFDFu;  // '[' '{' -> 5B, ']' '}' -> 5D
            ups += (uint32_t)__builtin_popcount(zero_bytes32(x ^ 0x5B5B5B5Bu));
            downs += (uint32_t)__builtin_popcount(zero_bytes32(x ^ 0x5D5D5D5Du));
        }
    } else {
        for (uint32_t i = first; i < n; i++) {
            const uint32_t c = type[i] & 0xDFu;
            ups += c == 0x5Bu;
            downs += c == 0x5Du;
        }
    }
    // 128 threads = one block of 2 048 tokens: two waves, summed through LDS
    ups = wave_incl_sum(ups);
    downs = wave_incl_sum(downs);
    __shared__ uint32_t s_up[4], s_dn[4];
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 63u) {
        s_up[wave] = ups;
        s_dn[wave] = downs;
    }
    __syncthreads();
    if ((threadIdx.x & 127u) == 0u) {
        const uint32_t b = t >> 7;
        if ((uint64_t)b * kBlock < n) {
            const uint32_t u = s_up[wave] + s_up[wave + 1], d = s_dn[wave] + s_dn[wave + 1];
            *reinterpret_cast<int4 *>(block_agg + 4 * (uint64_t)b) = make_int4((int)u - (int)d, kNone, -kNone, (int)u);
        }
    }
}
}  // namespace msj_tokens

extern "C" int msj_launch_depth_from_types(const uint8_t *d_type, uint64_t n, int32_t *d_depth, uint32_t *d_match, msj_tokens_result *d_result,
                                           int32_t *d_ws, void *stream, const msj_token_opts &o) {
    using namespace msj_tokens;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t nb64 = (n + kBlock - 1) / kBlock;
    if (nb64 > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    if (n) hipLaunchKernelGGL(count_brackets, dim3((uint32_t)((nb64 * 128u + 255u) / 256u)), dim3(256), 0, s, d_type, (uint32_t)n, d_ws);
    return launch_depth_passes(nullptr, n, const_cast<uint8_t *>(d_type), d_depth, d_match, d_result, d_ws, s, o);
}

int msj_launch_tokens(const uint8_t *d_buf, uint64_t len, const uint32_t *d_idx, uint64_t n, uint8_t *d_type, int32_t *d_depth,
                      uint32_t *d_match, msj_tokens_result *d_result, int32_t *d_ws, void *stream, const msj_token_opts &o) {
    using namespace msj_tokens;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t nb64 = (n + kBlock - 1) / kBlock;
    if (nb64 > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
    const uint32_t nb = (uint32_t)nb64;
    const int wm = o.d_pairs ? 2 : (d_match != nullptr ? 1 : 0);
    int4 *sub = sub_of(d_ws, n, wm);
    if (n && by_tiles(o, n, len)) {
        launch_token_tiles<true, false>(d_buf, len, d_idx, n, nullptr, nullptr, d_type, sub, table_of(d_ws, n, wm), nullptr, s, o);
        hipLaunchKernelGGL(merge_chunk_counts, dim3((nb * 16u + 255u) / 256u), dim3(256), 0, s, sub, (uint32_t)chunk_count(n), d_ws, nb);
    } else if (n) {
        const uint32_t nsub = (uint32_t)((n + kSpanTokens - 1) / kSpanTokens);
        hipLaunchKernelGGL((token_spans<true, false>), dim3(nsub), dim3(kSpanThreads), 0, s, d_buf, len, d_idx, n, static_cast<uint32_t *>(nullptr),
                           static_cast<uint8_t *>(nullptr), kSpanLds, d_type, sub, static_cast<uint32_t *>(nullptr), 0u);
        hipLaunchKernelGGL(merge_chunk_counts, dim3((nb * 16u + 255u) / 256u), dim3(256), 0, s, sub, (uint32_t)chunk_count(n), d_ws, nb);
    }
    return launch_depth_passes(d_idx, n, d_type, d_depth, d_match, d_result, d_ws, s, o);
}

int msj_launch_stitch_partners(const msj_stitch_args &a, uint32_t *d_match, msj_tokens_result *d_results, const msj_tokens_result *d_prev,
                               void *stream) {
    if (a.n_segments < 2u) return 0;
    hipLaunchKernelGGL(msj_tokens::stitch_partners, dim3(a.n_segments - 1u), dim3(256), 0, static_cast<hipStream_t>(stream), a, d_match, d_results,
                       d_prev);
    return (int)hipGetLastError();
}
