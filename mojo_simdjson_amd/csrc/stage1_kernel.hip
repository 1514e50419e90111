// stage1_kernel.hip -- single-pass stage-1 structural indexer for gfx950 (MI355X).
//
// Replaces the reference's serial block loop
//   JsonStructuralIndexer.index[128] / step / next / finish
//   (src/mojo_simdjson/generic/stage1/json_structural_indexer.mojo:81-186)
// by one kernel launch over the whole buffer:
//
//   * every wave64 is an independent, persistent worker: it draws 4 KiB tiles from
//     an ordered ticket counter; each lane owns one 64-byte block = the unit of one
//     JsonScanner.next call, and all masks are uint64 with the reference's bit
//     order (lane_math.h).  The worker path contains no workgroup barrier;
//   * the three 1-bit carries the reference threads through its loop
//     (next_is_escaped json_escape_scanner.mojo:13, prev_in_string
//     json_string_scanner.mojo:49, prev_scalar json_scanner.mojo:57) are
//     resolved lane -> wave with __ballot + a 64-bit carry-lookahead add
//     (escape), ballot/mbcnt prefix parity (in-string) and a one-lane shuffle
//     (prev_scalar); the escape / prev_scalar / UTF-8 carries INTO a tile are
//     derived locally from the 64 bytes in front of it;
//   * across tiles only the in-string bit and the running structural count are
//     chained.  Each tile publishes a 64-bit aggregate (parity, count and error
//     bit for both possible incoming in-string states); one workgroup does not
//     index anything: it is the RESOLVER, whose four waves fold those aggregates
//     in order (chunks of 64*kResolveE tiles, pipelined across the waves, state
//     handed over through LDS) and publish every tile's prefix.  Workers read one
//     word.  All words are relaxed agent-scope 8-byte stores/loads: the data is
//     the flag;
//   * BitIndexer.write (json_structural_indexer.mojo:46-58) becomes a packed
//     (count|count<<16) wave scan, a per-lane ctz loop into a per-wave LDS staging
//     slice at the index's tile-relative position, and aligned 16-byte stores;
//   * software pipeline per wave: the ticket after next, the next tile's bytes and
//     the prefix of the tile to be emitted next are requested before the current
//     tile is computed, and the index emission of a tile is deferred by two tiles,
//     so ticket, HBM, prefix and store latencies overlap with compute.
//
// No MFMA (nothing here is a contraction); integer/bitwise work on u8 input,
// u64 masks, u32 output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/msj_stage1.h"
#include "lane_math.h"
#include "stage1_kernel.h"

namespace msj {

// ---- tile descriptors -------------------------------------------------------
// One 64-bit word per tile in each of two arrays.
// bits 63:62 status: 0 = not ready
// agg[t] (written by the tile's wave), status 1:
//   61 quote parity, 60 err(s_in=0), 59 err(s_in=1), 58 e_out, 57 ps_out,
//   56 utf8 err, 55 utf8 sequence pending at tile end, 54 poisoned (timeout),
//   30:15 count(s_in=1), 14:0 count(s_in=0)
// pre[t] (written by the resolver), status 2:
//   61 in_string before the tile, 60 unescaped err before, 56 utf8 err before,
//   54 poisoned, 31:0 structurals before the tile (launch-relative)
constexpr uint64_t kAgg = 1ull << 62;
constexpr uint64_t kPre = 2ull << 62;

__device__ __forceinline__ uint64_t ld_desc(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t bcast(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}

#ifdef MSJ_STAMPS
// Diagnostic build only: phase timestamps per tile (never compiled into the product .so).
#define MSJ_STAMP(t, k)                                                                   \
    do {                                                                                  \
        if ((threadIdx.x & 63u) == 0 && a.stamps)                                         \
            a.stamps[(uint64_t)(t) * 16 + (k)] = __builtin_amdgcn_s_memtime();            \
    } while (0)
#else
#define MSJ_STAMP(t, k) do {} while (0)
#endif

struct Shared {
    uint32_t role;
    uint32_t range_lo[2];    // worker workgroups: base tile of the next range (double buffered)
    // resolver hand-off between its waves
    uint32_t rs_seq, rs_s, rs_cnt, rs_err, rs_u8, rs_poison;
    uint32_t pad[3];
    // per-wave index staging for coalesced stores
    uint32_t stage[kWaves][kStageWords] __attribute__((aligned(16)));
};

// Bounded poll of one descriptor until its status is non-zero.
__device__ __forceinline__ uint64_t wait_desc(const uint64_t *p, uint32_t *timeout) {
    uint64_t d = ld_desc(p);
    uint32_t spins = 0;
    while ((d >> 62) == 0) {
        __builtin_amdgcn_s_sleep(2);
        d = ld_desc(p);
        if (++spins > kSpinLimit) {
            *timeout = 1;
            break;
        }
    }
    return d;
}

// The 64 bytes one lane owns, as loaded (4 x 16 B), plus one byte of the 64-byte
// window in front of the tile (lane l holds byte [tile_start - 64 + l]).
struct Block {
    uint4 q[4];
    uint32_t wb;
};

// Branch-free (so the compiler can leave all five loads in flight): pieces past
// the end are redirected to the last 16-B piece that starts inside the input;
// whatever they return is masked by `valid` in compute_tile.
__device__ __forceinline__ void load_block(const KernelArgs &a, uint32_t tile, uint32_t lane, Block &b) {
    const uint64_t blk_off = (uint64_t)tile * kTileBytes + (uint64_t)lane * 64u;
    const uint64_t last_piece = (a.len - 1u) & ~15ull;  // buf is 16-B aligned
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t off = blk_off + 16u * k;
        b.q[k] = *reinterpret_cast<const uint4 *>(a.buf + (off < last_piece ? off : last_piece));
    }
    const bool have_window = (tile > 0) || (a.flags & kFlagHasPrefix);
    const int64_t woff = have_window ? (int64_t)((uint64_t)tile * kTileBytes) - 64 + (int64_t)lane
                                     : (int64_t)(lane < a.len ? lane : a.len - 1u);
    b.wb = a.buf[woff];
}

// Forces the wait for prefetched registers HERE (their loads were issued a whole
// compute phase ago, so this costs nothing) instead of at their first use in the
// next iteration, where the vmcnt(0) the compiler needs would also wait for the
// stores issued in between (their number is data dependent, so a counted vmcnt
// is impossible).
__device__ __forceinline__ void touch_block(Block &b) {
#pragma unroll
    for (int k = 0; k < 4; k++)
        asm volatile("" : "+v"(b.q[k].x), "+v"(b.q[k].y), "+v"(b.q[k].z), "+v"(b.q[k].w));
    asm volatile("" : "+v"(b.wb));
}
__device__ __forceinline__ void touch_u64(uint64_t &v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    v = ((uint64_t)hi << 32) | lo;
}

// Ticket draw whose result is consumed much later.  atomicAdd() would be expanded
// into a wave-aggregated form whose result is needed (and waited for) at once;
// the asm form returns into lane 0's VGPR, which nothing reads until the explicit
// wait in ticket_value().
__device__ __forceinline__ uint32_t ticket_request(unsigned int *ctr, uint32_t lane, uint32_t count) {
    uint32_t ret = 0;
    if (lane == 0)
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ret) : "v"(ctr), "v"(count) : "memory");
    return ret;
}
__device__ __forceinline__ uint32_t ticket_value(uint32_t reg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return bcast(reg, 0);
}

// What a computed tile keeps in registers until its indices are emitted.
struct Pending {
    uint64_t T0, T1;     // structural_start masks for tile s_in = 0 / 1
    uint32_t excl;       // packed exclusive wave scan of the per-lane counts
    uint32_t tile_cnt;   // packed tile totals
    uint32_t tile;
};

// ---- one tile: masks, carries, counts; publishes the tile aggregate ------------
__device__ __forceinline__ Pending compute_tile(const KernelArgs &a, const uint32_t tile,
                                                const uint32_t lane, const Block &blk,
                                                uint32_t &timeout, uint64_t &agg_word) {
    uint64_t *agg = a.ws + kDescOffset;
    const uint64_t len = a.len;
    const uint64_t tile_start = (uint64_t)tile * kTileBytes;
    const uint64_t blk_off = tile_start + (uint64_t)lane * 64u;
    MSJ_STAMP(tile, 1);

    uint32_t x[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        x[4 * k + 0] = blk.q[k].x;
        x[4 * k + 1] = blk.q[k].y;
        x[4 * k + 2] = blk.q[k].z;
        x[4 * k + 3] = blk.q[k].w;
    }
    uint64_t valid;  // bytes past the end behave as the reference's 0x20 padding (:103-107)
    if (blk_off + 64u <= len)
        valid = ~0ull;
    else if (blk_off < len)
        valid = (1ull << (len - blk_off)) - 1ull;
    else
        valid = 0ull;

    // ---- carries into the tile from the 64 bytes in front of it (wave-uniform)
    const bool have_window = (tile > 0) || (a.flags & kFlagHasPrefix);
    uint32_t tile_e_in, tile_ps_in, tile_u8_in;
    {
        const uint32_t wb = blk.wb;
        // utf8 carry word of the window's last bytes (lane_math.h layout)
        {
            const bool l234 = (wb >= 0xC0u) && (wb < 0xF8u);
            const bool l34 = (wb >= 0xE0u) && (wb < 0xF8u);
            const bool l4 = (wb >= 0xF0u) && (wb < 0xF8u);
            const uint64_t m234 = __ballot(l234), m34 = __ballot(l34), m4 = __ballot(l4);
            const uint64_t mE0 = __ballot(wb == 0xE0u), mED = __ballot(wb == 0xEDu);
            const uint64_t mF0 = __ballot(wb == 0xF0u), mF4 = __ballot(wb == 0xF4u);
            tile_u8_in = (uint32_t)(m234 >> 63) | ((uint32_t)(m34 >> 62) << 1) |
                         ((uint32_t)(m4 >> 61) << 3) | ((uint32_t)(mE0 >> 63) << 6) |
                         ((uint32_t)(mED >> 63) << 7) | ((uint32_t)(mF0 >> 63) << 8) |
                         ((uint32_t)(mF4 >> 63) << 9);
            if (!have_window) tile_u8_in = 0;
        }
        if (tile == 0) {
            // exact state at the first byte of this launch
            tile_e_in = a.carry_in->next_is_escaped & 1u;
            tile_ps_in = a.carry_in->prev_scalar & 1u;
        } else {
            const uint64_t WB = __ballot(wb == 0x5Cu);
            const uint64_t WQ = __ballot(wb == 0x22u);
            const bool nonscalar = (wb == 0x20u) | (wb == 0x09u) | (wb == 0x0Au) | (wb == 0x0Du) |
                                   (wb == 0x0Cu) | (wb == 0x1Au) | (wb == 0x2Cu) | (wb == 0x3Au) |
                                   (wb == 0x5Bu) | (wb == 0x5Du) | (wb == 0x7Bu) | (wb == 0x7Du);
            const uint64_t WNS = __ballot(nonscalar);
            const uint32_t r = top_run(WB);  // backslashes ending at byte[-1]
            bool resolved = (r != 64u);
            tile_e_in = r & 1u;
            if (r >= 1u) {
                tile_ps_in = 1u;  // byte[-1] is a backslash: a non-quote scalar
            } else if ((WNS >> 63) & 1u) {
                tile_ps_in = 0u;
            } else if (!((WQ >> 63) & 1u)) {
                tile_ps_in = 1u;
            } else {
                // byte[-1] is '"': a real quote unless escaped by an odd run before it.
                // (WB<<1)|1 has bit 0 forced: a result of 64 means bits 1..63 are all set.
                const uint32_t r2 = top_run((WB << 1) | 1ull);  // run ending at byte[-2]
                if (r2 == 64u) resolved = false;
                tile_ps_in = r2 & 1u;
            }
            if (!resolved) {
                // >= 62 consecutive backslashes in front of the tile: take the exact
                // carries the predecessor publishes with its aggregate.
                uint32_t to = 0;
                const uint64_t d = wait_desc(&agg[tile - 1], &to);
                if (to) timeout = 1;
                tile_e_in = (uint32_t)(d >> 58) & 1u;
                tile_ps_in = (uint32_t)(d >> 57) & 1u;
            }
        }
    }

    MSJ_STAMP(tile, 2);
    // ---- bit-planes and character classes (lane_math.h)
    uint64_t p[8];
    bitplanes(x, p);
#pragma unroll
    for (int k = 0; k < 8; k++) p[k] &= valid;
    const Classes cls = classify(p, valid);

    // ---- escape carry, lane level: g = carry-out if carry-in were 0, pr = all 64
    //      bytes are backslashes (carry propagates).  Wave level: carry-lookahead add.
    const uint32_t tr = top_run(cls.backslash);
    const uint64_t G = __ballot((tr & 1u) != 0u);  // tr == 64 -> 0
    const uint64_t Pm = __ballot(tr == 64u);
    const uint64_t add_a = G | Pm, add_b = G;
    const uint64_t add_s = add_a + add_b + tile_e_in;
    const uint32_t tile_e_out = (uint32_t)(((add_a & add_b) | ((add_a | add_b) & ~add_s)) >> 63);
    const uint64_t carries = add_s ^ add_a ^ add_b;
    const uint32_t lane_e_in = (uint32_t)(carries >> lane) & 1u;
    MSJ_STAMP(tile, 3);

    // ---- strings (json_string_scanner.mojo:55-69) with the lane's exact escape carry
    uint32_t lane_e_out;
    const uint64_t escaped = escaped_mask(cls.backslash, lane_e_in, &lane_e_out);
    const uint64_t quote = cls.quote_chr & ~escaped;
    const uint64_t S0 = prefix_xor(quote);  // in_string if the lane started outside a string
    const uint64_t PM = __ballot((S0 >> 63) != 0);
    const uint32_t lane_par = lanes_below(PM) & 1u;  // parity of the lanes before me in the tile
    const uint32_t tile_par = (uint32_t)__popcll(PM) & 1u;

    // ---- scalars (json_scanner.mojo:64-79)
    const uint64_t scalar = ~(cls.op | cls.ws);
    const uint64_t nqs = scalar & ~quote;
    const uint32_t my_ps = (uint32_t)(nqs >> 63);
    uint32_t prev_ps = __shfl_up(my_ps, 1);
    if (lane == 0) prev_ps = tile_ps_in;
    const uint32_t tile_ps_out = bcast(my_ps, 63);

    const uint64_t lane_in = (uint64_t)(-(int64_t)lane_par);  // all-ones: inside a string
    // in_string / string_tail assuming the TILE starts outside a string
    const uint64_t in_string0 = S0 ^ lane_in;
    const uint64_t string_tail0 = in_string0 ^ quote;  // json_string_scanner.mojo:40-44
    const uint64_t follows = (nqs << 1) | prev_ps;     // json_scanner.mojo:76-79
    const uint64_t potential = cls.op | (scalar & ~follows);
    Pending r;
    r.T0 = potential & ~string_tail0;  // structural_start if tile s_in = 0
    r.T1 = potential & string_tail0;   //                  if tile s_in = 1
    const bool err0 = (cls.ctrl & in_string0) != 0;   // json_structural_indexer.mojo:143-145
    const bool err1 = (cls.ctrl & ~in_string0) != 0;
    MSJ_STAMP(tile, 4);

    // ---- utf8
    uint32_t tile_pend = 0;
    bool u8err = false;
    if (!(a.flags & kFlagNoUtf8)) {
        const Utf8Planes u8p = utf8_planes(p);
        const uint32_t my_u8c = utf8_carry_out(u8p);
        uint32_t prev_u8c = __shfl_up(my_u8c, 1);
        if (lane == 0) prev_u8c = tile_u8_in;
        u8err = utf8_errors(p, u8p, prev_u8c) != 0;
        tile_pend = (bcast(my_u8c, 63) & 0x3Fu) ? 1u : 0u;
    }
    MSJ_STAMP(tile, 5);

    // ---- packed inclusive scan of the per-lane structural counts
    const uint32_t pk = (uint32_t)__popcll(r.T0) | ((uint32_t)__popcll(r.T1) << 16);
    uint32_t inc = pk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += t;
    }
    r.excl = inc - pk;
    r.tile_cnt = bcast(inc, 63);
    r.tile = tile;
    const uint64_t me0 = __ballot(err0), me1 = __ballot(err1), mu8 = __ballot(u8err);
    MSJ_STAMP(tile, 6);
    agg_word = kAgg | ((uint64_t)tile_par << 61) | ((uint64_t)(me0 ? 1u : 0u) << 60) |
               ((uint64_t)(me1 ? 1u : 0u) << 59) | ((uint64_t)tile_e_out << 58) |
               ((uint64_t)tile_ps_out << 57) | ((uint64_t)(mu8 ? 1u : 0u) << 56) |
               ((uint64_t)tile_pend << 55) | ((uint64_t)timeout << 54) |
               ((uint64_t)(r.tile_cnt >> 16) << 15) | (uint64_t)(r.tile_cnt & 0xFFFFu);
    return r;
}

// ---- BitIndexer.write (json_structural_indexer.mojo:46-58) for one computed tile:
//      the ascending offsets are staged in this wave's LDS slice at their
//      tile-relative position and written out as aligned 16-byte stores (one L2
//      request per 64 B instead of one per index).  Wave-local: no barrier.
__device__ __forceinline__ void emit_tile(const KernelArgs &a, uint32_t *stage, const uint32_t lane,
                                          const Pending &r, const uint64_t pre_word,
                                          const uint64_t count0, uint32_t &timeout) {
    const uint64_t *pre = a.ws + kDescOffset + a.ntiles;
    MSJ_STAMP(r.tile, 8);
    // pre_word was requested a whole compute phase ago; poll only if the resolver
    // had not published this tile's prefix yet at that time.
    uint64_t d = pre_word;
    if ((d >> 62) == 0ull) {
        uint32_t to = 0;
        d = wait_desc(&pre[r.tile], &to);
        if (to) timeout = 1;
    }
    if ((d >> 54) & 1u) timeout = 1;
    MSJ_STAMP(r.tile, 9);
    if ((a.flags & kFlagNoEmit) || timeout) return;
    const uint32_t s_in = (uint32_t)(d >> 61) & 1u;
    const uint64_t base = count0 + (uint64_t)(uint32_t)d;
    const uint64_t T = s_in ? r.T1 : r.T0;
    const uint32_t lane_off = s_in ? (r.excl >> 16) : (r.excl & 0xFFFFu);
    const uint32_t my_cnt = s_in ? (r.tile_cnt >> 16) : (r.tile_cnt & 0xFFFFu);
    const bool fits = base + my_cnt <= a.capacity;
    const uint32_t shift = (uint32_t)(base & 3u);  // stage[j] <-> idx[base - shift + j]
    const uint32_t vend = shift + my_cnt;
    const uint32_t v0 = (uint32_t)((uint64_t)r.tile * kTileBytes) + lane * 64u;
    uint32_t vpos = shift + lane_off;
    uint32_t tlo = (uint32_t)T, thi = (uint32_t)(T >> 32);
    for (uint32_t r0 = 0; r0 < vend; r0 += kStageWords) {
        const uint32_t r1 = r0 + kStageWords;
        while (tlo && vpos < r1) {
            stage[vpos - r0] = v0 + (uint32_t)__builtin_ctz(tlo);
            tlo &= tlo - 1;
            vpos++;
        }
        if (!tlo) {
            while (thi && vpos < r1) {
                stage[vpos - r0] = v0 + 32u + (uint32_t)__builtin_ctz(thi);
                thi &= thi - 1;
                vpos++;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t lim = vend < r1 ? vend : r1;
        const uint64_t gbase = base - shift + r0;
        for (uint32_t q = lane; 4u * q < lim - r0; q += 64u) {
            const uint32_t vq = r0 + 4u * q;
            const uint4 val = *reinterpret_cast<const uint4 *>(&stage[4u * q]);
            const uint64_t g = gbase + 4u * q;
            if (fits && vq >= shift && vq + 4u <= lim) {
                *reinterpret_cast<uint4 *>(&a.idx[g]) = val;
            } else {
                const uint32_t vv[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t v = vq + j;
                    if (v >= shift && v < lim && g + j < a.capacity) a.idx[g + j] = vv[j];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // stage is reused by the next round / next tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    MSJ_STAMP(r.tile, 10);
}

// ---- worker: one wave, persistent ---------------------------------------------------
// Work distribution.  One atomic counter hands out tiles in ascending order, but a
// single word sustains only ~80-90 returning atomics per microsecond chip-wide, so one
// atomic must pay for many tiles: thread 0 of a workgroup draws a RANGE of
// kWaves * kBatch tiles, and wave w takes tiles lo + kWaves*j + w (j = 0..kBatch-1).
// The emission of a tile is deferred by exactly kBatch tiles of the same wave, i.e.
// to the same slot of the next range.  (kBatch <= deferral depth matters: a wave
// that had to emit inside its own range would need that range's first prefix, hence
// every lower range complete, and ranges would serialise.)  The four waves of a
// workgroup meet at one barrier per range to pick up the next range's base from LDS;
// everything else in the worker path is wave-local.
//
// Deadlock freedom: a wave only holds tiles once it is running, handles them in
// increasing order, never blocks while computing, and the resolver publishes a
// tile's prefix as soon as every earlier tile is in (partial progress).  The wave
// holding the smallest not-yet-computed tile is therefore never waiting on anything
// that needs a later tile, whatever the dispatch order or residency.
__device__ __forceinline__ void worker_wave(const KernelArgs &a, Shared &sh, const uint32_t lane,
                                            const uint32_t wave) {
    unsigned int *ticket_ctr = reinterpret_cast<unsigned int *>(a.ws);
    uint32_t *stage = sh.stage[wave];
    const uint32_t tid = threadIdx.x;
    const uint32_t ntiles = a.ntiles;
    constexpr uint32_t kRange = kWaves * kBatch;
    if (tid == 0) {
        sh.range_lo[0] = atomicAdd(ticket_ctr, kRange);
        sh.range_lo[1] = atomicAdd(ticket_ctr, kRange);
    }
    __syncthreads();
    uint32_t lo_cur = sh.range_lo[0], lo_next = sh.range_lo[1];
    const uint64_t count0 = a.carry_in->count;  // launch invariant: read once
    const uint64_t *pre = a.ws + kDescOffset + ntiles;
    uint32_t timeout = 0;

    Block cur;
    load_block(a, lo_cur + wave < ntiles ? lo_cur + wave : ntiles - 1u, lane, cur);
    touch_block(cur);  // loop invariant: `cur` has arrived (no vmcnt wait on it inside the loop)
    Pending pend[kBatch];
    bool has[kBatch];
#pragma unroll
    for (uint32_t j = 0; j < kBatch; j++) has[j] = false;
    uint64_t pre_next = 0;  // prefix word of the tile emitted in the next iteration
    uint32_t r = 0;
    while (lo_cur < ntiles) {  // uniform across the workgroup
        // range r+2, requested now, needed at the end of this range
        uint32_t req_reg = 0;
        if (tid == 0) req_reg = ticket_request(ticket_ctr, 0u, kRange);
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) {
            const uint32_t t_cur = lo_cur + kWaves * j + wave;
            const uint32_t t_nxt = (j + 1u < kBatch) ? t_cur + kWaves : lo_next + wave;
            MSJ_STAMP(t_cur < ntiles ? t_cur : ntiles - 1u, 0);
            if (has[j]) emit_tile(a, stage, lane, pend[j], pre_next, count0, timeout);
            has[j] = false;
            // request the next tile's bytes and the prefix of the tile emitted next
            Block nxt;  // past the last tile: harmless re-read of the last tile (branch-free)
            load_block(a, t_nxt < ntiles ? t_nxt : ntiles - 1u, lane, nxt);
            const uint32_t jn = (j + 1u) % kBatch;
            pre_next = ld_desc(&pre[has[jn] ? pend[jn].tile : 0u]);
            MSJ_STAMP(t_cur < ntiles ? t_cur : ntiles - 1u, 11);
            uint64_t agg_word = 0;
            if (t_cur < ntiles) {
                pend[j] = compute_tile(a, t_cur, lane, cur, timeout, agg_word);
                has[j] = true;
            }
            // everything requested above has had a whole compute phase to arrive; wait
            // for it before this iteration's first store goes into the queue
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            touch_block(nxt);
            touch_u64(pre_next);
            if (!has[jn]) pre_next = 0;  // nothing pending in that slot: the word read was a dummy
            if (lane == 0 && t_cur < ntiles) st_desc(&a.ws[kDescOffset + t_cur], agg_word);
            MSJ_STAMP(t_cur < ntiles ? t_cur : ntiles - 1u, 7);
            cur = nxt;
        }
        if (tid == 0) sh.range_lo[r & 1u] = req_reg;  // arrived: vmcnt(0) above
        __syncthreads();
        lo_cur = lo_next;
        lo_next = sh.range_lo[r & 1u];
        r++;
    }
#pragma unroll
    for (uint32_t j = 0; j < kBatch; j++) {
        if (has[j]) {
            // the prefix word was only prefetched for the first one
            emit_tile(a, stage, lane, pend[j], j == 0 ? pre_next : 0ull, count0, timeout);
        }
    }
}

// ---- resolver: the four waves of one workgroup turn tile aggregates into tile
// prefixes, in order.  Monoid: a tile's aggregate is (parity p, count c[q], error
// e[q]) for incoming in-string state q; composing left to right gives every tile
// its incoming state and the number of structurals before it.  Wave w owns chunks
// w, w+4, ... of kResolveChunk tiles: it polls its chunk until every aggregate is
// there, folds kResolveE consecutive tiles per lane (for both q), then takes the
// running state from LDS (published by the wave that owns the previous chunk),
// combines the lanes with a ballot (parity) and a shuffle scan (counts), hands
// the new state on, and only then writes its tiles' prefix words -- so the memory
// latency of four chunks overlaps.  finish() (json_structural_indexer.mojo:147-186)
// runs in the wave that owns the last chunk.
__device__ void resolver(const KernelArgs &a, Shared &sh) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint64_t *agg = a.ws + kDescOffset;
    uint64_t *pre = a.ws + kDescOffset + a.ntiles;
    const uint32_t ntiles = a.ntiles;
    const uint32_t nchunks = (ntiles + kResolveChunk - 1) / kResolveChunk;
    if (tid == 0) {
        sh.rs_seq = 0;
        sh.rs_s = a.carry_in->in_string & 1u;
        sh.rs_cnt = 0;
        sh.rs_err = 0;
        sh.rs_u8 = 0;
        sh.rs_poison = 0;
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    volatile uint32_t *seq = &sh.rs_seq;
    __builtin_amdgcn_s_setprio(3);  // the serial chain of the whole launch runs here
    for (uint32_t c = wave; c < nchunks; c += kWaves) {
        const uint32_t first = c * kResolveChunk + lane * kResolveE;
        uint64_t d[kResolveE];
#pragma unroll
        for (int e = 0; e < kResolveE; e++) d[e] = 0;
        uint32_t spins = 0, published = 0, force = 0;
        bool have_state = false, full = false;
        uint32_t s = 0, cnt = 0, err = 0, u8 = 0, poison = 0;
        uint32_t s_new = 0, cnt_new = 0, err_new = 0, u8_new = 0, poison_new = 0;
        uint32_t rl = 0, fl = 64u, m = 0;
        // lane aggregate of my kResolveE tiles under both incoming states (full chunk)
        uint32_t fs0 = 0, fc_0 = 0, fc_1 = 0, fe_0 = 0, fe_1 = 0;
        uint64_t fPM = 0, fUM = 0, fXM = 0;
        bool agg_done = false;
        for (;;) {
            if (!full) {
                // (re)load the aggregates that were not there yet; past the end: identity
                rl = 0;
                bool run = true;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    if ((d[e] >> 62) == 0ull) {
                        d[e] = (first + e < ntiles) ? ld_desc(&agg[first + e]) : kAgg;
                        if (force && (d[e] >> 62) == 0ull) d[e] = kAgg | (1ull << 54);  // gave up
                    }
                    run = run && ((d[e] >> 62) != 0ull);
                    rl += run ? 1u : 0u;
                }
                // m = number of leading tiles of the chunk whose aggregates are all there
                const uint64_t notfull = __ballot(rl < (uint32_t)kResolveE);
                fl = notfull ? (uint32_t)__builtin_ctzll(notfull) : 64u;
                m = (fl == 64u) ? kResolveChunk : fl * kResolveE + (uint32_t)__shfl((int)rl, (int)fl);
                full = (m == kResolveChunk);
            }
            if (full && !agg_done) {
                // everything that does not need the running state, done before it arrives
                uint32_t s0 = 0, s1 = 1, lu = 0, lpoison = 0;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    const uint64_t de = d[e];
                    const uint32_t p = (uint32_t)(de >> 61) & 1u;
                    const uint32_t c0 = (uint32_t)de & 0x7FFFu, c1 = (uint32_t)(de >> 15) & 0xFFFFu;
                    const uint32_t e0 = (uint32_t)(de >> 60) & 1u, e1 = (uint32_t)(de >> 59) & 1u;
                    fc_0 += s0 ? c1 : c0;
                    fe_0 |= s0 ? e1 : e0;
                    s0 ^= p;
                    fc_1 += s1 ? c1 : c0;
                    fe_1 |= s1 ? e1 : e0;
                    s1 ^= p;
                    lu |= (uint32_t)(de >> 56) & 1u;
                    lpoison |= (uint32_t)(de >> 54) & 1u;
                }
                fs0 = s0;
                fPM = __ballot((s0 & 1u) != 0u);
                fUM = __ballot(lu != 0u);
                fXM = __ballot(lpoison != 0u);
                agg_done = true;
            }
            if (!have_state && *seq == c) {
                // the owner of the previous chunk has handed the running state over
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                s = sh.rs_s;
                cnt = sh.rs_cnt;
                err = sh.rs_err;
                u8 = sh.rs_u8;
                poison = sh.rs_poison;
                have_state = true;
            }
            if (have_state && full) {
                // ---- fast finish: the serial section of the whole launch
                const uint32_t in_l = s ^ ((uint32_t)__popcll(fPM & below) & 1u);
                const uint32_t mycnt = in_l ? fc_1 : fc_0;
                uint32_t incl = mycnt;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t t = __shfl_up(incl, dd);
                    if (lane >= (uint32_t)dd) incl += t;
                }
                const uint64_t EM = __ballot((in_l ? fe_1 : fe_0) != 0u);
                s_new = s ^ ((uint32_t)__popcll(fPM) & 1u);
                cnt_new = cnt + bcast(incl, 63);
                err_new = err | (EM ? 1u : 0u);
                u8_new = u8 | (fUM ? 1u : 0u);
                poison_new = poison | (fXM ? 1u : 0u);
                if (lane == 0) {
                    // hand the running state to the owner of the next chunk first; this
                    // chunk's prefix words are written afterwards
                    sh.rs_s = s_new;
                    sh.rs_cnt = cnt_new;
                    sh.rs_err = err_new;
                    sh.rs_u8 = u8_new;
                    sh.rs_poison = poison_new;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    *seq = c + 1u;
                }
                uint32_t cs = in_l, cb = cnt + (incl - mycnt);
                uint32_t ce = err | ((EM & below) ? 1u : 0u);
                uint32_t cu = u8 | ((fUM & below) ? 1u : 0u);
                const uint64_t pz = (uint64_t)(poison_new ? 1u : 0u) << 54;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    const uint32_t idx_in_chunk = lane * kResolveE + (uint32_t)e;
                    if (idx_in_chunk >= published && first + e < ntiles) {
                        st_desc(&pre[first + e], kPre | ((uint64_t)cs << 61) | ((uint64_t)ce << 60) |
                                                     ((uint64_t)cu << 56) | pz | (uint64_t)cb);
                    }
                    const uint64_t de = d[e];
                    const uint32_t p = (uint32_t)(de >> 61) & 1u;
                    const uint32_t c0 = (uint32_t)de & 0x7FFFu, c1 = (uint32_t)(de >> 15) & 0xFFFFu;
                    const uint32_t e0 = (uint32_t)(de >> 60) & 1u, e1 = (uint32_t)(de >> 59) & 1u;
                    cb += cs ? c1 : c0;
                    ce |= cs ? e1 : e0;
                    cu |= (uint32_t)(de >> 56) & 1u;
                    cs ^= p;
                }
                (void)fs0;
                break;
            }
            if (have_state && m > published) {
                // PARTIAL PROGRESS (chunk not complete): every tile whose predecessors are
                // all in gets its prefix now.  A worker may be waiting for a prefix while it
                // still holds a later, not yet computed tile of this same chunk; with partial
                // progress the smallest not-yet-computed tile can always proceed, which
                // rules out deadlock.
                const uint32_t act = (lane < fl) ? (uint32_t)kResolveE : ((lane == fl) ? rl : 0u);
                uint32_t s0 = 0, s1 = 1, c_0 = 0, c_1 = 0, e_0 = 0, e_1 = 0, lu = 0, lpoison = 0;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    if ((uint32_t)e < act) {
                        const uint64_t de = d[e];
                        const uint32_t p = (uint32_t)(de >> 61) & 1u;
                        const uint32_t c0 = (uint32_t)de & 0x7FFFu, c1 = (uint32_t)(de >> 15) & 0xFFFFu;
                        const uint32_t e0 = (uint32_t)(de >> 60) & 1u, e1 = (uint32_t)(de >> 59) & 1u;
                        c_0 += s0 ? c1 : c0;
                        e_0 |= s0 ? e1 : e0;
                        s0 ^= p;
                        c_1 += s1 ? c1 : c0;
                        e_1 |= s1 ? e1 : e0;
                        s1 ^= p;
                        lu |= (uint32_t)(de >> 56) & 1u;
                        lpoison |= (uint32_t)(de >> 54) & 1u;
                    }
                }
                const uint64_t PM = __ballot((s0 & 1u) != 0u);  // lane parity (inactive lanes: 0)
                const uint32_t in_l = s ^ ((uint32_t)__popcll(PM & below) & 1u);
                const uint32_t mycnt = in_l ? c_1 : c_0;
                uint32_t incl = mycnt;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const uint32_t t = __shfl_up(incl, dd);
                    if (lane >= (uint32_t)dd) incl += t;
                }
                const uint64_t EM = __ballot((in_l ? e_1 : e_0) != 0u);
                const uint64_t UM = __ballot(lu != 0u);
                const uint64_t XM = __ballot(lpoison != 0u);
                uint32_t cs = in_l, cb = cnt + (incl - mycnt);
                uint32_t ce = err | ((EM & below) ? 1u : 0u);
                uint32_t cu = u8 | ((UM & below) ? 1u : 0u);
                const uint64_t pz = (uint64_t)(poison | ((XM & below) ? 1u : 0u) | lpoison) << 54;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    if ((uint32_t)e < act) {
                        const uint32_t idx_in_chunk = lane * kResolveE + (uint32_t)e;
                        if (idx_in_chunk >= published && first + e < ntiles) {
                            st_desc(&pre[first + e], kPre | ((uint64_t)cs << 61) | ((uint64_t)ce << 60) |
                                                         ((uint64_t)cu << 56) | pz | (uint64_t)cb);
                        }
                        const uint64_t de = d[e];
                        const uint32_t p = (uint32_t)(de >> 61) & 1u;
                        const uint32_t c0 = (uint32_t)de & 0x7FFFu, c1 = (uint32_t)(de >> 15) & 0xFFFFu;
                        const uint32_t e0 = (uint32_t)(de >> 60) & 1u, e1 = (uint32_t)(de >> 59) & 1u;
                        cb += cs ? c1 : c0;
                        ce |= cs ? e1 : e0;
                        cu |= (uint32_t)(de >> 56) & 1u;
                        cs ^= p;
                    }
                }
                published = m;
                spins = 0;
            }
            if (++spins > kSpinLimit) force = 1;  // next round substitutes poisoned identities
            // full but no state yet: spin on the LDS word only (no global traffic)
            if (!full) __builtin_amdgcn_s_sleep(1);
        }
        if (c + 1u == nchunks && lane == 0) {
            // ---- finish(): json_structural_indexer.mojo:147-186
            const msj_carry cin = *a.carry_in;
            const uint64_t last = ld_desc(&agg[ntiles - 1]);
            const bool do_utf8 = !(a.flags & kFlagNoUtf8);
            msj_carry out;
            const uint64_t n = cin.count + cnt_new;
            out.count = n;
            out.bytes = cin.bytes + a.len;
            out.in_string = s_new;
            out.next_is_escaped = (uint32_t)(last >> 58) & 1u;
            out.prev_scalar = (uint32_t)(last >> 57) & 1u;
            out.unescaped_error = (cin.unescaped_error | err_new) ? 1u : 0u;
            uint32_t u8e = cin.utf8_error | u8_new;
            // a multi-byte sequence cut exactly at the end of the last full tile
            if ((a.flags & kFlagFinal) && do_utf8 && (a.len % kTileBytes) == 0 && ((last >> 55) & 1u))
                u8e = 1;
            out.utf8_error = u8e ? 1u : 0u;
            out.internal_error = (cin.internal_error | poison_new) ? 1u : 0u;
            int32_t code = MSJ_SUCCESS;
            if (a.flags & kFlagFinal) {
                if (out.internal_error) {
                    code = MSJ_UNEXPECTED_ERROR;
                } else if (s_new) {
                    code = MSJ_UNCLOSED_STRING;  // :151-155
                } else if (out.unescaped_error) {
                    code = MSJ_UNESCAPED_CHARS;  // :157-158
                } else if (n + 3 > a.capacity) {
                    code = MSJ_CAPACITY;
                } else {
                    if (!(a.flags & kFlagNoEmit)) {
                        a.idx[n] = (uint32_t)a.trailer_len;      // :167-169
                        a.idx[n + 1] = (uint32_t)a.trailer_len;  // :170-172
                        a.idx[n + 2] = 0;                        // :173
                    }
                    if (n == 0)
                        code = MSJ_EMPTY;  // :176-177
                    else if ((a.flags & kFlagStrictUtf8) && out.utf8_error)
                        code = MSJ_UTF8_ERROR;
                }
            }
            out.code = code;
            for (int k = 0; k < 5; k++) out.reserved[k] = 0;
            *a.carry_out = out;
            if (a.segment) {
                a.segment->byte_base = a.segment_byte_base;
                a.segment->byte_len = a.len;
                a.segment->index_begin = cin.count;
                a.segment->count = cnt_new;
            }
        }
    }
}

__global__ __launch_bounds__(kThreads) void stage1_kernel(const KernelArgs a) {
    __shared__ Shared sh;
    const uint32_t tid = threadIdx.x;
    // The first workgroup to get here becomes the resolver (it is running, so the
    // workers that wait on its output can always make progress); every wave of
    // every other workgroup is an independent worker.
    if (tid == 0) sh.role = atomicAdd(reinterpret_cast<unsigned int *>(a.ws) + 2, 1u);
    __syncthreads();
    if (sh.role == 0u) {
        resolver(a, sh);
        return;
    }
    worker_wave(a, sh, tid & 63u, tid >> 6);
}

}  // namespace msj

extern "C" int msj_launch_stage1(const msj::KernelArgs *args, void *stream, uint32_t grid) {
    const msj::KernelArgs a = *args;
    // persistent workgroups of kWaves worker waves, plus the resolver workgroup
    const uint32_t need = (a.ntiles + msj::kWaves - 1u) / msj::kWaves + 1u;
    const uint32_t g = (grid == 0 || grid > need) ? need : grid;
    hipLaunchKernelGGL(msj::stage1_kernel, dim3(g < 2u ? 2u : g), dim3(msj::kThreads), 0,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int msj_stage1_occupancy(int *blocks_per_cu) {
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, msj::stage1_kernel,
                                                            msj::kThreads, 0);
}
