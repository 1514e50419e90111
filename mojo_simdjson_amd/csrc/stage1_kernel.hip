// stage1_kernel.hip -- single-pass stage-1 structural indexer for gfx950 (MI355X).
//
// Replaces the reference's serial block loop
//   JsonStructuralIndexer.index[128] / step / next / finish
//   (src/mojo_simdjson/generic/stage1/json_structural_indexer.mojo:81-186)
// by one kernel launch over the whole buffer:
//
//   * persistent workgroups (4 wave64 = 256 lanes) draw 16 KiB tiles from an
//     ordered ticket counter; every lane owns one 64-byte block = the unit of one
//     JsonScanner.next call, and all masks are uint64 with the reference's bit
//     order (lane_math.h);
//   * the three 1-bit carries the reference threads through its loop
//     (next_is_escaped json_escape_scanner.mojo:13, prev_in_string
//     json_string_scanner.mojo:49, prev_scalar json_scanner.mojo:57) are
//     resolved lane -> wave -> workgroup with __ballot + a 64-bit
//     carry-lookahead add (escape), ballot/mbcnt prefix parity (in-string) and
//     a one-lane shuffle (prev_scalar); the escape / prev_scalar carries into a
//     tile are derived locally from the 64 bytes in front of it;
//   * across tiles only the in-string bit and the running structural count are
//     chained.  Each tile publishes a 64-bit aggregate (parity, count and error
//     bit for both possible incoming in-string states); the workgroup holding
//     ticket 0 does not index anything: it is the RESOLVER, whose four waves
//     fold those aggregates in order (chunks of 256 tiles, pipelined across the
//     waves, state handed over through LDS) and publish every tile's prefix.
//     Workers poll one word.  All words are relaxed agent-scope 8-byte
//     stores/loads: the data is the flag;
//   * BitIndexer.write (json_structural_indexer.mojo:46-58) becomes a packed
//     (count|count<<16) wave scan, a per-lane ctz loop into an LDS staging
//     buffer at the index's tile-relative position, and aligned 16-byte stores;
//   * software pipeline per workgroup: the ticket after next and the next
//     tile's bytes are requested before the current tile is computed, and the
//     index emission of tile i is deferred until tile i+1 has been computed, so
//     ticket, HBM and prefix latencies overlap with compute.
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
// agg[t] (written by the tile's workgroup), status 1:
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

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

#ifdef MSJ_STAMPS
// Diagnostic build only: phase timestamps per tile (never compiled into the product .so).
#define MSJ_STAMP(t, k)                                                                   \
    do {                                                                                  \
        if (threadIdx.x == 0 && a.stamps)                                                 \
            a.stamps[(uint64_t)(t) * 16 + (k)] = __builtin_amdgcn_s_memtime();            \
    } while (0)
#else
#define MSJ_STAMP(t, k) do {} while (0)
#endif

struct Shared {
    uint32_t tk[4];          // tickets: [0],[1] initial pair, [2] the one requested last
    uint32_t tile_e_in, tile_ps_in, tile_u8_in;
    uint32_t esc[kWaves];    // bit0 = escape carry-out if carry-in 0, bit1 = if carry-in 1
    uint32_t par[kWaves];    // quote parity of the wave
    uint32_t ps[kWaves];     // prev_scalar out of the wave's last lane
    uint32_t u8c[kWaves];    // utf8 carry word out of the wave's last lane
    uint32_t cnt[kWaves];    // packed structural counts (s_in=0 | s_in=1 << 16)
    uint32_t flg[kWaves];    // bit0 err(s_in=0) bit1 err(s_in=1) bit2 utf8 err
    uint32_t s_in;
    uint32_t timeout;
    uint64_t base;           // absolute output position of the tile's first index
    // resolver hand-off between its waves
    uint32_t rs_seq, rs_s, rs_cnt, rs_err, rs_u8, rs_poison;
    uint32_t stage[kStageWords] __attribute__((aligned(16)));  // index staging for coalesced stores
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

// The 64 bytes one lane owns, as loaded (4 x 16 B).
struct Block {
    uint4 q[4];
    uint32_t wb;  // wave 0 only: byte [tile_start - 64 + lane] (the look-back window)
};

// Branch-free (so the compiler can leave all five loads in flight): pieces past
// the end are redirected to the last 16-B piece that starts inside the input;
// whatever they return is masked by `valid` in compute_tile.
__device__ __forceinline__ void load_block(const KernelArgs &a, uint32_t tile, Block &b) {
    const uint64_t blk_off = (uint64_t)tile * kTileBytes + (uint64_t)threadIdx.x * 64u;
    const uint64_t last_piece = (a.len - 1u) & ~15ull;  // buf is 16-B aligned
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint64_t off = blk_off + 16u * k;
        b.q[k] = *reinterpret_cast<const uint4 *>(a.buf + (off < last_piece ? off : last_piece));
    }
    const bool have_window = (tile > 0) || (a.flags & kFlagHasPrefix);
    const uint64_t lane = threadIdx.x & 63u;
    const int64_t woff = have_window ? (int64_t)((uint64_t)tile * kTileBytes) - 64 + (int64_t)lane
                                     : (int64_t)(lane < a.len ? lane : a.len - 1u);
    b.wb = a.buf[woff];
}

// Ticket draw whose result is consumed much later.  atomicAdd() would be expanded
// into a wave-aggregated form whose result is needed (and waited for) at once;
// the asm form returns into a VGPR that nothing reads until ticket_ready().
__device__ __forceinline__ uint32_t ticket_request(unsigned int *ctr) {
    uint32_t ret = 0;
    const uint32_t one = 1;
    if (threadIdx.x == 0)
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ret) : "v"(ctr), "v"(one) : "memory");
    return ret;
}
__device__ __forceinline__ void ticket_ready() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// What a computed tile keeps in registers until its indices are emitted.
struct Pending {
    uint64_t T0, T1;     // structural_start masks for tile s_in = 0 / 1
    uint32_t inc, pk;    // packed inclusive wave scan / own packed count
    uint32_t wave_off;   // packed counts of the waves before mine
    uint32_t tile_cnt;   // packed tile totals
    uint32_t tile;
};

// ---- one tile: masks, carries, counts; publishes the tile aggregate ------------
// Forces the wait for a prefetched block HERE (its loads were issued a whole
// compute phase ago, so this costs nothing) instead of at its first use in the
// next iteration, where the vmcnt(0) the compiler needs would also wait for the
// index stores issued in between (the number of stores is data dependent, so a
// counted vmcnt is impossible).
__device__ __forceinline__ void touch_block(Block &b) {
#pragma unroll
    for (int k = 0; k < 4; k++)
        asm volatile("" : "+v"(b.q[k].x), "+v"(b.q[k].y), "+v"(b.q[k].z), "+v"(b.q[k].w));
    asm volatile("" : "+v"(b.wb));
}

__device__ __forceinline__ Pending compute_tile(const KernelArgs &a, Shared &sh, const uint32_t tile,
                                                const Block &blk, Block &prefetched,
                                                uint64_t &prefetched_pre,
                                                const uint32_t next_ticket_reg) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    uint64_t *agg = a.ws + kDescOffset;
    const uint64_t len = a.len;
    const uint64_t tile_start = (uint64_t)tile * kTileBytes;
    const uint64_t blk_off = tile_start + (uint64_t)tid * 64u;
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

    // ---- wave 0: carries into the tile from the 64 bytes in front of it
    const bool have_window = (tile > 0) || (a.flags & kFlagHasPrefix);
    if (wave == 0) {
        const uint32_t wb = blk.wb;
        uint32_t e_in, ps_in, u8_in;
        // utf8 carry word of the window's last bytes (lane_math.h layout)
        {
            const bool l234 = (wb >= 0xC0u) && (wb < 0xF8u);
            const bool l34 = (wb >= 0xE0u) && (wb < 0xF8u);
            const bool l4 = (wb >= 0xF0u) && (wb < 0xF8u);
            const uint64_t m234 = __ballot(l234), m34 = __ballot(l34), m4 = __ballot(l4);
            const uint64_t mE0 = __ballot(wb == 0xE0u), mED = __ballot(wb == 0xEDu);
            const uint64_t mF0 = __ballot(wb == 0xF0u), mF4 = __ballot(wb == 0xF4u);
            u8_in = (uint32_t)(m234 >> 63) | ((uint32_t)(m34 >> 62) << 1) |
                    ((uint32_t)(m4 >> 61) << 3) | ((uint32_t)(mE0 >> 63) << 6) |
                    ((uint32_t)(mED >> 63) << 7) | ((uint32_t)(mF0 >> 63) << 8) |
                    ((uint32_t)(mF4 >> 63) << 9);
            if (!have_window) u8_in = 0;
        }
        if (tile == 0) {
            // exact state at the first byte of this launch
            e_in = a.carry_in->next_is_escaped & 1u;
            ps_in = a.carry_in->prev_scalar & 1u;
        } else {
            const uint64_t WB = __ballot(wb == 0x5Cu);
            const uint64_t WQ = __ballot(wb == 0x22u);
            const bool nonscalar = (wb == 0x20u) | (wb == 0x09u) | (wb == 0x0Au) | (wb == 0x0Du) |
                                   (wb == 0x0Cu) | (wb == 0x1Au) | (wb == 0x2Cu) | (wb == 0x3Au) |
                                   (wb == 0x5Bu) | (wb == 0x5Du) | (wb == 0x7Bu) | (wb == 0x7Du);
            const uint64_t WNS = __ballot(nonscalar);
            const uint32_t r = top_run(WB);  // backslashes ending at byte[-1]
            bool resolved = (r != 64u);
            e_in = r & 1u;
            if (r >= 1u) {
                ps_in = 1u;  // byte[-1] is a backslash: a non-quote scalar
            } else if ((WNS >> 63) & 1u) {
                ps_in = 0u;
            } else if (!((WQ >> 63) & 1u)) {
                ps_in = 1u;
            } else {
                // byte[-1] is '"': a real quote unless escaped by an odd run before it.
                // (WB<<1)|1 has bit 0 forced: a result of 64 means bits 1..63 are all set.
                const uint32_t r2 = top_run((WB << 1) | 1ull);  // run ending at byte[-2]
                if (r2 == 64u) resolved = false;
                ps_in = r2 & 1u;
            }
            if (!resolved) {
                // >= 62 consecutive backslashes in front of the tile: take the exact
                // carries the predecessor publishes with its aggregate.
                uint32_t to = 0;
                const uint64_t d = wait_desc(&agg[tile - 1], &to);
                if (to && lane == 0) sh.timeout = 1;
                e_in = (uint32_t)(d >> 58) & 1u;
                ps_in = (uint32_t)(d >> 57) & 1u;
            }
        }
        if (lane == 0) {
            sh.tile_e_in = e_in;
            sh.tile_ps_in = ps_in;
            sh.tile_u8_in = u8_in;
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
    {
        const uint64_t s0 = add_a + add_b;
        const uint64_t s1 = s0 + 1ull;
        const uint32_t c0 = (uint32_t)(((add_a & add_b) | ((add_a | add_b) & ~s0)) >> 63);
        const uint32_t c1 = (uint32_t)(((add_a & add_b) | ((add_a | add_b) & ~s1)) >> 63);
        if (lane == 0) sh.esc[wave] = c0 | (c1 << 1);
    }
    MSJ_STAMP(tile, 3);
    __syncthreads();  // B1: tile carries + per-wave escape transfer published
    MSJ_STAMP(tile, 4);

    uint32_t wave_e_in = sh.tile_e_in;
    for (uint32_t w = 0; w < wave; w++) wave_e_in = (sh.esc[w] >> wave_e_in) & 1u;
    // escape carry out of the whole tile (kept in a register: sh.esc is rewritten by
    // the next tile before every wave has passed the last barrier of this one)
    uint32_t tile_e_out = wave_e_in;
    for (uint32_t w = wave; w < kWaves; w++) tile_e_out = (sh.esc[w] >> tile_e_out) & 1u;
    const uint64_t carries = (add_a + add_b + wave_e_in) ^ add_a ^ add_b;
    const uint32_t lane_e_in = (uint32_t)(carries >> lane) & 1u;

    // ---- strings (json_string_scanner.mojo:55-69) with the lane's exact escape carry
    uint32_t lane_e_out;
    const uint64_t escaped = escaped_mask(cls.backslash, lane_e_in, &lane_e_out);
    const uint64_t quote = cls.quote_chr & ~escaped;
    const uint64_t S0 = prefix_xor(quote);  // in_string if the lane started outside a string
    const uint64_t PM = __ballot((S0 >> 63) != 0);
    const uint32_t lane_par = lanes_below(PM) & 1u;  // parity of the lanes before me in the wave

    // ---- scalars (json_scanner.mojo:64-79)
    const uint64_t scalar = ~(cls.op | cls.ws);
    const uint64_t nqs = scalar & ~quote;
    const uint32_t my_ps = (uint32_t)(nqs >> 63);
    uint32_t prev_ps = __shfl_up(my_ps, 1);

    // ---- utf8 planes
    const bool do_utf8 = !(a.flags & kFlagNoUtf8);
    Utf8Planes u8p;
    uint32_t my_u8c = 0, prev_u8c = 0;
    if (do_utf8) {
        u8p = utf8_planes(p);
        my_u8c = utf8_carry_out(u8p);
        prev_u8c = __shfl_up(my_u8c, 1);
    }
    if (lane == 63) {
        sh.ps[wave] = my_ps;
        sh.u8c[wave] = my_u8c;
        sh.par[wave] = (uint32_t)__popcll(PM) & 1u;
    }
    __syncthreads();  // B2: wave parities / prev_scalar / utf8 carries published
    MSJ_STAMP(tile, 5);

    uint32_t wave_par = 0;
    for (uint32_t w = 0; w < wave; w++) wave_par ^= sh.par[w];
    if (lane == 0) {
        prev_ps = (wave == 0) ? sh.tile_ps_in : sh.ps[wave - 1];
        prev_u8c = (wave == 0) ? sh.tile_u8_in : sh.u8c[wave - 1];
    }
    const uint32_t tile_ps_out = sh.ps[kWaves - 1];
    const uint32_t tile_pend = (sh.u8c[kWaves - 1] & 0x3Fu) ? 1u : 0u;
    const uint64_t lane_in = (uint64_t)(-(int64_t)(lane_par ^ wave_par));  // all-ones: inside a string
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
    bool u8err = false;
    if (do_utf8) u8err = utf8_errors(p, u8p, prev_u8c) != 0;

    // ---- packed inclusive scan of the per-lane structural counts
    r.pk = (uint32_t)__popcll(r.T0) | ((uint32_t)__popcll(r.T1) << 16);
    uint32_t inc = r.pk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += t;
    }
    r.inc = inc;
    {
        const uint64_t me0 = __ballot(err0), me1 = __ballot(err1), mu8 = __ballot(u8err);
        if (lane == 63) {
            sh.cnt[wave] = inc;
            sh.flg[wave] = (me0 ? 1u : 0u) | (me1 ? 2u : 0u) | (mu8 ? 4u : 0u);
        }
    }
    if (tid == 0) {
        ticket_ready();  // requested before this tile was computed: arrived long ago
        sh.tk[2] = next_ticket_reg;
    }
    __syncthreads();  // B3: wave totals published
    MSJ_STAMP(tile, 6);

    uint32_t wave_off = 0, tile_cnt = 0, tile_flg = 0, tile_par = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; w++) {
        if (w < wave) wave_off += sh.cnt[w];
        tile_cnt += sh.cnt[w];
        tile_flg |= sh.flg[w];
        tile_par ^= sh.par[w];
    }
    r.wave_off = wave_off;
    r.tile_cnt = tile_cnt;
    r.tile = tile;
    touch_block(prefetched);  // before any store of this iteration is issued
    {
        uint32_t lo = (uint32_t)prefetched_pre, hi = (uint32_t)(prefetched_pre >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        prefetched_pre = ((uint64_t)hi << 32) | lo;
    }
    if (tid == 0) {
        st_desc(&agg[tile], kAgg | ((uint64_t)tile_par << 61) | ((uint64_t)(tile_flg & 1u) << 60) |
                                ((uint64_t)((tile_flg >> 1) & 1u) << 59) |
                                ((uint64_t)tile_e_out << 58) | ((uint64_t)tile_ps_out << 57) |
                                ((uint64_t)((tile_flg >> 2) & 1u) << 56) |
                                ((uint64_t)tile_pend << 55) | ((uint64_t)sh.timeout << 54) |
                                ((uint64_t)(tile_cnt >> 16) << 15) | (uint64_t)(tile_cnt & 0xFFFFu));
    }
    MSJ_STAMP(tile, 7);
    return r;
}

// ---- BitIndexer.write (json_structural_indexer.mojo:46-58) for one computed tile:
//      wait for the tile's prefix, stage the ascending offsets in LDS at their
//      tile-relative position, write them out as aligned 16-byte stores (one L2
//      request per 64 B instead of one per index).
__device__ __forceinline__ void emit_tile(const KernelArgs &a, Shared &sh, const Pending &r,
                                          const uint64_t pre_word, const uint64_t count0) {
    const uint32_t tid = threadIdx.x;
    const uint64_t *pre = a.ws + kDescOffset + a.ntiles;
    MSJ_STAMP(r.tile, 8);
    if (tid < 64u) {
        // pre_word was requested a whole compute phase ago; poll only if the
        // resolver had not published this tile's prefix yet at that time.
        uint32_t to = 0;
        uint64_t d = pre_word;
        if ((d >> 62) == 0ull) d = wait_desc(&pre[r.tile], &to);
        if (tid == 0) {
            sh.s_in = (uint32_t)(d >> 61) & 1u;
            sh.base = count0 + (uint64_t)(uint32_t)d;
            if (to || ((d >> 54) & 1u)) sh.timeout = 1;
        }
    }
    MSJ_STAMP(r.tile, 9);
    __syncthreads();  // B4: s_in / base known to every wave
    if ((a.flags & kFlagNoEmit) || sh.timeout) {
        __syncthreads();  // keep sh.s_in / sh.base stable until everyone has read them
        return;
    }
    const uint32_t s_in = sh.s_in;
    const uint64_t base = sh.base;
    const uint64_t T = s_in ? r.T1 : r.T0;
    const uint32_t excl = r.inc - r.pk;
    const uint32_t lane_off =
        s_in ? ((excl >> 16) + (r.wave_off >> 16)) : ((excl & 0xFFFFu) + (r.wave_off & 0xFFFFu));
    const uint32_t my_cnt = s_in ? (r.tile_cnt >> 16) : (r.tile_cnt & 0xFFFFu);
    const bool fits = base + my_cnt <= a.capacity;
    const uint32_t shift = (uint32_t)(base & 3u);  // stage[j] <-> idx[base - shift + j]
    const uint32_t vend = shift + my_cnt;
    const uint32_t v0 = (uint32_t)((uint64_t)r.tile * kTileBytes) + tid * 64u;
    uint32_t vpos = shift + lane_off;
    uint32_t tlo = (uint32_t)T, thi = (uint32_t)(T >> 32);
    for (uint32_t r0 = 0; r0 < vend; r0 += kStageWords) {
        const uint32_t r1 = r0 + kStageWords;
        while (tlo && vpos < r1) {
            sh.stage[vpos - r0] = v0 + (uint32_t)__builtin_ctz(tlo);
            tlo &= tlo - 1;
            vpos++;
        }
        if (!tlo) {
            while (thi && vpos < r1) {
                sh.stage[vpos - r0] = v0 + 32u + (uint32_t)__builtin_ctz(thi);
                thi &= thi - 1;
                vpos++;
            }
        }
        __syncthreads();
        const uint32_t lim = vend < r1 ? vend : r1;
        const uint64_t gbase = base - shift + r0;
        for (uint32_t q = tid; 4u * q < lim - r0; q += kThreads) {
            const uint32_t vq = r0 + 4u * q;
            const uint4 val = *reinterpret_cast<const uint4 *>(&sh.stage[4u * q]);
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
        __syncthreads();  // stage is reused by the next round / next tile
    }
    if (vend == 0) __syncthreads();  // same barrier count on the empty path as on the timeout path
    MSJ_STAMP(r.tile, 10);
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
    for (uint32_t c = wave; c < nchunks; c += kWaves) {
        const uint32_t first = c * kResolveChunk + lane * kResolveE;
        uint64_t d[kResolveE];
        uint32_t spins = 0, poisoned = 0;
        for (;;) {
            bool all_ready = true;
#pragma unroll
            for (int e = 0; e < kResolveE; e++) {
                d[e] = (first + e < ntiles) ? ld_desc(&agg[first + e]) : kAgg;  // past the end: identity
                all_ready = all_ready && ((d[e] >> 62) != 0ull);
            }
            if (__all(all_ready)) break;
            if (++spins > kSpinLimit) {
                poisoned = 1;  // give up: missing aggregates count as identity, result is flagged
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        // lane aggregate under both incoming states
        uint32_t s0 = 0, s1 = 1, c_0 = 0, c_1 = 0, e_0 = 0, e_1 = 0, lu = 0, lpoison = poisoned;
#pragma unroll
        for (int e = 0; e < kResolveE; e++) {
            const uint64_t de = ((d[e] >> 62) != 0ull) ? d[e] : kAgg;
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
        const uint64_t PM = __ballot((s0 & 1u) != 0u);  // lane parity
        const uint64_t UM = __ballot(lu != 0u);
        const uint64_t XM = __ballot(lpoison != 0u);
        // ---- take the running state from the owner of the previous chunk
        spins = 0;
        while (*seq != c) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kSpinLimit) break;  // cannot happen unless a sibling wave died
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t s = sh.rs_s, cnt = sh.rs_cnt, err = sh.rs_err, u8 = sh.rs_u8, poison = sh.rs_poison;
        const uint32_t in_l = s ^ ((uint32_t)__popcll(PM & below) & 1u);
        const uint32_t mycnt = in_l ? c_1 : c_0;
        uint32_t incl = mycnt;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t t = __shfl_up(incl, dd);
            if (lane >= (uint32_t)dd) incl += t;
        }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63);
        const uint64_t EM = __ballot((in_l ? e_1 : e_0) != 0u);
        const uint32_t s_new = s ^ ((uint32_t)__popcll(PM) & 1u);
        const uint32_t cnt_new = cnt + total;
        const uint32_t err_new = err | (EM ? 1u : 0u);
        const uint32_t u8_new = u8 | (UM ? 1u : 0u);
        const uint32_t poison_new = poison | (XM ? 1u : 0u);
        if (lane == 0) {
            sh.rs_s = s_new;
            sh.rs_cnt = cnt_new;
            sh.rs_err = err_new;
            sh.rs_u8 = u8_new;
            sh.rs_poison = poison_new;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            *seq = c + 1u;
        }
        // ---- walk my tiles again with the real state, publishing each tile's prefix
        uint32_t cs = in_l, cb = cnt + (incl - mycnt);
        uint32_t ce = err | ((EM & below) ? 1u : 0u);
        uint32_t cu = u8 | ((UM & below) ? 1u : 0u);
        const uint64_t pz = (uint64_t)poison_new << 54;
#pragma unroll
        for (int e = 0; e < kResolveE; e++) {
            if (first + e < ntiles) {
                st_desc(&pre[first + e], kPre | ((uint64_t)cs << 61) | ((uint64_t)ce << 60) |
                                             ((uint64_t)cu << 56) | pz | (uint64_t)cb);
                const uint64_t de = ((d[e] >> 62) != 0ull) ? d[e] : kAgg;
                const uint32_t p = (uint32_t)(de >> 61) & 1u;
                const uint32_t c0 = (uint32_t)de & 0x7FFFu, c1 = (uint32_t)(de >> 15) & 0xFFFFu;
                const uint32_t e0 = (uint32_t)(de >> 60) & 1u, e1 = (uint32_t)(de >> 59) & 1u;
                cb += cs ? c1 : c0;
                ce |= cs ? e1 : e0;
                cu |= (uint32_t)(de >> 56) & 1u;
                cs ^= p;
            }
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
    unsigned int *ticket_ctr = reinterpret_cast<unsigned int *>(a.ws);
    // ---- ordered tickets: ticket 0 is the resolver, ticket t+1 works on tile t.  A
    //      workgroup only holds tickets once it is running, and always handles its
    //      tickets in increasing order, so everything a waiter depends on (the
    //      aggregates of earlier tiles, the resolver) belongs to a running workgroup
    //      that is not waiting on anything later: no deadlock whatever the dispatch
    //      order or residency.
    if (tid == 0) {
        const uint32_t t0 = atomicAdd(ticket_ctr, 1u);
        sh.tk[0] = t0;
        sh.tk[1] = (t0 != 0u) ? atomicAdd(ticket_ctr, 1u) : 0u;
        sh.timeout = 0;
    }
    __syncthreads();
    uint32_t t_cur = sh.tk[0];
    if (t_cur == 0u) {
        resolver(a, sh);
        return;
    }
    uint32_t t_next = sh.tk[1];
    const uint32_t last_ticket = a.ntiles;  // ticket k works on tile k-1
    if (t_cur > last_ticket) return;

    Block cur;
    load_block(a, t_cur - 1u, cur);
    touch_block(cur);  // loop invariant: `cur` has arrived (no vmcnt wait on it inside the loop)
    // Two tiles stay pending: tile i-2 is emitted at the top of iteration i, so its
    // prefix has had two compute phases to arrive, and its index stores are a whole
    // compute phase old (i.e. complete) when the next wait on the load queue comes.
    Pending older, newer;
    uint32_t n_pending = 0;
    const uint64_t count0 = a.carry_in->count;  // launch invariant: read once
    const uint64_t *pre = a.ws + kDescOffset + a.ntiles;
    uint64_t pre_older = 0;
    while (t_cur <= last_ticket) {
        MSJ_STAMP(t_cur - 1u, 0);
        if (n_pending == 2u) {
            emit_tile(a, sh, older, pre_older, count0);
            older = newer;
            n_pending = 1u;
        }
        // request the ticket after next, the next tile's bytes and the prefix of the
        // tile that is emitted next, before computing
        const uint32_t t_nn_reg = ticket_request(ticket_ctr);
        Block nxt;  // past the last ticket: harmless re-read of the last tile (keeps this branch-free)
        load_block(a, (t_next <= last_ticket ? t_next : last_ticket) - 1u, nxt);
        pre_older = ld_desc(&pre[n_pending ? older.tile : t_cur - 1u]);
        MSJ_STAMP(t_cur - 1u, 11);
        const Pending now = compute_tile(a, sh, t_cur - 1u, cur, nxt, pre_older, t_nn_reg);
        const uint32_t t_nn = sh.tk[2];
        if (n_pending == 0u) {
            older = now;
            pre_older = 0;  // the word read above was this tile's own (not published yet)
        } else {
            newer = now;
        }
        n_pending++;
        cur = nxt;
        t_cur = t_next;
        t_next = t_nn;
    }
    if (n_pending >= 1u) emit_tile(a, sh, older, pre_older, count0);
    if (n_pending == 2u) emit_tile(a, sh, newer, 0ull, count0);
}

}  // namespace msj

extern "C" int msj_launch_stage1(const msj::KernelArgs *args, void *stream, uint32_t grid) {
    const msj::KernelArgs a = *args;
    // persistent workgroups: at most one per tile plus the resolver (ticket 0)
    const uint32_t g = (grid == 0 || grid > a.ntiles + 1u) ? a.ntiles + 1u : grid;
    hipLaunchKernelGGL(msj::stage1_kernel, dim3(g), dim3(msj::kThreads), 0,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int msj_stage1_occupancy(int *blocks_per_cu) {
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, msj::stage1_kernel,
                                                            msj::kThreads, 0);
}
