// stage1_kernel.hip -- single-pass stage-1 structural indexer for gfx950 (MI355X).
//
// Replaces the reference's serial block loop
//   JsonStructuralIndexer.index[128] / step / next / finish
//   (src/mojo_simdjson/generic/stage1/json_structural_indexer.mojo:81-186)
// by one kernel launch over the whole buffer:
//
//   * a workgroup (4 wave64 = 256 lanes) owns one 16 KiB tile; every lane owns
//     one 64-byte block = the unit of one JsonScanner.next call, and all masks
//     are uint64 with the reference's bit order (lane_math.h);
//   * the three 1-bit carries the reference threads through its loop
//     (next_is_escaped json_escape_scanner.mojo:13, prev_in_string
//     json_string_scanner.mojo:49, prev_scalar json_scanner.mojo:57) are
//     resolved lane -> wave -> workgroup with __ballot + a 64-bit
//     carry-lookahead add (escape), ballot/mbcnt prefix parity (in-string) and
//     a one-lane shuffle (prev_scalar);
//   * across tiles only the in-string bit and the running structural count
//     are chained, with a decoupled look-back over 64-bit tile descriptors
//     (one relaxed agent-scope 8-byte store / load per hop: the data is the
//     flag); the escape / prev_scalar carries into a tile are derived locally
//     from the 64 bytes in front of it;
//   * BitIndexer.write (json_structural_indexer.mojo:46-58) becomes a packed
//     (count|count<<16) wave scan and a per-lane ctz loop that writes the
//     ascending offsets straight to their final position.
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
// bits 63:62 status (0 = not ready, 1 = aggregate, 2 = inclusive)
// aggregate : 61 parity, 60 err(s_in=0), 59 err(s_in=1), 58 e_out, 57 ps_out,
//             56 utf8 err, 30:15 count(s_in=1), 14:0 count(s_in=0)
// inclusive : 61 in_string after tile, 60 unescaped err so far, 58 e_out,
//             57 ps_out, 56 utf8 err so far, 31:0 count so far (launch-relative)
constexpr uint64_t kAgg = 1ull << 62;
constexpr uint64_t kInc = 2ull << 62;

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

struct Shared {
    uint32_t tile;
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

__global__ __launch_bounds__(kThreads) void stage1_kernel(const KernelArgs a) {
    __shared__ Shared sh;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    // ---- ordered tile id: predecessors have all started (no look-back deadlock)
    if (tid == 0) {
        sh.tile = atomicAdd(reinterpret_cast<unsigned int *>(a.ws), 1u);
        sh.timeout = 0;
    }
    __syncthreads();
    const uint32_t tile = sh.tile;
    uint64_t *desc = a.ws + kDescOffset;
    const uint64_t len = a.len;
    const uint64_t tile_start = (uint64_t)tile * kTileBytes;
    const uint64_t blk_off = tile_start + (uint64_t)tid * 64u;

    // ---- load this lane's 64-byte block (4 x 16 B), bytes past the end masked
    uint32_t x[16];
    uint64_t valid;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.buf + blk_off);
        uint4 q[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (blk_off + 16u * k < len)
                q[k] = src[k];
            else
                q[k] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            x[4 * k + 0] = q[k].x;
            x[4 * k + 1] = q[k].y;
            x[4 * k + 2] = q[k].z;
            x[4 * k + 3] = q[k].w;
        }
        if (blk_off + 64u <= len)
            valid = ~0ull;
        else if (blk_off < len)
            valid = (1ull << (len - blk_off)) - 1ull;
        else
            valid = 0ull;
    }

    // ---- wave 0: carries into the tile from the 64 bytes in front of it
    const bool have_window = (tile > 0) || (a.flags & kFlagHasPrefix);
    if (wave == 0) {
        uint32_t wb = 0;
        if (have_window) wb = a.buf[(int64_t)tile_start - 64 + (int64_t)lane];
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
                // byte[-1] is '"': a real quote unless escaped by an odd run before it
                const uint32_t r2 = top_run((WB << 1) | 1ull) - 0u;  // run ending at byte[-2]
                // (WB<<1)|1 has bit0 forced: r2 == 64 means bits 1..63 all set
                if (r2 == 64u) resolved = false;
                ps_in = r2 & 1u;
            }
            if (!resolved) {
                // >= 62 consecutive backslashes in front of the tile: take the exact
                // carries the predecessor publishes with its descriptor.
                uint32_t to = 0;
                const uint64_t d = wait_desc(&desc[tile - 1], &to);
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
    __syncthreads();  // B1: tile carries + per-wave escape transfer published

    uint32_t wave_e_in = sh.tile_e_in;
    for (uint32_t w = 0; w < wave; w++) wave_e_in = (sh.esc[w] >> wave_e_in) & 1u;
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

    uint32_t wave_par = 0;
    for (uint32_t w = 0; w < wave; w++) wave_par ^= sh.par[w];
    if (lane == 0) {
        prev_ps = (wave == 0) ? sh.tile_ps_in : sh.ps[wave - 1];
        prev_u8c = (wave == 0) ? sh.tile_u8_in : sh.u8c[wave - 1];
    }
    const uint64_t lane_in = (uint64_t)(-(int64_t)(lane_par ^ wave_par));  // all-ones: inside a string
    // in_string / string_tail assuming the TILE starts outside a string
    const uint64_t in_string0 = S0 ^ lane_in;
    const uint64_t string_tail0 = in_string0 ^ quote;  // json_string_scanner.mojo:40-44
    const uint64_t follows = (nqs << 1) | prev_ps;     // json_scanner.mojo:76-79
    const uint64_t potential = cls.op | (scalar & ~follows);
    const uint64_t T0 = potential & ~string_tail0;  // structural_start if tile s_in = 0
    const uint64_t T1 = potential & string_tail0;   //                  if tile s_in = 1
    const bool err0 = (cls.ctrl & in_string0) != 0;   // json_structural_indexer.mojo:143-145
    const bool err1 = (cls.ctrl & ~in_string0) != 0;
    bool u8err = false;
    if (do_utf8) u8err = utf8_errors(p, u8p, prev_u8c) != 0;

    // ---- packed inclusive scan of the per-lane structural counts
    const uint32_t pk = (uint32_t)__popcll(T0) | ((uint32_t)__popcll(T1) << 16);
    uint32_t inc = pk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += t;
    }
    {
        const uint64_t me0 = __ballot(err0), me1 = __ballot(err1), mu8 = __ballot(u8err);
        if (lane == 63) {
            sh.cnt[wave] = inc;
            sh.flg[wave] = (me0 ? 1u : 0u) | (me1 ? 2u : 0u) | (mu8 ? 4u : 0u);
        }
    }
    __syncthreads();  // B3: wave totals published

    uint32_t wave_off = 0, tile_cnt = 0, tile_flg = 0, tile_par = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWaves; w++) {
        if (w < wave) wave_off += sh.cnt[w];
        tile_cnt += sh.cnt[w];
        tile_flg |= sh.flg[w];
        tile_par ^= sh.par[w];
    }
    const uint32_t tile_c0 = tile_cnt & 0xFFFFu, tile_c1 = tile_cnt >> 16;

    // ---- wave 0: publish aggregate, decoupled look-back, publish inclusive
    if (wave == 0) {
        uint32_t tile_e_out = sh.tile_e_in;
        for (uint32_t w = 0; w < kWaves; w++) tile_e_out = (sh.esc[w] >> tile_e_out) & 1u;
        const uint32_t tile_ps_out = sh.ps[kWaves - 1];
        const uint64_t common = ((uint64_t)tile_e_out << 58) | ((uint64_t)tile_ps_out << 57);
        uint32_t s_in, err_in = 0, u8_in = 0, timeout = 0;
        uint64_t base_rel = 0;
        if (tile == 0) {
            s_in = a.carry_in->in_string & 1u;
        } else {
            if (lane == 0) {
                st_desc(&desc[tile], kAgg | ((uint64_t)tile_par << 61) |
                                         ((uint64_t)(tile_flg & 1u) << 60) |
                                         ((uint64_t)((tile_flg >> 1) & 1u) << 59) | common |
                                         ((uint64_t)((tile_flg >> 2) & 1u) << 56) |
                                         ((uint64_t)tile_c1 << 15) | (uint64_t)tile_c0);
            }
            // accumulated aggregate of the tiles between the window and me
            uint32_t accP = 0, accE0 = 0, accE1 = 0, accU = 0;
            uint64_t accC0 = 0, accC1 = 0;
            int64_t j = (int64_t)tile - 1;
            const uint32_t virt_s = a.carry_in->in_string & 1u;
            for (;;) {
                const int64_t my = j - (int64_t)lane;
                uint64_t d = kInc | ((uint64_t)virt_s << 61);  // before the launch: carry_in
                uint64_t incm, rdym, need;
                uint32_t spins = 0;
                for (;;) {
                    if (my >= 0) d = ld_desc(&desc[my]);
                    const uint32_t st = (uint32_t)(d >> 62);
                    rdym = __ballot(st != 0u);
                    incm = __ballot(st == 2u);
                    // lanes nearer than the nearest inclusive must all be ready
                    need = incm ? ((incm & (0ull - incm)) - 1ull) : ~0ull;
                    if ((rdym & need) == need) break;
                    if (++spins > kSpinLimit) {
                        timeout = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (timeout) {
                    s_in = 0;
                    break;
                }
                const uint32_t f = incm ? (uint32_t)__builtin_ctzll(incm) : 64u;
                const bool is_agg = lane < f;
                const uint32_t pbit = is_agg ? (uint32_t)(d >> 61) & 1u : 0u;
                const uint64_t PW = __ballot(pbit != 0u);
                // parity of the aggregates farther than me (applied before me)
                const uint32_t far_par = (lane >= 63u) ? 0u : ((uint32_t)__popcll(PW >> (lane + 1u)) & 1u);
                const uint32_t c0 = (uint32_t)d & 0x7FFFu, c1 = (uint32_t)(d >> 15) & 0xFFFFu;
                const uint32_t e0 = (uint32_t)(d >> 60) & 1u, e1 = (uint32_t)(d >> 59) & 1u;
                const uint64_t U = __ballot((lane <= f) && ((d >> 56) & 1u));
                const uint32_t win_par = (uint32_t)__popcll(PW) & 1u;
                if (f < 64u) {
                    const uint64_t dinc = __shfl(d, (int)f);
                    const uint32_t sb = (uint32_t)(dinc >> 61) & 1u;
                    const uint32_t in_k = sb ^ far_par;
                    const uint32_t sum = wave_sum(is_agg ? (in_k ? c1 : c0) : 0u);
                    const uint64_t E = __ballot(is_agg && (in_k ? e1 : e0));
                    const uint32_t s1 = sb ^ win_par;  // state entering the accumulated part
                    s_in = s1 ^ accP;
                    base_rel = (uint64_t)(uint32_t)dinc + sum + (s1 ? accC1 : accC0);
                    err_in = ((uint32_t)(dinc >> 60) & 1u) | (E ? 1u : 0u) | (s1 ? accE1 : accE0);
                    u8_in = (U ? 1u : 0u) | accU;
                    break;
                }
                // no inclusive among these 64: fold them into the accumulated aggregate
                const uint32_t in0 = far_par, in1 = far_par ^ 1u;
                const uint32_t sum0 = wave_sum(in0 ? c1 : c0);
                const uint32_t sum1 = wave_sum(in1 ? c1 : c0);
                const uint64_t E0 = __ballot(in0 ? e1 : e0), E1 = __ballot(in1 ? e1 : e0);
                const uint64_t nC0 = sum0 + (win_par ? accC1 : accC0);
                const uint64_t nC1 = sum1 + (win_par ? accC0 : accC1);
                const uint32_t nE0 = (E0 ? 1u : 0u) | (win_par ? accE1 : accE0);
                const uint32_t nE1 = (E1 ? 1u : 0u) | (win_par ? accE0 : accE1);
                accC0 = nC0;
                accC1 = nC1;
                accE0 = nE0;
                accE1 = nE1;
                accP ^= win_par;
                accU |= (U ? 1u : 0u);
                j -= 64;
            }
        }
        const uint32_t my_cnt = s_in ? tile_c1 : tile_c0;
        const uint32_t s_out = s_in ^ tile_par;
        const uint32_t err_out = err_in | ((s_in ? (tile_flg >> 1) : tile_flg) & 1u);
        const uint32_t u8_out = u8_in | ((tile_flg >> 2) & 1u);
        const uint64_t cnt_out = base_rel + my_cnt;
        if (lane == 0) {
            st_desc(&desc[tile], kInc | ((uint64_t)s_out << 61) | ((uint64_t)err_out << 60) |
                                     common | ((uint64_t)u8_out << 56) | (cnt_out & 0xFFFFFFFFull));
            sh.s_in = s_in;
            sh.base = a.carry_in->count + base_rel;
            if (timeout) sh.timeout = 1;

            if (tile == a.ntiles - 1) {
                // ---- finish(): json_structural_indexer.mojo:147-186
                const msj_carry cin = *a.carry_in;
                msj_carry out;
                const uint64_t n = cin.count + cnt_out;
                out.count = n;
                out.bytes = cin.bytes + len;
                out.in_string = s_out;
                out.next_is_escaped = tile_e_out;
                out.prev_scalar = tile_ps_out;
                out.unescaped_error = (cin.unescaped_error | err_out) ? 1u : 0u;
                uint32_t u8e = cin.utf8_error | u8_out;
                // a multi-byte sequence cut exactly at the end of the last full tile
                if ((a.flags & kFlagFinal) && do_utf8 && (len % kTileBytes) == 0 &&
                    (sh.u8c[kWaves - 1] & 0x3Fu))
                    u8e = 1;
                out.utf8_error = u8e ? 1u : 0u;
                out.internal_error = cin.internal_error | sh.timeout;
                int32_t code = MSJ_SUCCESS;
                if (a.flags & kFlagFinal) {
                    if (out.internal_error) {
                        code = MSJ_UNEXPECTED_ERROR;
                    } else if (s_out) {
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
                    a.segment->byte_len = len;
                    a.segment->index_begin = cin.count;
                    a.segment->count = cnt_out;
                }
            }
        }
    }
    __syncthreads();  // B4: s_in / base known to every wave

    // ---- BitIndexer.write (json_structural_indexer.mojo:46-58): ascending offsets
    if (!(a.flags & kFlagNoEmit)) {
        const uint32_t s_in = sh.s_in;
        uint64_t T = s_in ? T1 : T0;
        const uint32_t excl = inc - pk;
        const uint32_t lane_off =
            s_in ? ((excl >> 16) + (wave_off >> 16)) : ((excl & 0xFFFFu) + (wave_off & 0xFFFFu));
        uint64_t pos = sh.base + lane_off;
        const uint32_t v0 = (uint32_t)blk_off;
        const uint32_t my_cnt = s_in ? tile_c1 : tile_c0;
        if (sh.base + my_cnt <= a.capacity) {
            while (T) {
                a.idx[pos++] = v0 + (uint32_t)__builtin_ctzll(T);
                T &= T - 1;
            }
        } else {
            while (T) {
                if (pos < a.capacity) a.idx[pos] = v0 + (uint32_t)__builtin_ctzll(T);
                pos++;
                T &= T - 1;
            }
        }
    }
}

}  // namespace msj

extern "C" int msj_launch_stage1(const msj::KernelArgs *args, void *stream) {
    const msj::KernelArgs a = *args;
    hipLaunchKernelGGL(msj::stage1_kernel, dim3(a.ntiles), dim3(msj::kThreads), 0,
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
