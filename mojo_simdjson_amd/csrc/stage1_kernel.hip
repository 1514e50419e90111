// stage1_kernel.hip -- single-pass stage-1 structural indexer for gfx950 (MI355X).
//
// Replaces the reference's serial block loop
//   JsonStructuralIndexer.index[128] / step / next / finish
//   (src/mojo_simdjson/generic/stage1/json_structural_indexer.mojo:81-186)
// by one kernel launch over the whole buffer:
//
//   * persistent workgroups of four worker waves; a workgroup draws RANGES of 8 consecutive
//     4 KiB tiles from sharded ticket counters, each wave takes two of the tiles.  Each lane
//     owns one 64-byte block = the unit of one JsonScanner.next call, and all masks are
//     uint64 with the reference's bit order (lane_math.h);
//   * the three 1-bit carries the reference threads through its loop
//     (next_is_escaped json_escape_scanner.mojo:13, prev_in_string
//     json_string_scanner.mojo:49, prev_scalar json_scanner.mojo:57) are
//     resolved lane -> wave with __ballot + a 64-bit carry-lookahead add
//     (escape), ballot/mbcnt prefix parity (in-string) and a DPP lane shift
//     (prev_scalar); the escape / prev_scalar / UTF-8 carries INTO a tile are
//     derived locally from the 64 bytes in front of it;
//   * across tiles only the in-string bit and the running structural count are
//     chained.  Each tile yields a 64-bit aggregate (parity, count and error
//     bit for both possible incoming in-string states); the four waves fold a range's
//     eight aggregates at one barrier and publish one range aggregate.  One workgroup
//     does not index anything: it is the RESOLVER, whose four waves fold the range
//     aggregates in order (chunks of 64*kResolveE ranges, pipelined across the waves, state
//     handed over through LDS) and publish every range's prefix.  Workers read one
//     word.  All words are relaxed agent-scope 8-byte stores/loads: the data is
//     the flag;
//   * BitIndexer.write (json_structural_indexer.mojo:46-58) becomes a packed
//     (count|count<<16) wave scan, a straight-line per-lane ctz chain into a per-wave LDS
//     staging slice at the index's tile-relative position, and aligned 16-byte stores;
//   * per range iteration (worker_wave): compute two tiles -> publish -> request the next
//     range's bytes -> emit the range computed two iterations ago (parked in LDS).  The
//     resolver retires ranges in order, so the loop keeps the time between drawing a range
//     and publishing it short and free of anything that can block.
//
// No MFMA (nothing here is a contraction); integer/bitwise work on u8 input,
// u64 masks, u32 output.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/msj_stage1.h"
#include "lane_math.h"
#include "stage1_kernel.h"

// Issue priority of a worker wave by phase (s_setprio; the resolver runs at 3).  The four waves of a SIMD are in
// different phases of their range iterations; left to the default arbitration, a wave in a latency-bound phase -- the
// dependent steps of the scatter chain, the LDS round trip of the copy-out, the fold and the hand-over behind the
// barrier -- queues for issue slots behind waves that are in the middle of 300 independent vector instructions of a
// compute phase.  Raising the priority of those phases shortens them without costing the compute phase anything it
// would notice: +3.5 % minified, +4.8 % UTF-8-heavy, +5.8 % pretty-printed in one session
// (profiles/r03/ab_wave_priority.txt; any level above the compute phase's does it, the levels differ by < 1 %);
// the second tile of the compute phase one level above the first: another +1.5 % / +1.8 % / +3.5 %.  The ladder
// 0, 1, 2, 3 follows a wave's progress through its iteration.
// EXPERIMENT (round 5, off by default): request the NEXT range's first tile at the start of the present range's second
// compute phase (its registers are free from there) and its second tile right before the barrier -- the ticket then has to
// be known one iteration earlier (drawn two ranges ahead).  What it is for: on sparse input the range loop is a latency
// chain (load -> compute -> barrier -> hand-over -> load), not issue-bound.
#ifndef MSJ_EARLY_A
#define MSJ_EARLY_A 0
#endif
#ifndef MSJ_TICKET_AT_TOP
#define MSJ_TICKET_AT_TOP 0  // EXPERIMENT (round 5, off): see worker_wave
#endif
#ifndef MSJ_PRIO_COORD
#define MSJ_PRIO_COORD 2    // behind the barrier: fold, publish, hand-over, issue of the next range's loads
#endif
#ifndef MSJ_PRIO_EMIT
#define MSJ_PRIO_EMIT 3     // staging chains and copy-out of the parked tiles
#endif
#ifndef MSJ_PRIO_COMPUTE
#define MSJ_PRIO_COMPUTE 0  // bit-planes, classification, masks, scans of the range's first tile ...
#endif
#ifndef MSJ_PRIO_COMPUTE2
#define MSJ_PRIO_COMPUTE2 1  // ... and of its second: the closer a wave is to the barrier its three siblings wait at, the sooner it issues
#endif

namespace msj {

// ---- tile descriptors -------------------------------------------------------
// One 64-bit word per tile in each of two arrays.
// bits 63:62 status: 0 = not ready
// agg[t] (written by the tile's wave), status 1:
//   61 quote parity, 60 err(s_in=0), 59 err(s_in=1), 58 e_out, 57 ps_out,
//   56 utf8 err, 55 utf8 sequence pending at tile end, 54 poisoned (timeout),
//   31:16 count(s_in=1), 15:0 count(s_in=0)
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
// Wave-wide inclusive add scan and one-lane shift on the DPP path (gfx9 row_shr /
// row_bcast / wave_shr): 6 resp. 1 VALU operations, no LDS round trip.
__device__ __forceinline__ uint32_t dpp_add_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31
    return v;
}
// lane l gets v of lane l-1; lane 0 gets `first`
__device__ __forceinline__ uint32_t dpp_shift_up1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xF, 0xF, false);  // wave_shr:1
}

// (x << K) of a 64-bit per-lane plane, with the top K bits of the PREVIOUS lane's plane shifted
// in at the bottom (lane 0: `first_low`, K bits): the planes of a tile form one 4096-bit
// sequence.  One DPP move and two v_alignbit_b32.
template <int K>
__device__ __forceinline__ uint64_t wave_shl_in(uint64_t x, uint32_t first_low) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t prev_hi = dpp_shift_up1(hi, first_low << (32 - K));
    return u64(__builtin_amdgcn_alignbit(lo, prev_hi, 32 - K), __builtin_amdgcn_alignbit(hi, lo, 32 - K));
}
// The same with zeros shifted into lane 0 (bound_ctrl: no register has to hold a value for it)
template <int K>
__device__ __forceinline__ uint64_t wave_shl_0(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t prev_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x138, 0xF, 0xF, true);  // wave_shr:1
    return u64(__builtin_amdgcn_alignbit(lo, prev_hi, 32 - K), __builtin_amdgcn_alignbit(hi, lo, 32 - K));
}

// (x >> 1) of a 64-bit per-lane plane, with bit 0 of the NEXT lane's plane shifted in at the top
// (the last lane: 0): the backward counterpart of wave_shl_in.  One DPP move, two v_alignbit_b32.
__device__ __forceinline__ uint64_t wave_shr1_in(uint64_t x) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t next_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x130, 0xF, 0xF, true);  // wave_shl:1, 0 into the last lane
    return u64(__builtin_amdgcn_alignbit(hi, lo, 1), __builtin_amdgcn_alignbit(next_lo, hi, 1));
}

// A value every lane holds identically (loaded from one address): move it to an SGPR so
// that everything derived from it is scalar code.
__device__ __forceinline__ uint32_t uniform32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    return ((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v);
}

#ifdef MSJ_STAMPS
// Diagnostic build only: phase timestamps per tile (never compiled into the product .so).
// -DMSJ_STAMPS=2: only the real-time stamps (one per range and a few per workgroup): light enough to
// leave the launch's timeline as it is.
#if MSJ_STAMPS == 2
#define MSJ_STAMP(t, k) do {} while (0)
#else
#define MSJ_STAMP(t, k)                                                                   \
    do {                                                                                  \
        if ((threadIdx.x & 63u) == 0 && a.stamps)                                         \
            a.stamps[(uint64_t)(t) * 16 + (k)] = __builtin_amdgcn_s_memtime();            \
    } while (0)
#endif
#define MSJ_RSTAMP(t, k, cond)                                                            \
    do {                                                                                  \
        if ((cond) && a.stamps)                                                           \
            a.stamps[(uint64_t)(t) * 16 + (k)] = __builtin_amdgcn_s_memrealtime();        \
    } while (0)
#elif defined(MSJ_MARKS)
// Analysis build only (make marks): section markers in the .s, for per-section instruction counts.
#define MSJ_STAMP(t, k) asm volatile("; MSJ_MARK " #k)
#define MSJ_RSTAMP(t, k, cond) do {} while (0)
#else
#define MSJ_STAMP(t, k) do {} while (0)
#define MSJ_RSTAMP(t, k, cond) do {} while (0)
#endif

struct Shared {
    uint32_t role;
    uint32_t shard;          // worker workgroups: the ticket shard they draw from
    uint32_t first_lo;       // ... and the base tile of their first range
    uint32_t second_lo;      // MSJ_EARLY_A: ... and of their second (two draws at start-up)
    // worker workgroups: wave 0 hands the next range to the other waves with ONE 8-byte LDS store:
    // low word = base tile of the next range, high word = the iteration it is for (r + 1)
    uint64_t handoff __attribute__((aligned(8)));
    uint64_t tagg[2][kRange]; // worker workgroups: the range's tile aggregates, in tile order
    // resolver hand-off between its waves
    uint32_t rs_seq, rs_s, rs_cnt, rs_err, rs_u8, rs_poison;
    // per-wave index staging for coalesced stores
    uint32_t stage[kWaves][kStageSlack + kStageWords + kStageSlack] __attribute__((aligned(16)));
    // computed-but-not-yet-emitted tiles (two ranges deep), per wave and slot:
    // per lane T0|T1 masks and the packed exclusive count scan; per slot a few words
    uint4 pend_masks[kWaves][kPendSlots][64];
    uint32_t pend_excl[kWaves][kPendSlots][64];
    uint32_t pend_meta[kWaves][kPendSlots][4] __attribute__((aligned(16)));  // tile (~0 = empty), tile_cnt, in_cnt, in_s
};

// The state at the launch's first byte: *carry_in, or -- kFlagCarryByValue -- three bits of the kernel arguments
// (everything else zero).  All scalar loads / scalar code.
__device__ __forceinline__ uint64_t cin_count(const KernelArgs &a) {
    return (a.flags & kFlagCarryByValue) ? 0ull : a.carry_in->count;
}
// next_is_escaped | prev_scalar << 1
__device__ __forceinline__ uint32_t cin_carry0(const KernelArgs &a) {
    return (a.flags & kFlagCarryByValue) ? (a.carry_bits >> 1) & 3u
                                         : (a.carry_in->next_is_escaped & 1u) | ((a.carry_in->prev_scalar & 1u) << 1);
}
__device__ __forceinline__ uint32_t cin_in_string(const KernelArgs &a) {
    return (a.flags & kFlagCarryByValue) ? a.carry_bits & 1u : a.carry_in->in_string & 1u;
}
__device__ __forceinline__ uint32_t cin_internal_error(const KernelArgs &a) {
    return (a.flags & kFlagCarryByValue) ? 0u : a.carry_in->internal_error & 1u;
}
__device__ __forceinline__ msj_carry cin_all(const KernelArgs &a) {
    if (!(a.flags & kFlagCarryByValue)) return *a.carry_in;
    msj_carry c;
    c.count = c.bytes = 0;
    c.in_string = a.carry_bits & 1u;
    c.next_is_escaped = (a.carry_bits >> 1) & 1u;
    c.prev_scalar = (a.carry_bits >> 2) & 1u;
    c.unescaped_error = c.utf8_error = c.internal_error = c.capacity_error = 0;
    c.code = 0;
    for (int k = 0; k < 4; k++) c.reserved[k] = 0;
    return c;
}

// Waits are bounded by WALL TIME (s_memrealtime: a constant 100 MHz counter), not by a number
// of polls: a.wait_ticks (default 2 s; tests lower it to force the expiry path).  The clock is read
// once per 16 polls, for the first time after 16 (a short wait never reads it).  Expiry poisons
// the launch (internal_error); the host then re-runs the segment through the two-pass kernels,
// which wait for nothing (api.cpp).
struct WaitClock {
    uint32_t spins = 0, t0 = 0;
    __device__ __forceinline__ bool expired(const uint32_t wait_ticks) {
        if ((++spins & 15u) != 0u) return false;
        const uint32_t now = (uint32_t)__builtin_amdgcn_s_memrealtime();  // 32 bits of 10 ns ticks: 42 s
        if (spins == 16u) {
            t0 = now;
            return false;
        }
        return now - t0 > wait_ticks;
    }
    __device__ __forceinline__ void restart() { spins = 0; }
};

// Bounded poll of one descriptor until its status is non-zero.
__device__ __forceinline__ uint64_t wait_desc(const uint64_t *p, const uint32_t wait_ticks, uint32_t *timeout) {
    // one address for the whole wave: the value is uniform (scalar loop control)
    uint64_t d = uniform64(ld_desc(p));
    WaitClock clk;
    while ((d >> 62) == 0ull) {
        __builtin_amdgcn_s_sleep(2);
        d = uniform64(ld_desc(p));
        if (clk.expired(wait_ticks)) {
            *timeout = 1;
            break;
        }
    }
    return d;
}

// LDS written by some lanes of the wave, read by others: in order for one wave, the compiler only has to keep it so
__device__ __forceinline__ void lds_wave_sync_early() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The 64 bytes one lane owns, as loaded (4 x 16 B), plus one byte of the 64-byte
// window in front of the tile (lane l holds byte [tile_start - 64 + l]).
struct Block {
    uint4 q[4];
    uint32_t wb;
};

// Branch-free (so the compiler can leave all five loads in flight) and free of vector address
// arithmetic beyond one clamp per piece: the tile's base is scalar, the lane's piece offsets
// are loop invariants, and pieces past the end are redirected to the last 16-B piece that starts
// inside the input -- whatever they return is masked by `valid` in compute_tile.  For a tile inside
// the input the clamp is a no-op.
//
// Two shapes:
//   kCoalesced = false  lane_off[k] = 64 * lane + 16 * k: the lane gets its own 64-byte block, every wave
//       instruction touches 64 separate blocks and the other three quarters of each line come from L1 (four
//       passes of the texture path over the same 32 lines).  The two-pass kernels' shape.
//   kCoalesced = true   lane_off[k] = 16 * lane + 1024 * k: every wave instruction is 1 KiB contiguous; piece k of
//       lane l is chunk s = 64 k + l of the tile = quarter s & 3 of block s >> 2, and chunks_to_block() hands every
//       lane its own block through LDS.  The single-pass kernel's shape: measured against the block shape in one
//       session (1 GiB, settled clocks, profiles/r03/load_shape_ab.txt) sparse input 0.221 -> 0.210 ms, pretty-printed
//       0.286 -> 0.276, minified 0.354 -> 0.346, UTF-8-heavy 0.317 -> 0.308.  A bare read loop shows no difference
//       between the two shapes (6.2 TB/s both, scripts/ubench/read_shape.hip): what the kernel gains is the texture
//       path's time, which its stores and descriptor traffic share.
//   kNt (coalesced shape only): NON-TEMPORAL loads.  A bare loop reads 6.9 TB/s with them against 6.2.  Round 3 found
//       no gain in the kernel and 4-6 % lost on the dense extremes; with the wave priorities in place (round 4, same
//       box, alternating: profiles/r04/ab_nt_loads.txt) they are worth +2.5 .. +3.5 % on the minified workload, +2 .. +3 %
//       on pretty-printed input, +1 % on UTF-8-heavy (instruction-bound), and still COST the dense extremes 3 - 6 %
//       (d >= 0.5: the index stream is twice the input and more).  So the policy
//       follows the data: a wave requests its next range's bytes non-temporally unless the range it has just computed
//       was dense (worker_wave: more than kNtMaxIndices structurals in its two tiles) or the launch is short (kNtMinTiles).
template <bool kCoalesced, bool kNt = false>
__device__ __forceinline__ void load_block(const KernelArgs &a, const uint32_t tile, const uint32_t (&lane_off)[4],
                                           const uint32_t lane, Block &b) {
    // one launch covers < 2^32 bytes (kSegmentBytes), so offsets fit 32 bits
    const uint32_t len32 = (uint32_t)a.len;
    const uint32_t tile_off = tile * kTileBytes;                // uniform; tile < ntiles, so tile_off < len
    const uint8_t *base = a.buf + tile_off;                     // uniform
    const uint32_t lim = ((len32 - 1u) & ~15u) - tile_off;      // last piece that starts inside the input (buf is 16-B aligned)
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t off = lane_off[k] < lim ? lane_off[k] : lim;
        if (kCoalesced) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 *src = reinterpret_cast<const u32x4 *>(base + off);
            const u32x4 v = kNt ? __builtin_nontemporal_load(src) : *src;
            b.q[k] = make_uint4(v.x, v.y, v.z, v.w);
        } else {
            b.q[k] = *reinterpret_cast<const uint4 *>(base + off);
        }
    }
    // the 64 bytes in front of the tile; the launch's first tile has them only with kFlagHasPrefix
    // (without: any readable bytes will do, compute_tile ignores them)
    const bool have_window = (tile > 0) || (a.flags & kFlagHasPrefix);
    const uint8_t *wbase = have_window ? base - 64 : a.buf;
    const uint32_t wlim = have_window ? 63u : len32 - 1u;
    b.wb = wbase[lane < wlim ? lane : wlim];
}

// From the coalesced shape to "a lane owns a block": the wave's staging slice (4 KiB, free between two emissions)
// takes the chunks and gives the blocks back.  Lane l's piece k is chunk s = 64 k + l = quarter l & 3 of block
// bb = 16 k + (l >> 2); it goes to byte 64 bb + 16 ((l & 3) ^ ((bb >> 1) & 3)) -- (bb >> 1) & 3 = (l >> 3) & 3 whatever k
// is, so one lane-invariant address and three immediate offsets -- and lane L reads quarter c of its block at
// 64 L + 16 (c ^ ((L >> 1) & 3)).  The XOR keeps the eight lanes of a ds_read_b128 group on different banks (their
// blocks lie 64 bytes apart: without it four of them would share a bank); the writes are permutations inside
// 64-byte groups.  Eight LDS instructions per tile, no vector arithmetic beyond three XORs.
struct ChunkAddr {
    uint32_t wr, rd, sw16;  // LDS byte addresses: this lane's chunk slot for piece 0; this lane's block; 16 x its swizzle
};
__device__ __forceinline__ ChunkAddr chunk_addresses(const uint32_t *stage, const uint32_t lane) {
    const uint32_t s0 = (uint32_t)(uintptr_t)stage;
    ChunkAddr c;
    c.wr = s0 + 64u * (lane >> 2) + 16u * ((lane & 3u) ^ ((lane >> 3) & 3u));
    c.rd = s0 + 64u * lane;
    c.sw16 = 16u * ((lane >> 1) & 3u);
    return c;
}
__device__ __forceinline__ void chunks_to_block(Block &b, const ChunkAddr &ca) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 *lds_u32x4_ptr;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32x4 v = {b.q[k].x, b.q[k].y, b.q[k].z, b.q[k].w};
        *(lds_u32x4_ptr)(uintptr_t)(ca.wr + 1024u * k) = v;
    }
    lds_wave_sync_early();
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const u32x4 v = *(lds_u32x4_ptr)(uintptr_t)(ca.rd + (ca.sw16 ^ (16u * c)));
        b.q[c] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

// The kBatch tiles of this wave in the range that starts at tile `lo`.
template <bool kNt>
__device__ __forceinline__ void load_range(const KernelArgs &a, const uint32_t lo, const uint32_t wave,
                                           const uint32_t (&lane_off)[4], const uint32_t lane, Block (&blk)[kBatch]) {
    const uint32_t ntiles = a.ntiles;
#pragma unroll
    for (uint32_t j = 0; j < kBatch; j++) {
        const uint32_t t = lo + kWaves * j + wave;
        load_block<true, kNt>(a, t < ntiles ? t : ntiles - 1u, lane_off, lane, blk[j]);  // past the end: harmless re-read
    }
}
// structurals in a wave's two tiles of a range (either in-string state) from which on the NEXT range's bytes are
// requested with plain loads: 0.37 of the bytes (`[1234,` = 0.40 is neutral to -1 % non-temporally, `[123,` = 0.50 loses 3 %)
constexpr uint32_t kNtMaxIndices = 3000;
// ... and launches shorter than this many tiles (0.375 GiB) use plain loads throughout: a short launch is mostly start-up
// and tail, where a non-temporal load's bytes arrive later -- distinct 1/8 GiB inputs in turn (nothing to find in a
// cache): 0.0507 ms plain, 0.0551 non-temporal; 1/4 GiB 0.0886 / 0.0912; 1/2 GiB 0.1673 / 0.1652; 1 GiB 0.3216 / 0.3126
// (profiles/r04/small_launch_variants.txt)
constexpr uint32_t kNtMinTiles = 98304;

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
// range of a shard's k-th draw (shard 0's draw 0 is the resolver's: no range)
__device__ __forceinline__ uint32_t ticket_range(uint32_t k, uint32_t shard, uint32_t shards) {
    return (k - (shard == 0u ? 1u : 0u)) * shards + shard;
}

// BitIndexer.write_index (json_structural_indexer.mojo:39-44) for one 32-bit half mask:
// writes (value_base + bit position) of every set bit, ascending, to consecutive LDS slots
// [lds_byte_addr, lds_byte_addr + 4 * popcount(t)).
// Straight-line and from BOTH ends: a step takes the lowest and the highest set bit (v_ffbl /
// v_ffbh), writes them to the lane's next slot from the front and from the back (immediate
// offsets), and clears both with one three-input operation -- half the steps of a one-ended
// chain, eight vector instructions per two indices.  A mask with one bit left writes it twice
// (same slot, same value).  Lanes drop out through EXEC as their mask runs empty (v_cmpx), the
// wave leaves as soon as no lane is left.  (The compiler's version of such a loop spends ~8 scalar
// instructions per step on exec-mask bookkeeping.)
// The back slots are addressed as (addr_back + 124 - 4k): lds_byte_addr >= 128 is required (the
// staging slices lie behind the first 128 bytes of LDS, static_assert below).
__device__ __forceinline__ void scatter_bits32(uint32_t t, uint32_t lds_byte_addr, uint32_t nbits, uint32_t value_base) {
    uint32_t lo, hi, tm1;
    uint64_t save;
    const uint32_t back = lds_byte_addr + 4u * nbits - 128u;  // slot (nbits - 1 - k) = back + 124 - 4k
    const uint32_t vb31 = value_base + 31u;
    const uint32_t topbit = 0x80000000u;
    asm volatile(
        "s_mov_b64 %[save], exec\n"
        ".set msj_sb_k, 0\n"
        ".rept 16\n"
        "v_cmpx_ne_u32_e32 vcc, 0, %[t]\n"
        "s_cbranch_execz 1f\n"
        "v_ffbl_b32_e32 %[lo], %[t]\n"
        "v_ffbh_u32_e32 %[hi], %[t]\n"
        "v_add_u32_e32 %[tm1], -1, %[t]\n"
        "v_or_b32_e32 %[lo], %[lo], %[vb]\n"
        "ds_write_b32 %[front], %[lo] offset:4*msj_sb_k\n"
        "v_sub_u32_e32 %[lo], %[vb31], %[hi]\n"
        "v_lshrrev_b32_e32 %[hi], %[hi], %[top]\n"
        "ds_write_b32 %[back], %[lo] offset:124-4*msj_sb_k\n"
        "v_bitop3_b32 %[t], %[t], %[tm1], %[hi] bitop3:0x40\n"
        ".set msj_sb_k, msj_sb_k+1\n"
        ".endr\n"
        "1:\n"
        "s_mov_b64 exec, %[save]\n"
        : [t] "+v"(t), [lo] "=&v"(lo), [hi] "=&v"(hi), [tm1] "=&v"(tm1), [save] "=&s"(save)
        : [front] "v"(lds_byte_addr), [back] "v"(back), [vb] "v"(value_base), [vb31] "v"(vb31), [top] "v"(topbit)
        : "vcc", "memory");
}

static_assert(offsetof(Shared, stage) >= 128, "scatter_bits32 addresses the back slots from 128 bytes below the lane's end slot");

// What a computed tile keeps in registers until it is parked.
struct Pending {
    uint64_t T0, T1;     // structural_start masks for tile s_in = 0 / 1
    uint32_t excl;       // packed exclusive wave scan of the per-lane counts
    uint32_t tile_cnt;   // packed tile totals (uniform)
};

// Carries into a tile from the 64 bytes in front of it: wave-uniform, and apart from one compare
// (the window's backslash mask) and three lane reads, branch-free scalar code that the scheduler can
// run beside the bit-plane transpose.  The escape scanner (json_escape_scanner.mojo:18-45) on the
// window's backslash mask with carry-in 0 is exact unless the run of backslashes in front of the
// tile reaches back to the window's first byte (>= 63 bytes): `resolved` is false then and the tile
// takes the exact carries its predecessor publishes with its aggregate.
struct TileCarry {
    uint32_t e_in, ps_in, u8_in;
    bool resolved;
};
__device__ __forceinline__ TileCarry window_carries(const KernelArgs &a, const uint32_t tile, const uint32_t wb,
                                                    const uint32_t carry0) {
    TileCarry c;
    const bool have_window = (tile > 0) || (a.flags & kFlagHasPrefix);
    const uint32_t b1 = bcast(wb, 63), b2 = bcast(wb, 62), b3 = bcast(wb, 61);  // byte[-1], [-2], [-3]
    c.u8_in = 0;
    if (have_window && ((b1 | b2 | b3) & 0x80u)) {
        // utf8 carry word of the window's last bytes (lane_math.h layout)
        const uint32_t l1 = (b1 >= 0xC0u && b1 < 0xF8u), l1_34 = (b1 >= 0xE0u && b1 < 0xF8u);
        const uint32_t l1_4 = (b1 >= 0xF0u && b1 < 0xF8u);
        const uint32_t l2_34 = (b2 >= 0xE0u && b2 < 0xF8u), l2_4 = (b2 >= 0xF0u && b2 < 0xF8u);
        const uint32_t l3_4 = (b3 >= 0xF0u && b3 < 0xF8u);
        c.u8_in = l1 | (l2_34 << 1) | (l1_34 << 2) | (l3_4 << 3) | (l2_4 << 4) | (l1_4 << 5) |
                  ((uint32_t)(b1 == 0xE0u) << 6) | ((uint32_t)(b1 == 0xEDu) << 7) |
                  ((uint32_t)(b1 == 0xF0u) << 8) | ((uint32_t)(b1 == 0xF4u) << 9);
    }
    const uint64_t WB = __ballot(wb == 0x5Cu);
    constexpr uint64_t ODD = 0xAAAAAAAAAAAAAAAAull;
    const uint64_t t = (((WB << 1) | ODD) - WB) ^ ODD;      // :39-45 with next_is_escaped = 0
    const uint32_t esc1 = (uint32_t)((t ^ WB) >> 63);       // byte[-1] is escaped
    // op | ws as a 128-entry bit table (haswell.mojo:22-74):
    // {09,0A,0C,0D,1A,20,2C,3A} in the low word, {5B,5D,7B,7D} in the high word
    constexpr uint64_t kNsLo = (1ull << 0x09) | (1ull << 0x0A) | (1ull << 0x0C) | (1ull << 0x0D) |
                               (1ull << 0x1A) | (1ull << 0x20) | (1ull << 0x2C) | (1ull << 0x3A);
    constexpr uint64_t kNsHi = (1ull << (0x5B - 64)) | (1ull << (0x5D - 64)) |
                               (1ull << (0x7B - 64)) | (1ull << (0x7D - 64));
    const uint64_t tab = (b1 & 0x40u) ? kNsHi : kNsLo;
    const uint32_t nonscalar = (uint32_t)(tab >> (b1 & 63u)) & (uint32_t)(b1 < 0x80u);
    // byte[-1] an unescaped backslash: the tile's first byte is escaped
    c.e_in = (uint32_t)((t & WB) >> 63);
    // prev_scalar: byte[-1] is a scalar character (a backslash is one) and not a real quote
    c.ps_in = ~nonscalar & 1u & ((b1 == 0x22u) ? esc1 : 1u);
    // the run in front of the tile reaches the window's first byte: not decided by the window
    c.resolved = (WB | (1ull << 63)) != ~0ull;
    if (tile == 0) {
        // exact state at the first byte of this launch
        c.e_in = carry0 & 1u;
        c.ps_in = (carry0 >> 1) & 1u;
        c.resolved = true;
    }
    return c;
}

// ---- one tile: masks, carries, counts; returns the tile aggregate in agg_word ------------
// Written for latency as much as for instruction count: the wave-level decisions (a block starts
// escaped? any byte >= 0x80?) are taken from ballots issued long before the branch that needs
// them, and there is ONE uniform branch for the blocks that start escaped instead of one per special case.
__device__ __forceinline__ Pending compute_tile(const KernelArgs &a, const uint32_t tile,
                                                const uint32_t lane, const Block &blk, const uint32_t carry0,
                                                uint32_t &timeout, uint64_t &agg_word, const uint32_t exact = 0u) {
    const uint32_t len = (uint32_t)a.len;  // < 2^32 per launch
    MSJ_STAMP(tile, 1);
    uint32_t x[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        x[4 * k + 0] = blk.q[k].x;
        x[4 * k + 1] = blk.q[k].y;
        x[4 * k + 2] = blk.q[k].z;
        x[4 * k + 3] = blk.q[k].w;
    }
    TileCarry tc = window_carries(a, tile, blk.wb, carry0);
    if (exact & 1u) {
        // two-pass path: the escape / prev_scalar carries of a tile behind >= 63 backslashes come from the
        // scan pass (bit 1, bit 2), never from waiting for the neighbour
        tc.e_in = (exact >> 1) & 1u;
        tc.ps_in = (exact >> 2) & 1u;
        tc.resolved = true;
    }
    // a window of a document stream starts at a document, its 16-byte aligned base up to 15 bytes
    // earlier: those bytes (the end of the previous document) read as blanks
    const uint32_t skip = tile == 0 ? (a.flags >> kFlagSkipShift) & 15u : 0u;  // uniform
    const bool partial = tile * kTileBytes + kTileBytes > len;  // uniform: the launch's last tile, cut short

#ifdef MSJ_UNIFORM_TILES
    // MEASURED NO-GAIN, not in the product (round 5, profiles/r05/ab_uniform_tiles.txt; build with -DMSJ_UNIFORM_TILES to
    // repeat): with this path a tile of blanks / of one scalar character runs ~15 vector instructions instead of 311, and
    // the d ~ 0 rows of the density sweep do not move (0.1790 / 0.1791 ms per GiB without it, 0.1714 / 0.1836 with it, same
    // box, alternating) -- those rows are bound by the range loop's latency chain (one 8 KiB request per wave and
    // iteration: load -> barrier -> hand-over -> load), not by instruction issue -- while the filter's 7 instructions
    // cost every ordinary tile ~1 % (pretty-printed 0.2510 -> 0.2548 ms, UTF-8-heavy 0.2897 -> 0.2942).
    // ---- a tile whose 4 096 bytes are ONE byte value that is a blank or a plain scalar character (padding between
    //      records, the inside of a run of one character): nothing to transpose or classify.  Round 5 (VERDICT round 4,
    //      "311 vector instructions are paid to learn that a tile of blanks holds nothing"): a three-dword filter that
    //      ordinary text fails at once (7 vector instructions per tile), then the other thirteen dwords.  What
    //      JsonScanner.next (json_scanner.mojo:64-70) computes for such a block is known in closed form:
    //        blanks        no structural, no scalar carry out                       (whitespace: haswell.mojo:23-65)
    //        scalar bytes  one structural -- the tile's first byte, unless the byte in front of the tile is a scalar
    //                      (follows, json_scanner.mojo:76-79) and only outside a string -- and the scalar carry out set.
    //      Neither holds a quote or a backslash, so parity, escape carry and both error flags are zero whatever the
    //      carries in (an escaped first byte that is no quote and no backslash changes nothing).
    if (!partial && !skip && tc.resolved && tc.u8_in == 0u) {  // uniform
        const uint32_t splat = perm(x[0], x[0], 0u);  // byte 0 of the lane's block, four times
        const uint32_t first = uniform32(splat);
        uint32_t dif = lut3<MSJ_TT(TA | (TB ^ TC))>(x[0] ^ splat, x[7], splat);
        dif = lut3<MSJ_TT(TA | (TB ^ TC))>(dif, x[15], splat);
        if (__ballot((dif | (splat ^ first)) != 0u) == 0ull) {  // uniform: the filter passed in every lane
            const uint32_t c = first & 0xFFu;
            // op | ws | quote | backslash | control | >= 0x80 as a 128-entry bit table (haswell.mojo:22-74 + 22, 5C, < 20)
            constexpr uint64_t kSpLo = 0xFFFFFFFFull | (1ull << 0x20) | (1ull << 0x22) | (1ull << 0x2C) | (1ull << 0x3A);
            constexpr uint64_t kSpHi = (1ull << (0x5B - 64)) | (1ull << (0x5C - 64)) | (1ull << (0x5D - 64)) |
                                       (1ull << (0x7B - 64)) | (1ull << (0x7D - 64));
            const bool plain = c < 0x80u && !((((c & 0x40u) ? kSpHi : kSpLo) >> (c & 63u)) & 1ull);
            if (c == 0x20u || plain) {  // uniform
                uint32_t acc = 0;
#pragma unroll
                for (int k = 1; k < 15; k++)
                    if (k != 7) acc = lut3<MSJ_TT(TA | (TB ^ TC))>(acc, x[k], splat);
                if (__ballot(acc != 0u) == 0ull) {  // uniform: all 4 096 bytes are c
                    Pending r;
                    const uint32_t one = (plain && tc.ps_in == 0u) ? 1u : 0u;  // the tile's first byte starts a scalar
                    r.T0 = (lane == 0u) ? (uint64_t)one : 0ull;
                    r.T1 = 0ull;
                    r.excl = (lane == 0u) ? 0u : one;  // packed (count if outside | count if inside << 16) in front of the lane
                    r.tile_cnt = one;
                    const uint32_t hi = (uint32_t)(kAgg >> 32) | ((plain ? 1u : 0u) << 25) | (timeout << 22);
                    agg_word = u64(r.tile_cnt, hi);
                    return r;
                }
            }
        }
    }
#endif
    MSJ_STAMP(tile, 2);
    // ---- bit-planes and character classes (lane_math.h)
    uint64_t p[8];
    bitplanes(x, p);
    Classes cls;
    uint64_t valid = ~0ull;
    if (!partial && !skip) {  // uniform: every byte of the tile is input
        cls = classify(p, ~0ull);
    } else {
        // bytes past the end behave as the reference's 0x20 padding (:103-107)
        const uint32_t blk_off = tile * kTileBytes + lane * 64u;
        if (blk_off >= len)
            valid = 0ull;
        else if (len - blk_off < 64u)
            valid = (1ull << (len - blk_off)) - 1ull;
        if (lane == 0) valid &= ~0ull << skip;
#pragma unroll
        for (int k = 0; k < 8; k++) p[k] &= valid;
        cls = classify(p, valid);
    }
    // any byte >= 0x80 in the tile (decides the UTF-8 pass much later)
    const uint64_t nonascii = __ballot(((uint32_t)p[7] | (uint32_t)(p[7] >> 32)) != 0u);

    // ---- escapes (json_escape_scanner.mojo:18-45).  Per lane for carry-in 0: tt, and the carry the
    //      block would hand on (g).  Wave level: ballot g and "all 64 bytes are backslashes" (the
    //      carry propagates); the carries of the 64-bit addition (G|P) + G + carry_in are exactly
    //      the per-lane carries (carry-lookahead).  A block that starts escaped is the exception (the byte
    //      before it ends an odd run of backslashes): quotes and strings are computed for carry-in 0
    //      first, and only a tile that has such a block redoes them with the lanes' carries.
    // A tile without a quote or a backslash (the inside of a long string, a run of blanks, a column of numbers)
    // has no escape or string work at all: its in-string state is its predecessor's, for every byte.
    const uint64_t qb = cls.quote_chr | cls.backslash;
    const bool quiet = __ballot(((uint32_t)qb | (uint32_t)(qb >> 32)) != 0u) == 0ull && !partial && tc.resolved;  // uniform
    uint64_t quote = 0, in_string0 = 0, escaped = 0;  // escaped: only needed (and only right) where the scanner is redone
    uint32_t tile_e_out = 0, tile_par = 0;
    if (!quiet) {
        const uint64_t tt = escape_tt0(cls.backslash);
        const uint64_t G = __ballot(escape_out0(tt, cls.backslash) != 0u);
        const uint64_t Pm = __ballot(((uint32_t)cls.backslash & (uint32_t)(cls.backslash >> 32)) == 0xFFFFFFFFu);
        quote = unescaped_quotes0(cls.quote_chr, tt);  // eq['"'] & ~escaped  (json_string_scanner.mojo:58)
        const uint64_t add_a = G | Pm, add_b = G;
        uint64_t add_s = add_a + add_b + tc.e_in;
        uint64_t carries = add_s ^ add_a ^ add_b;
        MSJ_STAMP(tile, 3);
        if (carries != 0ull || partial || !tc.resolved) {
            // Uniform, and NOT rare: a block starts escaped when the byte in front of it is an unescaped backslash, and
            // with 7 .. 13 backslashes per KiB (the BASELINE workloads: `\/` in URLs, `\"`, `\uXXXX`) one of a tile's 64
            // blocks does in 36 % (pretty-printed), 43 % (UTF-8-heavy), 58 % (minified) of the tiles.
            if (!tc.resolved) {
                // >= 63 consecutive backslashes in front of the tile: exact carries from the predecessor
                uint32_t to = 0;
                const uint64_t d = wait_desc(&a.ws[kDescOffset + tile - 1], a.wait_ticks, &to);
                if (to) timeout = 1;
                tc.e_in = (uint32_t)(d >> 58) & 1u;
                tc.ps_in = (uint32_t)(d >> 57) & 1u;
                add_s = add_a + add_b + tc.e_in;
                carries = add_s ^ add_a ^ add_b;
            }
            // What the carry changes in a block whose first byte is NOT a backslash: that byte is escaped and nothing
            // else (pe = backslash & ~1 = backslash, json_escape_scanner.mojo:39-45) -- a quote there is no quote.  Two
            // operations on the lanes the carries name.  Only where an escaped block STARTS with a backslash (the
            // run it opens changes parity: `\\\\` across a block border) is the scanner redone with the lanes' carries.
            const uint64_t B0 = __ballot(((uint32_t)cls.backslash & 1u) != 0u);
            if ((carries & B0) == 0ull && !partial) {  // uniform
                const bool starts_escaped = __builtin_amdgcn_inverse_ballot_w64(carries);
                const uint32_t qlo = (uint32_t)quote;
                quote = u64(starts_escaped ? qlo & ~1u : qlo, (uint32_t)(quote >> 32));
            } else {
                uint32_t lane_e_out;
                escaped = escaped_mask(cls.backslash, (uint32_t)(carries >> lane) & 1u, &lane_e_out);
                quote = cls.quote_chr & ~escaped;
            }
        }
        // ---- strings (json_string_scanner.mojo:55-69): in_string if the lane started outside a string
        const uint64_t S0 = prefix_xor(quote);
        tile_e_out = (uint32_t)(((add_a & add_b) | ((add_a | add_b) & ~add_s)) >> 63);
        const uint64_t PM = __ballot((int32_t)(uint32_t)(S0 >> 32) < 0);
        const uint32_t lane_par = lanes_below(PM);  // bit 0: parity of the lanes before me in the tile
        tile_par = (uint32_t)__popcll(PM) & 1u;
        const uint32_t lane_in32 = (uint32_t)__builtin_amdgcn_sbfe((int)lane_par, 0, 1);  // all-ones: inside a string
        const uint64_t lane_in = u64(lane_in32, lane_in32);
        // in_string assuming the TILE starts outside a string
        in_string0 = S0 ^ lane_in;
    }

    // ---- scalars (json_scanner.mojo:64-79); three-input mask operations throughout
    const uint64_t nqs = lut3<MSJ_TT(~TA & ~TB & ~TC)>(cls.op, cls.ws, quote);  // scalar & ~quote
    const uint32_t my_ps = (uint32_t)(nqs >> 63);
    const uint32_t prev_ps = dpp_shift_up1(my_ps, tc.ps_in);
    uint32_t tile_ps_out = bcast((uint32_t)(nqs >> 32), 63) >> 31;
    // follows = nonquote_scalar << 1 | carry  (json_scanner.mojo:76-79)
    const uint64_t follows = u64(((uint32_t)nqs << 1) | prev_ps, __builtin_amdgcn_alignbit((uint32_t)(nqs >> 32), (uint32_t)nqs, 31));
    // potential_structural_start = op | (scalar & ~follows) = op | (~ws & ~follows)   (json_scanner.mojo:40-49)
    const uint64_t potential = lut3<MSJ_TT(TA | (~TB & ~TC))>(cls.op, cls.ws, follows);
    Pending r;
    // structural_start = potential & ~string_tail, string_tail = in_string ^ quote
    // (json_scanner.mojo:24-26, json_string_scanner.mojo:40-44)
    r.T0 = lut3<MSJ_TT(TA & ~(TB ^ TC))>(potential, in_string0, quote);  // if tile s_in = 0
    r.T1 = lut3<MSJ_TT(TA & (TB ^ TC))>(potential, in_string0, quote);   // if tile s_in = 1
    // ---- packed inclusive scan of the per-lane structural counts
    const uint32_t pk = (uint32_t)__popcll(r.T0) | ((uint32_t)__popcll(r.T1) << 16);
    const uint32_t inc = dpp_add_scan(pk);
    r.excl = inc - pk;
    r.tile_cnt = bcast(inc, 63);
    if (partial) {
        // the carries OUT of a shard that does not end on a tile are the state after its last BYTE,
        // not after the space padding: an unescaped backslash there escapes the next shard's first
        // byte, a non-quote scalar there makes it a continuation
        const uint32_t lastpos = len - 1u - tile * kTileBytes, ll = lastpos >> 6, lb = lastpos & 63u;
        tile_e_out = bcast((uint32_t)((cls.backslash & ~escaped) >> lb) & 1u, (int)ll);
        tile_ps_out = bcast((uint32_t)(nqs >> lb) & 1u, (int)ll);
    }
    // unescaped_chars_error |= ctrl & in_string (json_structural_indexer.mojo:143-145), for both tile states
    const uint64_t e0 = cls.ctrl & in_string0;
    const uint64_t me0 = __ballot(e0 != 0ull);
    const uint64_t me1 = __ballot(e0 != cls.ctrl);
    MSJ_STAMP(tile, 4);

    // ---- utf8 (lane_math.h: structure from lead planes moved forward, the four second-byte ranges judged at
    //      the lead byte from the next byte's b5 / b5 | b4 moved backward: five cross-lane shifts)
    uint32_t tile_pend = 0;
    uint64_t mu8 = 0;
    if (!(a.flags & kFlagNoUtf8) && (tc.u8_in != 0u || nonascii != 0ull)) {
        const uint32_t lane0_mask = lane == 0u ? ~0u : 0u;                // loop invariants of the caller's loop
        const uint32_t last_pair_hi = lane == 63u ? 0x7FFFFFFFu : ~0u;
        const Utf8Leads u8l = utf8_leads(p);
        // forward: across lanes by DPP (zeros into lane 0); what the bytes in front of the tile ask of its first
        // three bytes comes from the carry word of the window bytes -- one three-bit value, ORed into lane 0
        const uint32_t c = tc.u8_in;
        Utf8Shifted sh8;
        sh8.exp1 = wave_shl_0<1>(u8l.lead234);
        sh8.exp2 = wave_shl_0<2>(u8l.lead34);
        const uint64_t e3 = wave_shl_0<3>(u8l.lead4);
        const uint32_t c_low3 = (c & 1u) | ((c >> 1) & 3u) | ((c >> 3) & 7u);  // scalar
        sh8.exp3 = u64(lut3<MSJ_TT(TA | (TB & TC))>((uint32_t)e3, lane0_mask, c_low3), (uint32_t)(e3 >> 32));
        // backward: the tile's last byte has no next byte here -- that pair is the next tile's (its window bytes)
        sh8.nb5 = wave_shr1_in(p[5]);
        sh8.nb54 = wave_shr1_in(p[5] | p[4]);
        uint64_t bad2;
        uint64_t u8bad = utf8_errors_shifted(p, u8l, sh8, &bad2);
        if (!partial) {  // uniform
            // every pair but (the tile's last byte, the next tile's first) is judged here
            u8bad |= u64((uint32_t)bad2, (uint32_t)(bad2 >> 32) & last_pair_hi);
        } else {
            asm volatile("; partial tile" ::: "memory");  // a real branch: if-converted, its masks cost every tile four operations
            // in a tile cut short by the end of the input the pair is judged only where the next byte is input too
            // (at the end of the stream the lead byte is a truncated sequence anyway; at the end of a non-final
            // shard the next shard judges the pair from its window)
            u8bad |= bad2 & wave_shr1_in(valid);
            // a character cut by the end of a NON-final shard continues in the next one (which checks it
            // from its window bytes); only at the end of the stream is the missing continuation an error
            if (!(a.flags & kFlagFinal)) u8bad &= valid;
        }
        mu8 = __ballot(u8bad != 0ull);
        // the pair (last byte in front of the tile, the tile's first byte): rare (that byte is E0 / ED / F0 / F4)
        if (c >> 6) {  // uniform
            const uint32_t f5 = bcast((uint32_t)p[5], 0), f4 = bcast((uint32_t)p[4], 0);
            if (utf8_boundary_error(c, f5, f4)) mu8 |= 1ull;
        }
        // a sequence still open at the end of the tile (the last lane's top lead bits)
        tile_pend = ((bcast((uint32_t)(u8l.lead234 >> 32), 63) >> 31) | (bcast((uint32_t)(u8l.lead34 >> 32), 63) >> 30) |
                     (bcast((uint32_t)(u8l.lead4 >> 32), 63) >> 29)) ? 1u : 0u;
    }
    MSJ_STAMP(tile, 5);
    MSJ_STAMP(tile, 6);
    // low word: the packed counts as they are; high word: status and flags
    const uint32_t hi = (uint32_t)(kAgg >> 32) | (tile_par << 29) | ((me0 ? 1u : 0u) << 28) | ((me1 ? 1u : 0u) << 27) |
                        (tile_e_out << 26) | (tile_ps_out << 25) | ((mu8 ? 1u : 0u) << 24) | (tile_pend << 23) |
                        (timeout << 22);
    agg_word = u64(r.tile_cnt, hi);
    return r;
}

// ---- BitIndexer.write (json_structural_indexer.mojo:46-58) for one parked tile, in two
//      steps so that the global stores of one range never sit between a load and its use:
//      emit_stage() writes the offsets into the wave's LDS slice at their position in the
//      output (LDS only), emit_store() turns them into aligned 16-byte
//      stores (one L2 request per 64 B instead of one per index).  Wave-local: no barrier.
// Everything that decides HOW a tile is emitted is wave-uniform and lives in scalar registers.
enum : uint32_t { kEmitNone = 0, kEmitStaged, kEmitStaged2, kEmitDense, kEmitGeneral };
struct EmitU {
    uint32_t mode;
    uint32_t tile;
    uint32_t tile_base;    // value of the tile's first byte: tile * 4096 (+ the launch's index bias)
    uint32_t s_in;         // the tile's incoming in-string state
    uint32_t shift, vend;  // stage[j] <-> idx[base - shift + j] for j in [shift, vend)
    uint64_t base;         // index of the tile's first structural in the output
};
// ... and the lane's part: its structural_start mask for that state and its first stage slot
struct EmitV {
    uint32_t tlo, thi, vpos;
};

// Picks the counts for the tile's actual incoming state from its pending slot (scalar code).
__device__ __forceinline__ EmitU emit_prepare(const KernelArgs &a, const Shared &sh, const uint32_t wave,
                                              const uint32_t slot, const uint64_t rpre_word, const uint64_t count0,
                                              uint32_t &timeout) {
    EmitU e;
    e.mode = kEmitNone;
    e.s_in = e.shift = e.vend = 0;
    e.base = 0;
    const uint4 meta = *reinterpret_cast<const uint4 *>(sh.pend_meta[wave][slot]);  // tile, tile_cnt, in_cnt, in_s
    e.tile = uniform32(meta.x);
    e.tile_base = 0;
    if (e.tile == 0xFFFFFFFFu) return e;
    e.tile_base = e.tile * kTileBytes + a.index_bias;  // < 2^32 per document
    const uint32_t tile_cnt = uniform32(meta.y);
    const uint32_t in_cnt = uniform32(meta.z);
    const uint32_t in_s = uniform32(meta.w);
    // the range's prefix (resolver) + the tile's position inside the range (local fold)
    const uint32_t q = (uint32_t)(rpre_word >> 61) & 1u;
    e.s_in = (in_s >> q) & 1u;
    e.base = count0 + (uint64_t)(uint32_t)rpre_word + (q ? (in_cnt >> 16) : (in_cnt & 0xFFFFu));
    if ((rpre_word >> 54) & 1u) timeout = 1;
    timeout = uniform32(timeout);
    if ((a.flags & kFlagNoEmit) || timeout) return e;
    const uint32_t cnt = e.s_in ? (tile_cnt >> 16) : (tile_cnt & 0xFFFFu);
    if (cnt == 0u) return e;
    e.shift = (uint32_t)e.base & 3u;
    e.vend = e.shift + cnt;
    if (e.base + cnt > a.capacity)
        e.mode = kEmitGeneral;  // the output buffer has no room for all of the tile's indices
    else if (e.vend <= kStageWords)
        e.mode = kEmitStaged;   // per-lane chains into the staging slice, 16-byte stores (up to 1020 indices: the common case)
    else if (e.vend <= 2u * kStageWords + 3u)
        e.mode = kEmitStaged2;  // up to half of the tile's bytes: the same in two rounds of the slice (the second round's
                                // copy-out moves kStageWords / 4 full quads and up to three more elements)
    else
        e.mode = kEmitDense;    // more: block by block, straight to the output
    e.mode = uniform32(e.mode);
    return e;
}

__device__ __forceinline__ EmitV emit_lane(const Shared &sh, const uint32_t wave, const uint32_t slot,
                                           const uint32_t lane, const EmitU &e) {
    EmitV v;
    // the masks of the tile's actual state: 8 of the lane's 16 parked bytes
    const uint2 m = reinterpret_cast<const uint2 *>(&sh.pend_masks[wave][slot][lane])[e.s_in];
    const uint32_t excl = sh.pend_excl[wave][slot][lane];
    v.tlo = m.x;
    v.thi = m.y;
    v.vpos = e.shift + __builtin_amdgcn_ubfe(excl, e.s_in * 16u, 16u);
    return v;
}

__device__ __forceinline__ void stage_indices(const EmitU &e, const EmitV &v, uint32_t *stage, const uint32_t lane64) {
    // the block offset is a multiple of 64, so value_base | bit == value_base + bit
    const uint32_t v0 = e.tile_base + lane64;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(stage + v.vpos);  // LDS byte address
    const uint32_t nlo = (uint32_t)__builtin_popcount(v.tlo), nhi = (uint32_t)__builtin_popcount(v.thi);
    scatter_bits32(v.tlo, lds0, nlo, v0);
    scatter_bits32(v.thi, lds0 + 4u * nlo, nhi, v0 | 32u);
}

// The index array is written once and read by a later kernel, if at all: streaming (non-temporal) stores keep
// it from displacing the input in L2 / MALL (a trivial kernel with this read : write mix gains 2 % from them,
// profiles/r02/hbm_bw_ubench_nt.txt).
__device__ __forceinline__ void st_index_quad(uint8_t *p, const uint4 v) {
#ifdef MSJ_PLAIN_INDEX_STORES
    *reinterpret_cast<uint4 *>(p) = v;
#else
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4 *>(p));
#endif
}

__device__ __forceinline__ void copy_out(const KernelArgs &a, const EmitU &e, const uint32_t *stage,
                                         const uint32_t lane, const uint32_t lane_p) {
    // uniform 64-bit base + 32-bit lane offsets; full 16-byte quads in the body, the (at
    // most two) partial quads at the ends element by element
    uint8_t *out = reinterpret_cast<uint8_t *>(a.idx + (e.base - e.shift));  // out[4v] <-> stage[v]; 16-byte aligned
    const uint8_t *src = reinterpret_cast<const uint8_t *>(stage);
    const uint32_t q_lo0 = (e.shift + 3u) >> 2;       // first quad with all four elements valid
    const uint32_t q_hi0 = e.vend >> 2;               // one past the last full quad
    // The streaming stores cover whole 128-byte lines only: the quads of the (at most two) lines this tile shares with
    // its neighbours leave as plain stores, which L2 merges with the neighbour's part -- a partially written line that
    // leaves as a non-temporal store reaches memory on its own (WRITE_SIZE 0.883 GB for 0.831 GB of indices on the 1 GiB
    // minified input).  Same box, alternating, 1 500 launches each (profiles/r03/ab_edge_plain.txt): minified 0.3289 ->
    // 0.3183 ms, 0.3261 -> 0.3199 (+2 .. +3.4 %); pretty-printed +-0.5 %; UTF-8-heavy -0.7 .. -1.1 % -- where the
    // instruction stream and not the traffic binds, the few extra instructions cost more than the lines save, so the
    // edges are only taken apart for tiles with many indices (MSJ_EDGE_PLAIN_MIN; 0 = always, ~0 = never).
#ifndef MSJ_EDGE_PLAIN_MIN
#define MSJ_EDGE_PLAIN_MIN 512u
#endif
    uint32_t q_lo = q_lo0, q_hi = q_hi0;
    if (e.vend - e.shift >= MSJ_EDGE_PLAIN_MIN) {  // uniform
        const uint32_t a16 = (uint32_t)(reinterpret_cast<uintptr_t>(out) >> 4);  // the output's address in quads (uniform)
        const uint32_t over = (a16 + q_hi0) & 7u;
        q_lo = q_lo0 + ((8u - ((a16 + q_lo0) & 7u)) & 7u);   // first quad that starts a line
        q_hi = q_hi0 >= over ? q_hi0 - over : 0u;            // one past the last quad that ends one
        if (q_lo > q_hi0) q_lo = q_hi0;
        if (q_hi < q_lo) q_hi = q_lo;
        if (lane_p < 16u) {  // the edges: [q_lo0, q_lo) and [q_hi, q_hi0), at most seven quads each
            const uint32_t q = lane_p < 8u ? q_lo0 + lane : q_hi + (lane - 8u);
            const bool ok = lane_p < 8u ? q < q_lo : q < q_hi0;
            if (ok) *reinterpret_cast<uint4 *>(out + 16u * q) = *reinterpret_cast<const uint4 *>(src + 16u * q);
        }
    }
    // at most kStageWords / 256 = 4 rounds of 64 quads: one byte offset per lane, the rounds are
    // immediate offsets of the LDS read and of the store; only the last round is partial
    const uint32_t off = (q_lo + lane) * 16u;
#pragma unroll
    for (uint32_t k = 0; k < kStageWords / 256u; k++) {
        if (q_lo + 64u * k >= q_hi) break;  // uniform
        if (q_lo + 64u * (k + 1u) <= q_hi) {  // uniform: a full round, every lane stores
            st_index_quad(out + off + 1024u * k, *reinterpret_cast<const uint4 *>(src + off + 1024u * k));
        } else {  // the last round: the first (q_hi - q_lo - 64k) lanes
            if (lane_p < q_hi - q_lo - 64u * k)
                st_index_quad(out + off + 1024u * k, *reinterpret_cast<const uint4 *>(src + off + 1024u * k));
            break;
        }
    }
    // head (elements shift .. 4*q_lo) and tail (4*q_hi .. vend): < 8 elements in total, one
    // element per lane of the first eight
    if (lane_p < 8u) {
        const bool head = lane_p < 4u;
        const uint32_t v = head ? lane : 4u * q_hi0 + (lane - 4u);
        const bool ok = head ? (v >= e.shift && v < 4u * q_lo0 && v < e.vend) : (v < e.vend && v >= 4u * q_lo0);
        if (ok) *reinterpret_cast<uint32_t *>(out + 4u * v) = *reinterpret_cast<const uint32_t *>(src + 4u * v);
    }
}

// kTypes (prototype): the type byte of four indices -- four byte gathers from the tile's bytes (L2 / MALL: the tile was
// read two iterations ago), packed into the dword that lies beside the quad in types[].  Measured (scripts/fused_types.py,
// profiles/r05/fused_types_*.txt): the gathers cost stage 1 0.235 ms per GiB minified (0.314 -> 0.549) -- 16 load
// instructions per tile, each 64 scattered bytes = up to 32 cache lines: eight times the requests of the tile's own read --
// and requesting all of a tile's rounds before the first is looked at made it 0.573, not faster: it is the texture
// path's request rate, not the round trips.  The next step would be the tile's 4 KiB re-read coalesced into the staging
// slice behind the copy-out and ds_read_u8 gathers from there.
__device__ __forceinline__ uint32_t gather_types4(const uint8_t *bytes, const uint4 v) {
    return (uint32_t)bytes[v.x] | ((uint32_t)bytes[v.y] << 8) | ((uint32_t)bytes[v.z] << 16) | ((uint32_t)bytes[v.w] << 24);
}
__device__ __forceinline__ void copy_out_types(const KernelArgs &a, const EmitU &e, const uint32_t *stage,
                                         const uint32_t lane, const uint32_t lane_p) {
    // types[] runs parallel to idx[]: element (e.base - e.shift) + v of either belongs to stage[v]; offsets are byte offsets
    // of the document (+ index_bias), so `bytes + offset` is the byte itself
    constexpr bool kTypes = true;
    uint8_t *tout = a.types + (e.base - e.shift);
    const uint8_t *bytes = a.buf - a.index_bias;
    // uniform 64-bit base + 32-bit lane offsets; full 16-byte quads in the body, the (at
    // most two) partial quads at the ends element by element
    uint8_t *out = reinterpret_cast<uint8_t *>(a.idx + (e.base - e.shift));  // out[4v] <-> stage[v]; 16-byte aligned
    const uint8_t *src = reinterpret_cast<const uint8_t *>(stage);
    const uint32_t q_lo0 = (e.shift + 3u) >> 2;       // first quad with all four elements valid
    const uint32_t q_hi0 = e.vend >> 2;               // one past the last full quad
    // The streaming stores cover whole 128-byte lines only: the quads of the (at most two) lines this tile shares with
    // its neighbours leave as plain stores, which L2 merges with the neighbour's part -- a partially written line that
    // leaves as a non-temporal store reaches memory on its own (WRITE_SIZE 0.883 GB for 0.831 GB of indices on the 1 GiB
    // minified input).  Same box, alternating, 1 500 launches each (profiles/r03/ab_edge_plain.txt): minified 0.3289 ->
    // 0.3183 ms, 0.3261 -> 0.3199 (+2 .. +3.4 %); pretty-printed +-0.5 %; UTF-8-heavy -0.7 .. -1.1 % -- where the
    // instruction stream and not the traffic binds, the few extra instructions cost more than the lines save, so the
    // edges are only taken apart for tiles with many indices (MSJ_EDGE_PLAIN_MIN; 0 = always, ~0 = never).
#ifndef MSJ_EDGE_PLAIN_MIN
#define MSJ_EDGE_PLAIN_MIN 512u
#endif
    uint32_t q_lo = q_lo0, q_hi = q_hi0;
    if (e.vend - e.shift >= MSJ_EDGE_PLAIN_MIN) {  // uniform
        const uint32_t a16 = (uint32_t)(reinterpret_cast<uintptr_t>(out) >> 4);  // the output's address in quads (uniform)
        const uint32_t over = (a16 + q_hi0) & 7u;
        q_lo = q_lo0 + ((8u - ((a16 + q_lo0) & 7u)) & 7u);   // first quad that starts a line
        q_hi = q_hi0 >= over ? q_hi0 - over : 0u;            // one past the last quad that ends one
        if (q_lo > q_hi0) q_lo = q_hi0;
        if (q_hi < q_lo) q_hi = q_lo;
        if (lane_p < 16u) {  // the edges: [q_lo0, q_lo) and [q_hi, q_hi0), at most seven quads each
            const uint32_t q = lane_p < 8u ? q_lo0 + lane : q_hi + (lane - 8u);
            const bool ok = lane_p < 8u ? q < q_lo : q < q_hi0;
            if (ok) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + 16u * q);
                *reinterpret_cast<uint4 *>(out + 16u * q) = v;
                if (kTypes) *reinterpret_cast<uint32_t *>(tout + 4u * q) = gather_types4(bytes, v);
            }
        }
    }
    // at most kStageWords / 256 = 4 rounds of 64 quads: one byte offset per lane, the rounds are
    // immediate offsets of the LDS read and of the store; only the last round is partial
    const uint32_t off = (q_lo + lane) * 16u;
#pragma unroll
    for (uint32_t k = 0; k < kStageWords / 256u; k++) {
        if (q_lo + 64u * k >= q_hi) break;  // uniform
        if (q_lo + 64u * (k + 1u) <= q_hi) {  // uniform: a full round, every lane stores
            const uint4 v = *reinterpret_cast<const uint4 *>(src + off + 1024u * k);
            st_index_quad(out + off + 1024u * k, v);
            if (kTypes) *reinterpret_cast<uint32_t *>(tout + (off >> 2) + 256u * k) = gather_types4(bytes, v);
        } else {  // the last round: the first (q_hi - q_lo - 64k) lanes
            if (lane_p < q_hi - q_lo - 64u * k) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src + off + 1024u * k);
                st_index_quad(out + off + 1024u * k, v);
                if (kTypes) *reinterpret_cast<uint32_t *>(tout + (off >> 2) + 256u * k) = gather_types4(bytes, v);
            }
            break;
        }
    }
    // head (elements shift .. 4*q_lo) and tail (4*q_hi .. vend): < 8 elements in total, one
    // element per lane of the first eight
    if (lane_p < 8u) {
        const bool head = lane_p < 4u;
        const uint32_t v = head ? lane : 4u * q_hi0 + (lane - 4u);
        const bool ok = head ? (v >= e.shift && v < 4u * q_lo0 && v < e.vend) : (v < e.vend && v >= 4u * q_lo0);
        if (ok) {
            const uint32_t x = *reinterpret_cast<const uint32_t *>(src + 4u * v);
            *reinterpret_cast<uint32_t *>(out + 4u * v) = x;
            if (kTypes) tout[v] = bytes[x];
        }
    }
}

__device__ __forceinline__ uint32_t lane64_of(const uint32_t lane) { return lane * 64u; }
__device__ __forceinline__ void lds_wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Dense tile (more than kStageWords - 4 indices: more than a quarter of its bytes are structural) whose indices
// fit the output buffer.  Per-lane chains would run with a few lanes at a time here (a staging slice holds the
// indices of ~15 lanes), so the roles are turned around: block by block, the block's 64-bit mask is read into
// scalar registers and becomes the EXEC mask; lane j then stands for byte j of the block, its rank among the
// set bits below it (v_mbcnt) is its slot, and one store instruction writes the block's indices -- up to 256
// contiguous bytes -- straight to the output.  16 instructions per 64-byte block whatever the density; no
// staging, no LDS.  (Plain stores: a block's store covers a part of one or two 128-byte lines and the next block's
// store the rest; as non-temporal stores the parts reach memory one by one -- `[10,10,...`: 0.76 -> 1.09 ms per GiB.)
__device__ __noinline__ void emit_dense(uint32_t *idx, const uint32_t tile_base, const uint64_t base, const uint32_t tlo,
                                        const uint32_t thi, const uint32_t first_slot, const uint32_t lane) {
    // out[k] = the tile's k-th index; wave-uniform (the arguments of a called function arrive in vector registers)
    const uint64_t out = uniform64(reinterpret_cast<uint64_t>(idx + base));
    const uint32_t tb = uniform32(tile_base);
    for (uint32_t b = 0; b < 64u; b++) {  // uniform
        const uint32_t mlo = bcast(tlo, (int)b), mhi = bcast(thi, (int)b);
        if ((mlo | mhi) == 0u) continue;
        const uint32_t slot0 = bcast(first_slot, (int)b);  // slot of the block's first index
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        const uint32_t off = (slot0 + rank) * 4u;
        const uint32_t val = tb + b * 64u + lane;
        const uint64_t m = u64(mlo, mhi);
        uint64_t save;
        asm volatile(
            "s_mov_b64 %[save], exec\n"
            "s_mov_b64 exec, %[m]\n"
            "global_store_dword %[off], %[val], %[out]\n"
            "s_mov_b64 exec, %[save]\n"
            : [save] "=&s"(save)
            : [m] "s"(m), [off] "v"(off), [val] "v"(val), [out] "s"(out)
            : "memory");
    }
}

// An index buffer that is too small (the launch reports CAPACITY): element-wise, clipped.
__device__ __noinline__ void emit_general(uint32_t *idx, const uint64_t capacity, const uint32_t tile_base,
                                          const uint64_t base, const uint32_t shift, const uint32_t vend,
                                          uint32_t tlo, uint32_t thi, uint32_t vpos, uint32_t *stage,
                                          const uint32_t lane) {
    const uint32_t v0 = tile_base + lane * 64u;
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
            if (vq >= shift && vq + 4u <= lim && g + 4u <= capacity) {
                *reinterpret_cast<uint4 *>(&idx[g]) = val;
            } else {
                const uint32_t vv[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t v = vq + j;
                    if (v >= shift && v < lim && g + j < capacity) idx[g + j] = vv[j];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // stage is reused by the next round / next tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// PROTOTYPE (kFlagEmitTypes): the same two with the type byte of every index written beside it
__device__ __noinline__ void emit_dense_types(uint32_t *idx, const uint32_t tile_base, const uint64_t base, const uint32_t tlo,
                                        const uint32_t thi, const uint32_t first_slot, const uint32_t lane,
                                        uint8_t *types, const uint8_t *bytes) {
    // out[k] = the tile's k-th index; wave-uniform (the arguments of a called function arrive in vector registers)
    const uint64_t out = uniform64(reinterpret_cast<uint64_t>(idx + base));
    const uint32_t tb = uniform32(tile_base);
    for (uint32_t b = 0; b < 64u; b++) {  // uniform
        const uint32_t mlo = bcast(tlo, (int)b), mhi = bcast(thi, (int)b);
        if ((mlo | mhi) == 0u) continue;
        const uint32_t slot0 = bcast(first_slot, (int)b);  // slot of the block's first index
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        const uint32_t off = (slot0 + rank) * 4u;
        const uint32_t val = tb + b * 64u + lane;
        const uint64_t m = u64(mlo, mhi);
        uint64_t save;
        asm volatile(
            "s_mov_b64 %[save], exec\n"
            "s_mov_b64 exec, %[m]\n"
            "global_store_dword %[off], %[val], %[out]\n"
            "s_mov_b64 exec, %[save]\n"
            : [save] "=&s"(save)
            : [m] "s"(m), [off] "v"(off), [val] "v"(val), [out] "s"(out)
            : "memory");
        if (types && ((m >> lane) & 1ull)) types[base + slot0 + rank] = bytes[val];  // (prototype: kFlagEmitTypes)
    }
}

// An index buffer that is too small (the launch reports CAPACITY): element-wise, clipped.
__device__ __noinline__ void emit_general_types(uint32_t *idx, const uint64_t capacity, const uint32_t tile_base,
                                          const uint64_t base, const uint32_t shift, const uint32_t vend,
                                          uint32_t tlo, uint32_t thi, uint32_t vpos, uint32_t *stage,
                                          const uint32_t lane, uint8_t *types, const uint8_t *bytes) {
    const uint32_t v0 = tile_base + lane * 64u;
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
            if (vq >= shift && vq + 4u <= lim && g + 4u <= capacity) {
                *reinterpret_cast<uint4 *>(&idx[g]) = val;
                if (types) *reinterpret_cast<uint32_t *>(&types[g]) = gather_types4(bytes, val);
            } else {
                const uint32_t vv[4] = {val.x, val.y, val.z, val.w};
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t v = vq + j;
                    if (v >= shift && v < lim && g + j < capacity) {
                        idx[g + j] = vv[j];
                        if (types) types[g + j] = bytes[vv[j]];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // stage is reused by the next round / next tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// Range prefix for this wave: requested once per range, polled only if the resolver
// had not published it yet.
__device__ __forceinline__ uint64_t range_prefix(const KernelArgs &a, const uint64_t *rpre, uint32_t range_id,
                                                 uint64_t word, uint32_t &timeout) {
    if ((word >> 62) == 0ull) {
        uint32_t to = 0;
        word = wait_desc(&rpre[range_id], a.wait_ticks, &to);
        if (to) {
            timeout = 1;
            word = kPre | (1ull << 54);
        }
    }
    return word;
}

// The two halves of one parked tile's emission: LDS-only staging (the one-round cases), then
// everything that stores to global memory (copy-out, or the multi-round / clipped paths).
// Two rounds of the staging slice for a tile with 1 021 .. 2 048 indices (a quarter to a half of its bytes:
// BitIndexer.write, json_structural_indexer.mojo:46-58, at the density of `[123,123,...`).  Round 0 is the lanes
// whose first slot lies in the slice's first kStageWords, round 1 the lanes whose last slot lies beyond them; the one
// lane that straddles the border runs its chain in both rounds and writes the other round's part into the slack
// words behind (round 0) or in front of (round 1) the slice, which nobody copies out.
__device__ __forceinline__ void stage_indices_round(const EmitU &e, EmitV v, uint32_t *stage, const uint32_t lane64,
                                                    const uint32_t round) {
    const uint32_t n = (uint32_t)__builtin_popcount(v.tlo) + (uint32_t)__builtin_popcount(v.thi);
    const bool mine = round == 0u ? v.vpos < kStageWords : v.vpos + n > kStageWords;
    if (!mine) v.tlo = v.thi = 0u;
    stage_indices(e, v, stage - round * kStageWords, lane64);
}
// what copy_out moves in a round: round 0 slots [shift, kStageWords), round 1 slots [kStageWords, vend) from the slice's start
__device__ __forceinline__ EmitU emit_round(const EmitU &e, const uint32_t round) {
    EmitU r = e;
    if (round == 0u) {
        r.vend = kStageWords;
    } else {
        r.base = e.base - e.shift + kStageWords;  // 16-byte aligned like base - shift
        r.shift = 0u;
        r.vend = e.vend - kStageWords;
    }
    return r;
}

__device__ __forceinline__ void emit_stage(const Shared &sh, EmitU &e, const uint32_t wave, const uint32_t slot,
                                           uint32_t *stage, const uint32_t lane, const uint32_t lane64) {
    if (e.mode == kEmitStaged) {  // uniform
        const EmitV v = emit_lane(sh, wave, slot, lane, e);
        stage_indices(e, v, stage, lane64);
    } else if (e.mode == kEmitStaged2) {
        const EmitV v = emit_lane(sh, wave, slot, lane, e);
        // A tile whose blocks all hold the same number of indices, a multiple of 32 (`[123,123,...`: 32 per block),
        // puts every lane's k-th write on the same LDS bank: measured 1.03 ms per GiB against 0.68 for the block-wise
        // emission.  More than half of the lanes on one bank: leave the tile to emit_dense.
        const uint64_t same = __ballot(((v.vpos ^ bcast(v.vpos, 0)) & 31u) == 0u);
        if (__popcll(same) > 32) {  // uniform
            e.mode = kEmitDense;
            return;
        }
        stage_indices_round(e, v, stage, lane64, 0u);
    }
}
template <bool kTypes = false>
__device__ __forceinline__ void emit_store(const KernelArgs &a, const Shared &sh, const EmitU &e, const uint32_t wave,
                                           const uint32_t slot, uint32_t *stage, const uint32_t lane, const uint32_t lane_p) {
    if (e.mode == kEmitStaged) {  // uniform
        if (kTypes) copy_out_types(a, e, stage, lane, lane_p); else copy_out(a, e, stage, lane, lane_p);
    } else if (e.mode == kEmitStaged2) {
        if (kTypes) copy_out_types(a, emit_round(e, 0u), stage, lane, lane_p); else copy_out(a, emit_round(e, 0u), stage, lane, lane_p);
        lds_wave_sync();  // the slice is reused by the second round
        const EmitV v = emit_lane(sh, wave, slot, lane, e);
        stage_indices_round(e, v, stage, lane64_of(lane), 1u);
        lds_wave_sync();
        if (kTypes) copy_out_types(a, emit_round(e, 1u), stage, lane, lane_p); else copy_out(a, emit_round(e, 1u), stage, lane, lane_p);
    } else if (e.mode == kEmitDense) {
        const EmitV v = emit_lane(sh, wave, slot, lane, e);
        if (kTypes)
            emit_dense_types(a.idx, e.tile_base, e.base, v.tlo, v.thi, v.vpos - e.shift, lane, a.types, a.buf - a.index_bias);
        else
            emit_dense(a.idx, e.tile_base, e.base, v.tlo, v.thi, v.vpos - e.shift, lane);
    } else if (e.mode == kEmitGeneral) {
        const EmitV v = emit_lane(sh, wave, slot, lane, e);
        if (kTypes)
            emit_general_types(a.idx, a.capacity, e.tile_base, e.base, e.shift, e.vend, v.tlo, v.thi, v.vpos, stage, lane, a.types, a.buf - a.index_bias);
        else
            emit_general(a.idx, a.capacity, e.tile_base, e.base, e.shift, e.vend, v.tlo, v.thi, v.vpos, stage, lane);
    }
}

// ---- worker: one wave, persistent ---------------------------------------------------
// Work distribution.  One atomic counter hands out tiles in ascending order, but a
// single word sustains only ~80-90 returning atomics per microsecond chip-wide, so one
// atomic must pay for many tiles: thread 0 of a workgroup draws a RANGE of
// kWaves * kBatch tiles, and wave w takes tiles lo + kWaves*j + w (j = 0..kBatch-1).
//
// One range per loop iteration, in this order:
//   1. compute the wave's kBatch tiles (their bytes are in registers), publish each tile
//      aggregate; thread 0 draws the ticket of the NEXT range after the first tile;
//   2. workgroup barrier (the only one); fold the range's tile aggregates and publish the
//      range aggregate for the resolver -- before anything in this iteration can block;
//   3. emit the tiles of the range computed kDefer iterations ago (parked in LDS): their
//      range prefix has had kDefer iterations to arrive.  In between, as soon as the ticket
//      is back (wave 0 hands it to the other waves through LDS, no barrier), the next
//      range's bytes are requested; they are waited for before the LAST tile's stores, so
//      no store sits between a load and its use, and those last stores drain during step 1;
//   4. park this iteration's tiles in the freed slots.
//
// The resolver retires ranges in order, so what matters is how far the publish times of
// neighbouring ranges spread: a ticket is drawn as late as the latencies (atomic, HBM)
// allow, one emission phase before the range is computed, and nothing that can block sits
// between the draw and the publish except that one emission.
//
// Deadlock freedom: a wave blocks only in step 3, after its own range is published, and
// only on the prefix of a range below every range its workgroup holds (drawn or parked).
// The workgroup holding the smallest unpublished range therefore never waits on anything
// that needs a later range, whatever the dispatch order or residency, and the resolver
// publishes a range's prefix as soon as every earlier range is in (partial progress).
//
// Vector instructions are what this kernel runs out of (profiles/): everything wave-uniform --
// tile numbers, addresses, counts, the emission's mode -- is kept in scalar registers, lane-derived
// offsets are loop invariants, and the rare cases (errors in a range, a tile cut by the end of
// the input, escapes that cross a block) branch off uniformly.
template <bool kTypes>
__device__ __forceinline__ void worker_wave(const KernelArgs &a, Shared &sh, const uint32_t lane,
                                            const uint32_t wave) {
    // Ticket shard and first range: drawn at the kernel's entry (stage1_kernel), one atomic for both.
    // A shard's k-th draw is range k * shards + shard; shard 0's first draw made the resolver, so its
    // numbering starts one later (ticket_range).
    const uint32_t workers = gridDim.x - 1u;
    const uint32_t shards = workers < kTicketShards ? workers : kTicketShards;
    const uint32_t shard = uniform32(sh.shard);
    uint32_t *stage = sh.stage[wave] + kStageSlack;
    const uint32_t tid = threadIdx.x;
    (void)tid;  // the diagnostic builds' stamps
    const uint32_t lane64 = lane * 64u;  // loop invariants in vector registers
    const uint32_t lane16 = lane * 16u;  // the coalesced load shape: piece k of lane l is chunk 64 k + l of the tile
    const uint32_t lane_off[4] = {lane16, lane16 + 1024u, lane16 + 2048u, lane16 + 3072u};
    const ChunkAddr chunk_at = chunk_addresses(stage, lane);
    const uint32_t ntiles = a.ntiles;
    const uint32_t nranges = (ntiles + kRange - 1u) / kRange;
    uint64_t *ragg = a.ws + kDescOffset + ntiles;
    const uint64_t *rpre = ragg + nranges;
    if (lane < kPendSlots) sh.pend_meta[wave][lane][0] = 0xFFFFFFFFu;  // all slots empty (read by this wave only)
    MSJ_RSTAMP(a.ntiles + 4096u + blockIdx.x, 2, tid == 0);
    // LDS words every lane reads identically: uniform (tile indices and all control flow stay scalar)
    uint32_t lo_cur = uniform32(sh.first_lo);
#if MSJ_EARLY_A
    uint32_t lo_nxt = uniform32(sh.second_lo);  // the range behind lo_cur, known an iteration early
    uint32_t nt_next = 0u;                      // ... and whether its bytes are requested non-temporally
#endif
    const uint64_t count0 = uniform64(cin_count(a));  // launch invariants: read once
    const uint32_t carry0 = uniform32(cin_carry0(a));
    uint32_t timeout = 0;

    Block blk[kBatch];
    load_range<false>(a, lo_cur, wave, lane_off, lane, blk);  // the first range: plain (all waves ask at once: + 0.2 .. 0.7 %)
#pragma unroll
    for (uint32_t j = 0; j < kBatch; j++) touch_block(blk[j]);  // loop invariant: the bytes have arrived
    MSJ_RSTAMP(a.ntiles + 4096u + blockIdx.x, 3, tid == 0);  // first bytes in registers

    uint32_t r = 0, ring = 0;  // ring = r % kDefer: the slots to emit from, then to park in
    while (lo_cur < ntiles) {  // uniform across the workgroup
        // Predicates on the lane / wave index and tests of the launch's flags are recomputed where they are used (one
        // compare each) instead of living in scalar register pairs across the whole loop: the loop holds more uniform
        // values than there are scalar registers, and every pair kept costs two v_readlane per use once it is spilled.
        uint32_t lane_p = lane, wave_p = wave;
        asm volatile("" : "+v"(lane_p));
        asm volatile("" : "+s"(wave_p));
        KernelArgs al = a;
        asm volatile("" : "+s"(al.flags));
        // (Round 4, measured: reading the arguments the emission and the rare paths use -- idx, capacity, ws_clean,
        // wait_ticks, index_bias -- again from the kernel-argument segment at the top of every iteration, so that they
        // need not live in scalar registers across it, RAISED the spilled registers from 44 to 52: the pressure peaks
        // inside the iteration, in the two tiles' wave-level masks, not across its back edge.  The ~31 reload sites in
        // the loop are <= 1.5 % of its vector instructions.)
        const uint32_t tid_p = wave_p * 64u + lane_p;
        uint32_t carry0_p = carry0, shard_p = shard;  // the same for values the loop only takes bits or addresses from
        asm volatile("" : "+s"(carry0_p), "+s"(shard_p));
        unsigned int *ticket_ctr_p = reinterpret_cast<unsigned int *>(al.ws + (uint64_t)shard_p * kTicketStrideWords);
        const uint32_t par = r & 1u;
        // prefix of the range parked in this ring position (published long ago, normally)
        const uint32_t old_first = uniform32(sh.pend_meta[wave][ring * kBatch][0]);  // its first tile of this wave
        const bool have_old = old_first != 0xFFFFFFFFu;  // later tiles of a wave are past the end if the first is
        const uint32_t old_range = have_old ? old_first / kRange : 0u;
        uint64_t rp_word = ld_desc(&rpre[old_range]);
        MSJ_RSTAMP(old_range * kRange, 10, tid == 0 && have_old);  // prefix word sampled (real time)
        // ---- 1. compute
        Pending now[kBatch];
        uint32_t req_reg = 0;
#if MSJ_TICKET_AT_TOP
        // EXPERIMENT (round 5): the next range's ticket drawn at the TOP of the iteration -- both compute phases cover the
        // atomic's round trip instead of one (on sparse input one phase is shorter than the round trip)
        if (tid_p == 0) req_reg = ticket_request(ticket_ctr_p, 0u, 1u);
#endif
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) {
            const uint32_t t_cur = lo_cur + kWaves * j + wave;
            const bool valid_tile = t_cur < ntiles;
            MSJ_STAMP(valid_tile ? t_cur : ntiles - 1u, 0);
            uint64_t agg_word = kAgg;  // a tile past the end: identity
            now[j].T0 = 0;
            now[j].T1 = 0;
            now[j].excl = 0;
            now[j].tile_cnt = 0;
            if (valid_tile) {
                // the chunks the coalesced loads brought become this lane's block (the staging slice is free: the
                // last emission's copy-out has been waited for)
                chunks_to_block(blk[j], chunk_at);
#if MSJ_EARLY_A
                if (j == 1) {
                    // the first tile's registers are free since its compute phase: the NEXT range's first tile goes there now
                    // and has this compute phase, the barrier and the emission to arrive
                    const uint32_t tn = lo_nxt + wave;
                    if (nt_next) {
                        asm volatile("; non-temporal early loads");
                        load_block<true, true>(a, tn < ntiles ? tn : ntiles - 1u, lane_off, lane, blk[0]);
                        asm volatile("; end of non-temporal early loads");
                    } else {
                        load_block<true, false>(a, tn < ntiles ? tn : ntiles - 1u, lane_off, lane, blk[0]);
                    }
                }
#endif
                now[j] = compute_tile(al, t_cur, lane, blk[j], carry0_p, timeout, agg_word);
                lds_wave_sync_early();  // the next tile's chunks (or the emission) reuse the slice
            }
#if MSJ_EARLY_A
            else if (j == 1) {  // (a range whose second tile lies past the end: the next range's does too -- nothing to ask for,
                                //  but the registers must hold SOMETHING loaded for the waits' bookkeeping: nothing is waited for)
            }
#endif
            if (j == 0) {
#if MSJ_PRIO_COMPUTE2 != MSJ_PRIO_COMPUTE
                __builtin_amdgcn_s_setprio(MSJ_PRIO_COMPUTE2);
#endif
                // the prefix word requested above has arrived (nothing younger is in flight yet) ...
#if MSJ_TICKET_AT_TOP
                // (wave 0's ticket is younger than the prefix word and may stay out: returns come in order)
                if (wave_p == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                touch_u64(rp_word);
#if MSJ_EARLY_A
                touch_block(blk[1]);  // ... and so has this range's second tile (requested before the last barrier)
#endif
                // ... and the next range's ticket is drawn: one compute + one staging phase
                // ahead of its use, which covers the atomic's round trip
#if !MSJ_TICKET_AT_TOP
                if (tid_p == 0) req_reg = ticket_request(ticket_ctr_p, 0u, 1u);
#endif
            }
            if (lane_p == 0) {
                if (valid_tile) st_desc(&a.ws[kDescOffset + t_cur], agg_word);  // carries for t_cur + 1
                sh.tagg[par][kWaves * j + wave] = agg_word;
            }
            MSJ_STAMP(valid_tile ? t_cur : ntiles - 1u, 7);
        }
#if MSJ_EARLY_A
        {   // the next range's SECOND tile: this range's has been consumed, its registers are free
            const uint32_t c0 = now[0].tile_cnt, c1 = now[1].tile_cnt;
            const uint32_t n0 = (c0 & 0xFFFFu) > (c0 >> 16) ? (c0 & 0xFFFFu) : (c0 >> 16);
            const uint32_t n1 = (c1 & 0xFFFFu) > (c1 >> 16) ? (c1 & 0xFFFFu) : (c1 >> 16);
            const uint32_t tn = lo_nxt + kWaves + wave;
            if (nt_next) {
                asm volatile("; non-temporal early loads B");
                load_block<true, true>(a, tn < ntiles ? tn : ntiles - 1u, lane_off, lane, blk[1]);
                asm volatile("; end of non-temporal early loads B");
            } else {
                load_block<true, false>(a, tn < ntiles ? tn : ntiles - 1u, lane_off, lane, blk[1]);
            }
            nt_next = (uniform32(n0 + n1) > kNtMaxIndices || ntiles < kNtMinTiles) ? 0u : 1u;  // for the range after that
        }
#endif
        const uint32_t srow = (lo_cur + kWaves + wave < ntiles) ? lo_cur + kWaves + wave : ntiles - 1u;  // stamp row (diagnostic builds)
        (void)srow;
        MSJ_STAMP(srow, 8);   // both tiles computed
        // workgroup barrier for the LDS words only: the descriptor stores above need not have
        // completed (__syncthreads() would wait for them)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MSJ_STAMP(srow, 9);   // barrier passed
        __builtin_amdgcn_s_setprio(MSJ_PRIO_COORD);
        // wave 0 passes the next range's ticket on (drawn a compute phase ago) ...
        if (wave_p == 0) {
            const uint32_t v = ticket_value(req_reg);
            if (lane_p == 0)
                __hip_atomic_store(&sh.handoff, ((uint64_t)(r + 1u) << 32) | (uint64_t)(ticket_range(v, shard_p, shards) * kRange),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // ---- 2. fold the range's kRange tile aggregates in tile order -> the range aggregate
        //      for the resolver, and every tile's state / count inside the range for both range
        //      states.  Lane k (< kRange) holds tile k's aggregate: the in-range parity comes from
        //      one ballot (scalar code from there), the counts from a three-step DPP scan.
        uint32_t in_cnt[kBatch], in_state[kBatch];
        {
            const uint64_t w = sh.tagg[par][lane & (kRange - 1u)];
            const uint32_t wl = (uint32_t)w, wh = (uint32_t)(w >> 32);
            // lanes 8.. repeat lanes 0..7: the low byte of a ballot is what the range holds
            const uint32_t P8 = (uint32_t)__ballot((wh & (1u << 29)) != 0u) & 0xFFu;  // tile parities (bit 61)
            const uint32_t s0 = __builtin_amdgcn_mbcnt_lo(P8, 0u) & 1u;  // tile k's state if the range starts outside a string
            // count of tile k under range state 0 | state 1 << 16  (state 1 sees the other one)
            const uint32_t ab = __builtin_amdgcn_alignbit(wl, wl, s0 << 4);  // the tile's packed counts, halves swapped if s0
            uint32_t inc = ab;
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, false);  // row_shr:1
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, false);  // row_shr:2
            inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, false);  // row_shr:4
            const uint32_t excl = inc - ab;
            // error / poison bits (60, 59, 56, 54): none in nearly every range
            const uint32_t anyflag = (uint32_t)__ballot((wh & 0x19400000u) != 0u) & 0xFFu;
            uint32_t e0s = 0, e1s = 0, u8s = 0, pzs = 0;
            if (anyflag) {  // uniform
                const uint32_t e0 = (wh >> 28) & 1u, e1 = (wh >> 27) & 1u;
                e0s = ((uint32_t)__ballot((s0 ? e1 : e0) != 0u) & 0xFFu) != 0u;
                e1s = ((uint32_t)__ballot((s0 ? e0 : e1) != 0u) & 0xFFu) != 0u;
                u8s = ((uint32_t)__ballot(((wh >> 24) & 1u) != 0u) & 0xFFu) != 0u;  // bit 56
                pzs = ((uint32_t)__ballot(((wh >> 22) & 1u) != 0u) & 0xFFu) != 0u;  // bit 54
            }
#pragma unroll
            for (uint32_t j = 0; j < kBatch; j++) {
                const uint32_t k = kWaves * j + wave;  // this wave's tiles of the range
                in_cnt[j] = bcast(excl, (int)k);
                // tile k's state under range state 0 (bit 0) and 1 (bit 1): parity of the tiles before it
                const uint32_t sk = (uint32_t)__builtin_popcount(P8 & ((1u << k) - 1u)) & 1u;
                in_state[j] = sk | ((sk ^ 1u) << 1);
            }
            if (tid_p == 0) {
                const uint32_t tot = bcast(inc, (int)kRange - 1);
                const uint32_t s_out = (uint32_t)__builtin_popcount(P8) & 1u;
                st_desc(&ragg[lo_cur / kRange], kAgg | ((uint64_t)s_out << 61) | ((uint64_t)e0s << 60) |
                                                    ((uint64_t)e1s << 59) | ((uint64_t)u8s << 56) |
                                                    ((uint64_t)pzs << 54) | ((uint64_t)(tot >> 16) << 16) |
                                                    (uint64_t)(tot & 0xFFFFu));
            }
        }
        if (al.ws_clean && wave_p == kWaves - 1) {  // uniform
            // double-buffered workspace: zero, in the buffer the NEXT launch will use, exactly
            // the words this range dirtied in the previous launch (same ntiles, same layout)
            const uint32_t rid = lo_cur / kRange;
            if (lane_p < kRange) {
                if (lo_cur + lane < ntiles) al.ws_clean[kDescOffset + lo_cur + lane] = 0ull;
            } else if (lane_p == kRange) {
                al.ws_clean[kDescOffset + ntiles + rid] = 0ull;
            } else if (lane_p == kRange + 1u) {
                al.ws_clean[kDescOffset + ntiles + nranges + rid] = 0ull;
            }
        }
        MSJ_STAMP(srow, 10);  // folded, range aggregate published
        // ... and everybody requests the next range's bytes: in flight during the whole emission
        uint32_t lo_next;
        {
            uint64_t h = uniform64(__hip_atomic_load(&sh.handoff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            WaitClock clk;
            while ((uint32_t)(h >> 32) != r + 1u) {
                __builtin_amdgcn_s_sleep(1);
                h = uniform64(__hip_atomic_load(&sh.handoff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (clk.expired(al.wait_ticks)) {
                    // wave 0 never came back with the ticket: give up (this wave drains and leaves;
                    // a wave that has ended does not hold up the others' barrier)
                    timeout = 1;
                    h = ((uint64_t)(r + 1u) << 32) | 0xFFFFFFFFull;
                }
            }
            lo_next = (uint32_t)h;
        }
#if !MSJ_EARLY_A
        {
            // the load policy follows the data (load_block): dense input -> plain loads
            const uint32_t c0 = now[0].tile_cnt, c1 = now[1].tile_cnt;
            const uint32_t n0 = (c0 & 0xFFFFu) > (c0 >> 16) ? (c0 & 0xFFFFu) : (c0 >> 16);
            const uint32_t n1 = (c1 & 0xFFFFu) > (c1 >> 16) ? (c1 & 0xFFFFu) : (c1 >> 16);
            // (the two markers keep the compiler from merging the branches: merged loads lose the non-temporal hint)
            if (uniform32(n0 + n1) > kNtMaxIndices || ntiles < kNtMinTiles) {  // uniform
                load_range<false>(a, lo_next, wave, lane_off, lane, blk);
            } else {
                asm volatile("; non-temporal range loads");
                load_range<true>(a, lo_next, wave, lane_off, lane, blk);
                asm volatile("; end of non-temporal range loads");
            }
        }
#endif
        MSJ_RSTAMP(lo_cur, 8, tid == 0);  // range aggregate published (real time)
        // ---- 3. emit the range parked kDefer iterations ago; hand the next range over in between
        if (have_old) {
            rp_word = range_prefix(al, rpre, old_range, uniform64(rp_word), timeout);
            MSJ_RSTAMP(old_range * kRange, 9, tid == 0);  // its prefix is in hand (real time)
        }
        // wave-uniform by construction; say so, or the whole emission is vector code with exec masks
        const uint64_t rp = uniform64(rp_word);
        timeout = uniform32(timeout);
        MSJ_STAMP(srow, 12);  // next range known, its loads issued, the old range's prefix in hand
        static_assert(kBatch == 2, "the emission below is written for two tiles per wave and range");
        __builtin_amdgcn_s_setprio(MSJ_PRIO_EMIT);
        const uint32_t slot0 = ring * kBatch;
        EmitU e0 = emit_prepare(al, sh, wave, slot0, rp, count0, timeout);
        emit_stage(sh, e0, wave, slot0, stage, lane, lane64);
        lds_wave_sync();
        MSJ_STAMP(srow, 13);  // tile A staged
        emit_store<kTypes>(al, sh, e0, wave, slot0, stage, lane, lane_p);
        lds_wave_sync();  // the staging slice is reused by the next tile
        MSJ_STAMP(srow, 14);  // tile A stored
        EmitU e1 = emit_prepare(al, sh, wave, slot0 + 1u, rp, count0, timeout);
        emit_stage(sh, e1, wave, slot0 + 1u, stage, lane, lane64);
        lds_wave_sync();
        MSJ_STAMP(srow, 15);  // tile B staged
#if MSJ_EARLY_A
        // the next range's FIRST tile has arrived (requested a compute phase, a barrier and an emission ago): loads return
        // in order, so "at most the five younger loads outstanding" says so whatever the stores in between are doing.  Its
        // second tile is waited for behind the next compute phase (the vmcnt(0) there).
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        touch_block(blk[0]);
#else
        // the bytes requested above (and the first tile's stores) have had a whole staging phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) touch_block(blk[j]);
#endif
        MSJ_STAMP(srow, 11);  // the next range's bytes have arrived
        emit_store<kTypes>(al, sh, e1, wave, slot0 + 1u, stage, lane, lane_p);
        lds_wave_sync();
        __builtin_amdgcn_s_setprio(MSJ_PRIO_COMPUTE);
        // ---- 4. park this iteration's tiles
#pragma unroll
        for (uint32_t j = 0; j < kBatch; j++) {
            const uint32_t slot = slot0 + j;
            const uint32_t t_cur = lo_cur + kWaves * j + wave;
            sh.pend_masks[wave][slot][lane] = make_uint4((uint32_t)now[j].T0, (uint32_t)(now[j].T0 >> 32),
                                                         (uint32_t)now[j].T1, (uint32_t)(now[j].T1 >> 32));
            sh.pend_excl[wave][slot][lane] = now[j].excl;
            if (lane_p == 0)
                *reinterpret_cast<uint4 *>(sh.pend_meta[wave][slot]) =
                    make_uint4(t_cur < ntiles ? t_cur : 0xFFFFFFFFu, now[j].tile_cnt, in_cnt[j], in_state[j]);
        }
#if MSJ_EARLY_A
        lo_cur = lo_nxt;
        lo_nxt = lo_next;  // (what the hand-over brought is the range after the next one)
#else
        lo_cur = lo_next;
#endif
        r++;
        ring = (ring + 1u == kDefer) ? 0u : ring + 1u;
    }
    MSJ_RSTAMP(a.ntiles + 4096u + blockIdx.x, 4, tid == 0);  // last range computed
    // ---- drain: oldest first (the launch's tail: nothing but emission is left for this wave)
    __builtin_amdgcn_s_setprio(MSJ_PRIO_EMIT);
    for (uint32_t step = 0; step < kDefer; step++) {
        lds_wave_sync();
        const uint32_t old_first = uniform32(sh.pend_meta[wave][ring * kBatch][0]);
        if (old_first != 0xFFFFFFFFu) {
            const uint64_t w = uniform64(range_prefix(a, rpre, old_first / kRange, 0ull, timeout));
            timeout = uniform32(timeout);
#pragma unroll
            for (uint32_t j = 0; j < kBatch; j++) {
                const uint32_t slot = ring * kBatch + j;
                EmitU e = emit_prepare(a, sh, wave, slot, w, count0, timeout);
                emit_stage(sh, e, wave, slot, stage, lane, lane64);
                lds_wave_sync();
                emit_store<kTypes>(a, sh, e, wave, slot, stage, lane, lane);
                lds_wave_sync();  // the staging slice is reused by the next tile
            }
        }
        ring = (ring + 1u == kDefer) ? 0u : ring + 1u;
    }
    MSJ_RSTAMP(a.ntiles + 4096u + blockIdx.x, 5, tid == 0);  // drained
    // A wait that expired after this wave's last aggregate was published (in the drain, typically) can no
    // longer travel with an aggregate: flag the result directly.  The resolver zeroes the word when the
    // launch starts and ORs its own verdict in at the end, so the two cannot undo each other.
    if (uniform32(timeout) && lane == 0) atomicOr(&a.carry_out->internal_error, 1u);
}

// ---- finish(): json_structural_indexer.mojo:147-186, by one lane, once every tile of the launch is
//      folded: cs = in-string state at the end, cc = structurals of this launch, ce / cu / cx = sticky
//      unescaped-character, UTF-8 and internal (timeout) errors.
__device__ __forceinline__ void finish_launch(const KernelArgs &a, const uint32_t cs, const uint32_t cc,
                                              const uint32_t ce, const uint32_t cu, const uint32_t cx,
                                              const uint64_t *tile_agg = nullptr, const bool late_poison = false) {
    const msj_carry cin = cin_all(a);
    if (!tile_agg) tile_agg = a.ws + kDescOffset;
    const uint64_t last = ld_desc(&tile_agg[a.ntiles - 1u]);  // last TILE's carries
    const bool do_utf8 = !(a.flags & kFlagNoUtf8);
    msj_carry out;
    const uint64_t n = cin.count + cc;
    out.count = n;
    out.bytes = cin.bytes + a.len;
    out.in_string = cs;
    out.next_is_escaped = (uint32_t)(last >> 58) & 1u;
    out.prev_scalar = (uint32_t)(last >> 57) & 1u;
    out.unescaped_error = (cin.unescaped_error | ce) ? 1u : 0u;
    uint32_t u8e = cin.utf8_error | cu;
    // a multi-byte sequence cut exactly at the end of the last full tile
    if ((a.flags & kFlagFinal) && do_utf8 && (a.len % kTileBytes) == 0 && ((last >> 55) & 1u)) u8e = 1;
    out.utf8_error = u8e ? 1u : 0u;
    out.internal_error = (cin.internal_error | cx) ? 1u : 0u;
    // an index buffer too small for the indices so far (+ the trailer on the stream's last segment): the
    // emission clipped its writes (emit_general).  Sticky, and reported by non-final shards too: the ranks of a
    // sharded stream learn it from each other's reports (msj_shard_global_code)
    const bool clipped = !(a.flags & kFlagNoEmit) && n + ((a.flags & kFlagFinal) ? 3u : 0u) > a.capacity;
    out.capacity_error = (cin.capacity_error | (clipped ? 1u : 0u)) ? 1u : 0u;
    int32_t code = MSJ_SUCCESS;
    if (a.flags & kFlagFinal) {
        if (out.internal_error) {
            code = MSJ_UNEXPECTED_ERROR;
        } else if (cs) {
            code = MSJ_UNCLOSED_STRING;  // :151-155
        } else if (out.unescaped_error) {
            code = MSJ_UNESCAPED_CHARS;  // :157-158
        } else if (out.capacity_error) {
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
    // reserved[0]: the three carry bits this launch STARTED from, with bit 31 set (msj_carry.reserved in
    // include/msj_stage1.h): what a rank of a sharded stream reports as "used" without a copy of its own
    out.reserved[0] = (a.flags & kFlagEchoThrough)
                          ? cin.reserved[0]  // a later segment of a chained shard hands the shard's echo on
                          : 0x80000000u | (cin.in_string & 1u) | ((cin.next_is_escaped & 1u) << 1) | ((cin.prev_scalar & 1u) << 2);
    for (int k = 1; k < 4; k++) out.reserved[k] = 0;
    if (late_poison) {
        // single-pass kernel: workers may still OR a late timeout into internal_error (worker_wave), before
        // or after this store: every other field is stored, this one is ORed
        msj_carry *o = a.carry_out;
        o->count = out.count;
        o->bytes = out.bytes;
        o->in_string = out.in_string;
        o->next_is_escaped = out.next_is_escaped;
        o->prev_scalar = out.prev_scalar;
        o->unescaped_error = out.unescaped_error;
        o->utf8_error = out.utf8_error;
        o->code = out.code;
        o->capacity_error = out.capacity_error;
        for (int k = 0; k < 4; k++) o->reserved[k] = out.reserved[k];
        if (out.internal_error) atomicOr(&o->internal_error, 1u);
    } else {
        *a.carry_out = out;
    }
    if (a.segment) {
        a.segment->byte_base = a.segment_byte_base;
        a.segment->byte_len = a.len;
        a.segment->index_begin = cin.count;
        a.segment->count = cc;
    }
}

// ---- resolver: the four waves of one workgroup turn tile aggregates into tile
// prefixes, in order.  Monoid: a tile's aggregate is (parity p, count c[q], error
// e[q]) for incoming in-string state q; composing left to right gives every tile
// its incoming state and the number of structurals before it.
//
// Wave w owns chunks w, w+4, ... of kResolveChunk = 64*kResolveE tiles.  Lane l
// holds tiles e*64 + l of the chunk (e = 0..kResolveE-1), so every descriptor
// load / store is one fully coalesced 512-byte access.  Per 64-tile sub-block the
// wave precomputes, BEFORE the running state arrives: the parity of the lanes below
// (ballot), the inclusive scan of the counts under sub-block state 0 and of
// c[0]+c[1] (state 1 = difference), and the error masks for both states.  When the
// owner of the previous chunk hands the running state over through LDS, the serial
// section is kResolveE steps of scalar arithmetic; the state is handed on first and
// the prefix words are written afterwards, so the memory latency of four chunks
// overlaps.  If the chunk is not complete yet, the tiles whose predecessors are all
// in still get their prefix (partial progress: needed for deadlock freedom, see
// worker_wave).  finish() (json_structural_indexer.mojo:147-186) runs in the wave
// that owns the last chunk.
struct SubBlock {
    uint32_t scanA;  // inclusive scan of the count under sub-block incoming state 0
    uint32_t scanB;  // inclusive scan of c0 + c1
    uint32_t a, b;   // own terms of the two scans
    uint32_t bl;     // parity of the lanes below me in the sub-block
};

__device__ void resolver(const KernelArgs &a, Shared &sh) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    // the resolver's "tiles" are the workers' ranges (kRange tiles each)
    const uint32_t ntiles = (a.ntiles + kRange - 1u) / kRange;
    const uint64_t *agg = a.ws + kDescOffset + a.ntiles;
    uint64_t *pre = a.ws + kDescOffset + a.ntiles + ntiles;
    const uint32_t nchunks = (ntiles + kResolveChunk - 1) / kResolveChunk;
    if (a.ws_clean && tid < kTicketShards) {
        a.ws_clean[(uint64_t)tid * kTicketStrideWords] = 0ull;  // ticket counters
    }
    if (tid == 0) {
        // sticky from the stream so far; workers OR late timeouts into it (worker_wave), finish() ORs its own
        a.carry_out->internal_error = cin_internal_error(a);
        sh.rs_seq = 0;
        sh.rs_s = cin_in_string(a);
        sh.rs_cnt = 0;
        sh.rs_err = 0;
        sh.rs_u8 = 0;
        sh.rs_poison = 0;
    }
    __syncthreads();
    if (a.flags & kFlagDebugStall) {
        // test hook: ~2 ms in which no range prefix appears, so that the workers' waits expire (with
        // a lowered wait_ticks) and the host's two-pass fallback is exercised
        for (uint32_t i = 0; i < 600u; i++) __builtin_amdgcn_s_sleep(127);
    }
    const uint64_t below = (1ull << lane) - 1ull;
    volatile uint32_t *seq = &sh.rs_seq;
    __builtin_amdgcn_s_setprio(3);  // the serial chain of the whole launch runs here
    for (uint32_t c = wave; c < nchunks; c += kWaves) {
        const uint32_t base = c * kResolveChunk;
        uint64_t d[kResolveE];
#pragma unroll
        for (int e = 0; e < kResolveE; e++) d[e] = 0;
        uint32_t published = 0, force = 0;
        WaitClock clk;  // restarted whenever the chunk makes progress
        MSJ_RSTAMP(a.ntiles + c, 0, lane == 0);
        uint32_t rounds = 0; (void)rounds;
        bool have_state = false, full = false, agg_done = false;
        uint32_t s = 0, cnt = 0, err = 0, u8 = 0, poison = 0;
        uint32_t m = 0;
        SubBlock sb[kResolveE];
        uint32_t par[kResolveE], tot0[kResolveE], totB[kResolveE];
        uint64_t E0[kResolveE], E1[kResolveE], UM[kResolveE], XM[kResolveE];
        for (;;) {
            if (!full) {
                // (re)load what was not there yet; m = leading tiles whose aggregates are in
                m = kResolveChunk;
#pragma unroll
                for (int e = kResolveE - 1; e >= 0; e--) {
                    const uint32_t t = base + (uint32_t)e * 64u + lane;
                    if ((d[e] >> 62) == 0ull) {
                        d[e] = (t < ntiles) ? ld_desc(&agg[t]) : kAgg;  // past the end: identity
                        if (force && (d[e] >> 62) == 0ull) d[e] = kAgg | (1ull << 54);  // gave up
                    }
                    const uint64_t nr = __ballot((d[e] >> 62) == 0ull);
                    if (nr) m = (uint32_t)e * 64u + (uint32_t)__builtin_ctzll(nr);
                }
                full = (m == kResolveChunk);
                rounds++;
                if (full) MSJ_RSTAMP(a.ntiles + c, 1, lane == 0);
            }
            if (full && !agg_done) {
                // everything that does not need the running state, done before it arrives
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    const uint64_t de = d[e];
                    const uint32_t p = (uint32_t)(de >> 61) & 1u;
                    const uint32_t c0 = (uint32_t)de & 0xFFFFu, c1 = (uint32_t)(de >> 16) & 0xFFFFu;
                    const uint32_t e0 = (uint32_t)(de >> 60) & 1u, e1 = (uint32_t)(de >> 59) & 1u;
                    const uint64_t PM = __ballot(p != 0u);
                    const uint32_t bl = (uint32_t)__popcll(PM & below) & 1u;
                    sb[e].bl = bl;
                    sb[e].a = bl ? c1 : c0;
                    sb[e].b = c0 + c1;
                    sb[e].scanA = dpp_add_scan(sb[e].a);
                    sb[e].scanB = dpp_add_scan(sb[e].b);
                    tot0[e] = bcast(sb[e].scanA, 63);
                    totB[e] = bcast(sb[e].scanB, 63);
                    par[e] = (uint32_t)__popcll(PM) & 1u;
                    E0[e] = __ballot((bl ? e1 : e0) != 0u);
                    E1[e] = __ballot((bl ? e0 : e1) != 0u);
                    UM[e] = __ballot(((de >> 56) & 1ull) != 0ull);
                    XM[e] = __ballot(((de >> 54) & 1ull) != 0ull);
                }
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
                MSJ_RSTAMP(a.ntiles + c, 2, lane == 0);
            }
            if (have_state && full) {
                // ---- the serial section of the whole launch: scalar steps only
                uint32_t q[kResolveE], cb[kResolveE], eb[kResolveE], ub[kResolveE];
                uint32_t cs = s, cc = cnt, ce = err, cu = u8, cx = poison;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    q[e] = cs;
                    cb[e] = cc;
                    eb[e] = ce;
                    ub[e] = cu;
                    cc += cs ? (totB[e] - tot0[e]) : tot0[e];
                    ce |= ((cs ? E1[e] : E0[e]) != 0ull) ? 1u : 0u;
                    cu |= (UM[e] != 0ull) ? 1u : 0u;
                    cx |= (XM[e] != 0ull) ? 1u : 0u;
                    cs ^= par[e];
                }
                if (lane == 0) {
                    // hand the running state to the owner of the next chunk first; this
                    // chunk's prefix words are written afterwards
                    sh.rs_s = cs;
                    sh.rs_cnt = cc;
                    sh.rs_err = ce;
                    sh.rs_u8 = cu;
                    sh.rs_poison = cx;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    *seq = c + 1u;
                }
                const uint64_t pz = (uint64_t)(cx ? 1u : 0u) << 54;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    const uint32_t idx_in_chunk = (uint32_t)e * 64u + lane;
                    const uint32_t t = base + idx_in_chunk;
                    if (idx_in_chunk >= published && t < ntiles) {
                        const uint32_t exclA = sb[e].scanA - sb[e].a, exclB = sb[e].scanB - sb[e].b;
                        const uint32_t before = cb[e] + (q[e] ? (exclB - exclA) : exclA);
                        const uint32_t s_in = q[e] ^ sb[e].bl;
                        const uint32_t e_in = eb[e] | ((((q[e] ? E1[e] : E0[e]) & below) != 0ull) ? 1u : 0u);
                        const uint32_t u_in = ub[e] | (((UM[e] & below) != 0ull) ? 1u : 0u);
                        st_desc(&pre[t], kPre | ((uint64_t)s_in << 61) | ((uint64_t)e_in << 60) |
                                             ((uint64_t)u_in << 56) | pz | (uint64_t)before);
                    }
                }
#ifdef MSJ_STAMPS
                if (lane == 0 && a.stamps) { a.stamps[(uint64_t)(a.ntiles + c) * 16 + 3] = __builtin_amdgcn_s_memrealtime(); a.stamps[(uint64_t)(a.ntiles + c) * 16 + 4] = rounds; a.stamps[(uint64_t)(a.ntiles + c) * 16 + 5] = published; }
#endif
                if (c + 1u == nchunks && lane == 0) finish_launch(a, cs, cc, ce, cu, cx, nullptr, true);
                break;
            }
            if (have_state && m > published) {
                // PARTIAL PROGRESS (chunk not complete): tiles [published, m) have all their
                // predecessors in.  Same arithmetic restricted to the leading m tiles; only
                // ever runs in the chunk at the frontier, while the launch is starved anyway.
                uint32_t cs = s, cc = cnt, ce = err, cu = u8;
                const uint32_t cx = poison;
#pragma unroll
                for (int e = 0; e < kResolveE; e++) {
                    const uint32_t lo_e = (uint32_t)e * 64u;
                    if (lo_e < m) {  // uniform
                        const uint32_t nact = (m - lo_e) < 64u ? (m - lo_e) : 64u;
                        const bool act = lane < nact;
                        const uint64_t de = act ? d[e] : kAgg;
                        const uint32_t p = (uint32_t)(de >> 61) & 1u;
                        const uint32_t c0 = (uint32_t)de & 0xFFFFu, c1 = (uint32_t)(de >> 16) & 0xFFFFu;
                        const uint32_t e0 = (uint32_t)(de >> 60) & 1u, e1 = (uint32_t)(de >> 59) & 1u;
                        const uint64_t PM = __ballot(p != 0u);
                        const uint32_t in_l = cs ^ ((uint32_t)__popcll(PM & below) & 1u);
                        const uint32_t mine = in_l ? c1 : c0;
                        const uint32_t incl = dpp_add_scan(mine);
                        const uint64_t EM = __ballot((in_l ? e1 : e0) != 0u);
                        const uint64_t U = __ballot(((de >> 56) & 1ull) != 0ull);
                        const uint64_t X = __ballot(((de >> 54) & 1ull) != 0ull);
                        const uint32_t t = base + lo_e + lane;
                        if (act && lo_e + lane >= published && t < ntiles) {
                            const uint32_t e_in = ce | (((EM & below) != 0ull) ? 1u : 0u);
                            const uint32_t u_in = cu | (((U & below) != 0ull) ? 1u : 0u);
                            const uint64_t pz = (uint64_t)((cx | (((X & below) != 0ull) ? 1u : 0u) |
                                                            ((uint32_t)(de >> 54) & 1u))) << 54;
                            st_desc(&pre[t], kPre | ((uint64_t)in_l << 61) | ((uint64_t)e_in << 60) |
                                                 ((uint64_t)u_in << 56) | pz | (uint64_t)(cc + incl - mine));
                            }
                        cc += bcast(incl, 63);
                        ce |= EM ? 1u : 0u;
                        cu |= U ? 1u : 0u;
                        cs ^= (uint32_t)__popcll(PM) & 1u;
                    }
                }
                published = m;
                clk.restart();
            }
            if (clk.expired(a.wait_ticks)) force = 1;  // next round substitutes poisoned identities
            // full but no state yet: spin on the LDS word only (no global traffic)
            if (!full) __builtin_amdgcn_s_sleep(1);
        }
    }
}


// ---- the two-pass path: three plain kernels in which no workgroup ever waits for another one --------
// What the host falls back to when a single-pass launch expires a wait (internal_error, api.cpp), and
// what MSJ_FLAG_TWO_PASS asks for directly: slower (the input is read twice, the scan pass is one wave)
// but there is nothing in it that could time out, so a valid document never comes back as code 24.
//   pass 1  one wave per tile: the tile aggregate (tp[t]); a tile whose carries the 64 bytes in front of it
//           do not decide (>= 63 backslashes) is left to the scan pass (status 3)
//   scan    one wave walks the aggregates in order, 64 tiles per step (the resolver's arithmetic); a left-over
//           tile is computed right there with the exact carries of its predecessor; writes per tile the state
//           and the count in front of it (tp[ntiles + t]), then finish()
//   pass 2  one wave per tile: the masks again, its state and position from the scan, emission as in the
//           single-pass kernel (through one parked slot)
constexpr uint64_t kTpUnresolved = 3ull << 62;

__global__ __launch_bounds__(kThreads) void twopass_summary_kernel(const KernelArgs a) {
    const uint32_t lane = threadIdx.x & 63u, wave = uniform32(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x * kWaves + wave;
    if (tile >= a.ntiles) return;
    const uint32_t lane64 = lane * 64u;
    const uint32_t lane_off[4] = {lane64, lane64 + 16u, lane64 + 32u, lane64 + 48u};
    const uint32_t carry0 = uniform32(cin_carry0(a));
    Block blk;
    load_block<false>(a, tile, lane_off, lane, blk);
    uint64_t *tp = a.tp;
    if (!window_carries(a, tile, blk.wb, carry0).resolved) {
        if (lane == 0) tp[tile] = kTpUnresolved;
        return;
    }
    uint32_t timeout = 0;
    uint64_t agg_word;
    (void)compute_tile(a, tile, lane, blk, carry0, timeout, agg_word);
    if (lane == 0) tp[tile] = agg_word;
}

__global__ __launch_bounds__(64) void twopass_scan_kernel(const KernelArgs a) {
    const uint32_t lane = threadIdx.x;
    const uint32_t ntiles = a.ntiles;
    uint64_t *agg = a.tp, *pre = a.tp + ntiles;
    const uint32_t lane64 = lane * 64u;
    const uint32_t lane_off[4] = {lane64, lane64 + 16u, lane64 + 32u, lane64 + 48u};
    const uint32_t carry0 = uniform32(cin_carry0(a));
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t cs = uniform32(cin_in_string(a)), cc = 0, ce = 0, cu = 0;
    uint32_t e_prev = carry0 & 1u, ps_prev = (carry0 >> 1) & 1u;  // carries out of the tile before
    for (uint32_t base = 0; base < ntiles; base += 64u) {  // uniform
        const uint32_t t = base + lane;
        uint64_t w = t < ntiles ? agg[t] : kAgg;  // past the end: identity
        const uint64_t unres = __ballot((w >> 62) == 3ull);
        if (unres == 0ull) {
            const uint32_t p = (uint32_t)(w >> 61) & 1u;
            const uint32_t c0 = (uint32_t)w & 0xFFFFu, c1 = (uint32_t)(w >> 16) & 0xFFFFu;
            const uint32_t e0 = (uint32_t)(w >> 60) & 1u, e1 = (uint32_t)(w >> 59) & 1u;
            const uint64_t PM = __ballot(p != 0u);
            const uint32_t in_l = cs ^ ((uint32_t)__popcll(PM & below) & 1u);
            const uint32_t mine = in_l ? c1 : c0;
            const uint32_t incl = dpp_add_scan(mine);
            const uint64_t EM = __ballot((in_l ? e1 : e0) != 0u);
            const uint64_t U = __ballot(((w >> 56) & 1ull) != 0ull);
            if (t < ntiles) pre[t] = kPre | ((uint64_t)in_l << 61) | (uint64_t)(cc + incl - mine);
            cc += bcast(incl, 63);
            ce |= EM ? 1u : 0u;
            cu |= U ? 1u : 0u;
            cs ^= (uint32_t)__popcll(PM) & 1u;
            const uint32_t last = (ntiles - base < 64u ? ntiles - base : 64u) - 1u;  // last tile of the group
            const uint32_t wh = bcast((uint32_t)(w >> 32), (int)last);
            e_prev = (wh >> 26) & 1u;
            ps_prev = (wh >> 25) & 1u;
        } else {
            // a tile of this group was left to the scan: walk the group tile by tile
            const uint32_t cnt = ntiles - base < 64u ? ntiles - base : 64u;
            for (uint32_t k = 0; k < cnt; k++) {  // uniform
                uint64_t wk = u64(bcast((uint32_t)w, (int)k), bcast((uint32_t)(w >> 32), (int)k));
                uint32_t exact = 0;
                if ((wk >> 62) == 3ull) {
                    Block blk;
                    load_block<false>(a, base + k, lane_off, lane, blk);
                    uint32_t timeout = 0;
                    exact = 1u | (e_prev << 1) | (ps_prev << 2);
                    (void)compute_tile(a, base + k, lane, blk, carry0, timeout, wk, exact);
                    if (lane == 0) agg[base + k] = wk;  // the launch's last tile: finish() reads its carries
                }
                const uint32_t c = cs ? (uint32_t)(wk >> 16) & 0xFFFFu : (uint32_t)wk & 0xFFFFu;
                if (lane == 0) pre[base + k] = kPre | ((uint64_t)cs << 61) | ((uint64_t)exact << 50) | (uint64_t)cc;
                cc += c;
                ce |= (uint32_t)(wk >> (cs ? 59 : 60)) & 1u;
                cu |= (uint32_t)(wk >> 56) & 1u;
                cs ^= (uint32_t)(wk >> 61) & 1u;
                e_prev = (uint32_t)(wk >> 58) & 1u;
                ps_prev = (uint32_t)(wk >> 57) & 1u;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) finish_launch(a, cs, cc, ce, cu, 0u, agg);
}

__global__ __launch_bounds__(kThreads) void twopass_emit_kernel(const KernelArgs a) {
    __shared__ Shared sh;
    const uint32_t lane = threadIdx.x & 63u, wave = uniform32(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x * kWaves + wave;
    if (tile >= a.ntiles) return;
    const uint32_t lane64 = lane * 64u;
    const uint32_t lane_off[4] = {lane64, lane64 + 16u, lane64 + 32u, lane64 + 48u};
    const uint32_t carry0 = uniform32(cin_carry0(a));
    const uint64_t count0 = uniform64(cin_count(a));
    Block blk;
    load_block<false>(a, tile, lane_off, lane, blk);
    const uint64_t pw = uniform64(a.tp[a.ntiles + tile]);  // state and count in front of the tile (scan pass)
    uint32_t timeout = 0;
    uint64_t agg_word;
    const Pending r = compute_tile(a, tile, lane, blk, carry0, timeout, agg_word, (uint32_t)(pw >> 50) & 7u);
    // through one parked slot, like a tile of the single-pass kernel whose range starts at the tile itself
    const uint32_t s_in = (uint32_t)(pw >> 61) & 1u;
    sh.pend_masks[wave][0][lane] = make_uint4((uint32_t)r.T0, (uint32_t)(r.T0 >> 32), (uint32_t)r.T1, (uint32_t)(r.T1 >> 32));
    sh.pend_excl[wave][0][lane] = r.excl;
    if (lane == 0) *reinterpret_cast<uint4 *>(sh.pend_meta[wave][0]) = make_uint4(tile, r.tile_cnt, 0u, s_in | (s_in << 1));
    lds_wave_sync();
    uint32_t *stage = sh.stage[wave] + kStageSlack;
    EmitU e = emit_prepare(a, sh, wave, 0, kPre | (uint64_t)(uint32_t)pw, count0, timeout);
    emit_stage(sh, e, wave, 0, stage, lane, lane64);
    lds_wave_sync();
    emit_store(a, sh, e, wave, 0, stage, lane, lane);
}

template <bool kTypes>
__device__ __forceinline__ void stage1_body(const KernelArgs &a) {
    __shared__ Shared sh;
    const uint32_t tid = threadIdx.x;
    // ONE atomic per workgroup at start-up gives it its job.  A workgroup belongs to a ticket shard
    // by its blockIdx -- blocks 8k .. 8k+7 (one per XCD) share a shard, so every shard has members on
    // every XCD and a starved XCD (other kernels on the GPU) leaves no shard unserved -- and draws
    // from that shard's counter.  The first draw from shard 0 makes the resolver: arrival order, not
    // blockIdx (with other kernels on the GPU block 0 is not necessarily resident first; measured:
    // four processes sharing one GPU time out with a static resolver), and it is running, so the
    // workers that wait on its output can always make progress.  Every other draw is a range of
    // tiles: ranges are DRAWN, the first one too (handing range w to worker w up front would give
    // ranges to workgroups that are not resident yet, and everybody waits for them).
    MSJ_RSTAMP(a.ntiles + 4096u + blockIdx.x, 0, tid == 0);  // workgroup started
    if (tid == 0) {
        const uint32_t workers = gridDim.x - 1u;
        const uint32_t shards = workers < kTicketShards ? workers : kTicketShards;
        const uint32_t shard = (gridDim.x >= 8u * kTicketShards ? blockIdx.x >> 3 : blockIdx.x) % shards;
        const uint32_t k = atomicAdd(reinterpret_cast<unsigned int *>(a.ws + (uint64_t)shard * kTicketStrideWords), 1u);
        sh.role = (shard == 0u && k == 0u) ? 0u : 1u;
        sh.shard = shard;
        sh.first_lo = ticket_range(k, shard, shards) * kRange;
        if (MSJ_EARLY_A && sh.role != 0u) {
            const uint32_t k2 = atomicAdd(reinterpret_cast<unsigned int *>(a.ws + (uint64_t)shard * kTicketStrideWords), 1u);
            sh.second_lo = ticket_range(k2, shard, shards) * kRange;
        }
        sh.handoff = 0ull;
    }
    __syncthreads();
    if (sh.role == 0u) {
        resolver(a, sh);
        return;
    }
    // the wave index is wave-uniform: say so, or every tile-derived value and branch is vector code
    worker_wave<kTypes>(a, sh, tid & 63u, uniform32(tid >> 6));
}
__global__ __launch_bounds__(kThreads, 4) void stage1_kernel(const KernelArgs a) { stage1_body<false>(a); }
// PROTOTYPE (round 5, kFlagEmitTypes): the same kernel, the type bytes written beside the indices
__global__ __launch_bounds__(kThreads, 4) void stage1_types_kernel(const KernelArgs a) { stage1_body<true>(a); }

}  // namespace msj

extern "C" int msj_launch_stage1(const msj::KernelArgs *args, void *stream, uint32_t grid) {
    const msj::KernelArgs a = *args;
    // persistent workgroups of kWaves worker waves, plus the resolver workgroup
    const uint32_t need = (a.ntiles + msj::kWaves - 1u) / msj::kWaves + 1u;
    const uint32_t g = (grid == 0 || grid > need) ? need : grid;
    if (a.flags & msj::kFlagEmitTypes)
        hipLaunchKernelGGL(msj::stage1_types_kernel, dim3(g < 2u ? 2u : g), dim3(msj::kThreads), 0, static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(msj::stage1_kernel, dim3(g < 2u ? 2u : g), dim3(msj::kThreads), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

// The two-pass path for one segment: summary, scan, emission (a.tp: 2 * ntiles words, need not be zeroed).
extern "C" int msj_launch_stage1_twopass(const msj::KernelArgs *args, void *stream) {
    const msj::KernelArgs a = *args;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t wgs = (a.ntiles + msj::kWaves - 1u) / msj::kWaves;
    hipLaunchKernelGGL(msj::twopass_summary_kernel, dim3(wgs), dim3(msj::kThreads), 0, s, a);
    hipLaunchKernelGGL(msj::twopass_scan_kernel, dim3(1), dim3(64), 0, s, a);
    if (!(a.flags & msj::kFlagNoEmit)) hipLaunchKernelGGL(msj::twopass_emit_kernel, dim3(wgs), dim3(msj::kThreads), 0, s, a);
    return (int)hipGetLastError();
}

extern "C" int msj_stage1_occupancy(int *blocks_per_cu) {
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, msj::stage1_kernel,
                                                            msj::kThreads, 0);
}
