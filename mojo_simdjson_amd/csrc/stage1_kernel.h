// stage1_kernel.h -- launch interface between the C-ABI host code (api.cpp) and
// the HIP kernel (stage1_kernel.hip).
#pragma once
#include <stdint.h>

#include "../../include/msj_stage1.h"

namespace msj {

constexpr int kWaves = 4;                          // wave64 per workgroup (each an independent worker)
constexpr int kThreads = kWaves * 64;
constexpr uint32_t kTileBytes = 64u * 64u;         // 4 KiB of input per wave: 64 lanes x one 64-byte block
// Workspace header: kTicketShards range-ticket counters, one per 4 KiB (a single word sustains
// only ~80 returning atomics per microsecond chip-wide; shard c hands out ranges c, c + S, ...;
// shard 0's first draw makes the resolver); the descriptor arrays follow.
constexpr uint32_t kTicketShards = 8;
constexpr uint32_t kTicketStrideWords = 512;       // in 8-byte words
constexpr uint32_t kDescOffset = kTicketShards * kTicketStrideWords;  // ws[kDescOffset..] = agg[], ragg[], rpre[]
constexpr uint32_t kStageWords = 1024;             // per-wave LDS staging of indices (4 KiB) per round
constexpr uint32_t kStageSlack = 64;               // words in front of and behind a staging slice: where the lane that straddles
                                                   // the border between the two rounds of a dense tile writes the indices of the other round
constexpr uint32_t kBatch = 2;                     // tiles per wave per ticket range
constexpr uint32_t kDefer = 2;                     // a tile is emitted kDefer ranges after it was computed
constexpr uint32_t kRange = kWaves * kBatch;       // tiles per ticket range = per range aggregate
constexpr uint32_t kPendSlots = kDefer * kBatch;   // parked tiles per wave
constexpr int kResolveE = 4;                       // tiles folded per resolver lane
constexpr uint32_t kResolveChunk = 64u * kResolveE; // tiles per resolver chunk (one wave, one round)
constexpr uint32_t kWaitTicksDefault = 200000000u; // bound of every inter-workgroup wait: 2 s of s_memrealtime (100 MHz)
// largest segment one launch indexes with uint32 offsets (multiple of the tile)
constexpr uint64_t kSegmentBytes = 0xFFFF0000ull;

// kernel-internal flags (KernelArgs.flags)
constexpr uint32_t kFlagStrictUtf8 = 1u;   // == MSJ_FLAG_STRICT_UTF8
constexpr uint32_t kFlagNoUtf8 = 2u;       // == MSJ_FLAG_NO_UTF8
constexpr uint32_t kFlagFinal = 4u;        // last segment of the stream: trailer + return code
constexpr uint32_t kFlagHasPrefix = 8u;    // buf[-64..0) holds the preceding stream bytes
constexpr uint32_t kFlagNoEmit = 16u;      // summary pass: no index writes
constexpr uint32_t kFlagDebugStall = 32u;  // test hook (MSJ_FLAG_DEBUG_STALL): the resolver idles ~2 ms before it starts
constexpr uint32_t kFlagCarryByValue = 64u; // the state at the launch's first byte is KernelArgs.carry_bits (count, bytes and the
                                            // sticky flags zero), not *carry_in: the start of a shard whose carries the host
                                            // knows (msj_stage1_shard_device_cv) -- nothing has to be copied to the device first
constexpr uint32_t kFlagEchoThrough = 128u; // a later segment of a chained shard: carry_out.reserved[0] = carry_in.reserved[0] (the
                                            // echo of the carry the SHARD started from travels down the chain)
constexpr uint32_t kFlagEmitTypes = 256u;   // PROTOTYPE (msj_stage1_types_device): the type byte of every structural beside its index
                                            // (what JsonIterator.advance dereferences, generic/stage2/json_iterator.mojo:256-262) from
                                            // the same emission -- a second instantiation of the kernel, the product's is untouched
constexpr uint32_t kFlagSkipShift = 24u;   // bits 24..27 == MSJ_FLAG_SKIP(n): the first n < 16 bytes of the launch read as blanks

struct KernelArgs {
    const uint8_t *buf;       // segment base, 16-byte aligned, device memory
    uint64_t len;             // segment length in bytes, 0 < len <= kSegmentBytes
    uint32_t *idx;            // index array base (absolute positions: carry_in->count + ...)
    uint64_t capacity;        // index array capacity in elements
    uint64_t *ws;             // zeroed workspace: ticket + one descriptor per tile
    uint64_t *ws_clean;       // optional: the OTHER workspace buffer, dirtied by the previous launch with the
                              // same ntiles; this launch zeroes it word for word (no memset between launches)
    const msj_carry *carry_in; // (ignored with kFlagCarryByValue)
    msj_carry *carry_out;
    msj_segment *segment;     // optional: segment-table entry to fill
    uint64_t segment_byte_base;
    uint64_t trailer_len;     // value of the two `len` trailer words (FINAL only)
    uint32_t ntiles;
    uint32_t flags;
    uint32_t index_bias;      // added to every index: byte offset of this launch's buffer inside the document
                              // (host-pointer pipeline: one launch per uploaded chunk, offsets stay absolute)
    uint32_t wait_ticks;      // bound of every wait in the kernel, in s_memrealtime ticks (10 ns); expiry poisons the launch
    uint32_t carry_bits;      // kFlagCarryByValue: bit 0 in_string, bit 1 next_is_escaped, bit 2 prev_scalar
    uint32_t reserved0;
    uint64_t *stamps;         // diagnostic builds only (-DMSJ_STAMPS): 16 words per tile, else null
    uint64_t *tp;             // two-pass path only: 2 * ntiles words (tile aggregates, tile prefixes)
    uint8_t *types;           // kFlagEmitTypes (prototype, round 5): types[k] = buf[idx[k]], written beside idx[k]
};

// ws: ticket shards (header), then per-tile carry words, range aggregates, range prefixes
inline uint64_t workspace_words(uint32_t ntiles) {
    const uint64_t nranges = (ntiles + kRange - 1u) / kRange;
    return (uint64_t)kDescOffset + ntiles + 2ull * nranges;
}

}  // namespace msj

// grid = number of persistent workgroups (0 = one per tile); clamped to ntiles + 1.
extern "C" int msj_launch_stage1(const msj::KernelArgs *args, void *stream, uint32_t grid);
extern "C" int msj_launch_stage1_twopass(const msj::KernelArgs *args, void *stream);
extern "C" int msj_stage1_occupancy(int *blocks_per_cu);
