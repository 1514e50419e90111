// lane_math.h -- per-lane math of the stage-1 kernel.
//
// One GPU lane owns one 64-byte block of input, i.e. exactly the unit the
// reference processes per JsonScanner.next call
// (generic/stage1/json_structural_indexer.mojo:116-126), and every mask below
// is a uint64 with bit i <-> byte i of the block, LSB = lowest address: the
// same convention as the reference's pack_bits masks (stuff.mojo:6-9).
//
// Instead of per-character compares (the reference's eq[]/classify,
// stuff.mojo:6-9, haswell.mojo:22-74) the block is bit-transposed into its
// eight bit-planes (b0..b7, 64 bits each) with a SWAR butterfly + v_perm byte
// gather (~3 VALU ops per input byte), after which every character class and
// the whole UTF-8 validator are boolean functions of those planes evaluated 64
// bytes at a time.
//
// This header is plain C++ so that tests/ can compile it with g++ and check
// each function on the CPU (tests/test_lane_math.py, no GPU needed); on the device
// msj_perm() is the v_perm_b32 instruction.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MSJ_HD __host__ __device__ __forceinline__
#else
#define MSJ_HD static inline
#endif

namespace msj {

// v_perm_b32: result byte i = byte sel[i] of the 8-byte value {s0:s1} (s1 = low
// dword); selector values 0..7 only are used here.
MSJ_HD uint32_t perm(uint32_t s0, uint32_t s1, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    uint64_t d = ((uint64_t)s0 << 32) | s1;
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t s = (sel >> (8 * i)) & 0xFF;
        r |= (uint32_t)((d >> (8 * (s & 7))) & 0xFF) << (8 * i);
    }
    return r;
#endif
}

MSJ_HD uint64_t u64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Any boolean function of three masks in ONE operation (v_bitop3_b32 on gfx950).  TT is
// the truth table: bit (4a + 2b + c) of TT is f(a, b, c).  MSJ_TT builds it from an
// expression in a, b, c at compile time.
template <uint32_t TT>
MSJ_HD uint32_t lut3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_bitop3_b32(a, b, c, TT);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < 8; i++)
        if ((TT >> i) & 1u) r |= ((i & 4u) ? a : ~a) & ((i & 2u) ? b : ~b) & ((i & 1u) ? c : ~c);
    return r;
#endif
}
template <uint32_t TT>
MSJ_HD uint64_t lut3(uint64_t a, uint64_t b, uint64_t c) {
    return u64(lut3<TT>((uint32_t)a, (uint32_t)b, (uint32_t)c),
               lut3<TT>((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)));
}
// Truth tables are written as expressions in TA, TB, TC: evaluating the expression on the
// three 8-bit constants below yields bit (4a + 2b + c) = f(a, b, c) for all eight inputs.
constexpr uint32_t TA = 0xF0u, TB = 0xCCu, TC = 0xAAu;
#define MSJ_TT(expr) ((uint32_t)(expr) & 0xFFu)

// 4x4 byte transpose: o[k] = bytes k of (a,b,c,d), a in the low byte.
MSJ_HD void byte_transpose4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t &o0,
                            uint32_t &o1, uint32_t &o2, uint32_t &o3) {
    uint32_t ab_lo = perm(b, a, 0x05010400u);  // a0 b0 a1 b1
    uint32_t ab_hi = perm(b, a, 0x07030602u);  // a2 b2 a3 b3
    uint32_t cd_lo = perm(d, c, 0x05010400u);
    uint32_t cd_hi = perm(d, c, 0x07030602u);
    o0 = perm(cd_lo, ab_lo, 0x05040100u);  // a0 b0 c0 d0
    o1 = perm(cd_lo, ab_lo, 0x07060302u);  // a1 b1 c1 d1
    o2 = perm(cd_hi, ab_hi, 0x05040100u);
    o3 = perm(cd_hi, ab_hi, 0x07060302u);
}

// Exchange, between two registers, the bits whose in-register position has bit
// log2(SH) set (in x) / clear (in y): afterwards x holds, for every position with
// that bit clear, its own bit, and at the position + SH the bit y had at the
// position; y symmetrically.  Two shift + two bit-select operations per pair.
template <uint32_t SH, uint32_t M0>
MSJ_HD void delta_swap(uint32_t &x, uint32_t &y) {
    // measured on gfx950 (scripts/ubench/valu_ops.hip): v_bitop3 / v_lshrrev / v_add run at
    // full rate, v_bfi / v_lshlrev at half rate -> selects are written as 3-input LUTs and
    // the shift by one as an add
    uint32_t yl;
#if defined(__HIP_DEVICE_COMPILE__)
    if (SH == 1)
        asm("v_add_u32_e32 %0, %1, %1" : "=v"(yl) : "v"(y));  // y << 1 at full rate
    else
#endif
        yl = y << SH;
    const uint32_t xr = x >> SH;
    const uint32_t m0 = M0;
    const uint32_t nx = lut3<MSJ_TT((TA & TB) | (~TA & TC))>(m0, x, yl);
    const uint32_t ny = lut3<MSJ_TT((TA & TB) | (~TA & TC))>(m0, xr, y);
    x = nx;
    y = ny;
}

// 64 bytes (16 little-endian dwords) -> 8 bit-planes: plane k bit i = bit k of byte i.
//
// Address view: input bit (dword d = d3 d2 d1 d0, byte j = j1 j0, bit k = k2 k1 k0) must
// end up in plane (k2 k1 k0), half d3, at bit position (d2 d1 d0 j1 j0).  Step 1 moves
// (d2 d1) into the byte address with 4x4 byte transposes (v_perm, 2 ops per register);
// step 2 exchanges the three bit-address bits k2, k1, k0 with the register-index bits
// d0, j1, j0 by three rounds of register-pair delta swaps (2 ops per register each):
// 128 operations for 64 bytes.
MSJ_HD void bitplanes(const uint32_t x[16], uint64_t p[8]) {
    // y[h][n][d0]: byte m of it = byte n of x[8h + 2m + d0]
    uint32_t y[2][4][2];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int d0 = 0; d0 < 2; d0++)
            byte_transpose4(x[8 * h + 0 + d0], x[8 * h + 2 + d0], x[8 * h + 4 + d0], x[8 * h + 6 + d0],
                            y[h][0][d0], y[h][1][d0], y[h][2][d0], y[h][3][d0]);
    // k2 <-> d0  (distance 4): afterwards index [h][n][k2]
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int n = 0; n < 4; n++) delta_swap<4, 0x0F0F0F0Fu>(y[h][n][0], y[h][n][1]);
    // k1 <-> j1  (distance 2): n = 2*j1 + j0; afterwards the j1 slot holds k1
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int j0 = 0; j0 < 2; j0++)
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) delta_swap<2, 0x33333333u>(y[h][j0][k2], y[h][2 + j0][k2]);
    // k0 <-> j0  (distance 1): afterwards the j0 slot holds k0
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k1 = 0; k1 < 2; k1++)
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) delta_swap<1, 0x55555555u>(y[h][2 * k1][k2], y[h][2 * k1 + 1][k2]);
    // y[h][2*k1 + k0][k2] = plane (k2 k1 k0), bytes 32h .. 32h+31
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int k2 = (k >> 2) & 1, k1 = (k >> 1) & 1, k0 = k & 1;
        p[k] = u64(y[0][2 * k1 + k0][k2], y[1][2 * k1 + k0][k2]);
    }
}

// Character classes of one block, as masks.
struct Classes {
    uint64_t backslash;  // 0x5C            (stuff.mojo:6-9 eq["\\"])
    uint64_t quote_chr;  // 0x22 raw, before escape resolution (eq['"'])
    uint64_t op;         // {0C,1A,2C,3A,5B,5D,7B,7D}  (haswell.mojo:44-69, effective set)
    uint64_t ws;         // {09,0A,0D,20}   (haswell.mojo:23-65)
    uint64_t ctrl;       // byte <= 0x1F    (json_structural_indexer.mojo:135)
};

// valid: bit i set iff byte i is inside the input; bytes past the end behave as
// the 0x20 padding the reference copies into its last block
// (json_structural_indexer.mojo:103-107).  Pass ~0 for a block that is entirely input.
//
// Every class is a product of a pattern of the low three bits (b2 b1 b0), of b3 and of the
// high nibble; written as ~20 three-input operations per 32 bytes.
MSJ_HD Classes classify(const uint64_t p[8], uint64_t valid) {
    const uint64_t b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    const uint64_t b4 = p[4], b5 = p[5], b6 = p[6], b7 = p[7];
    // patterns of (b2 b1 b0)
    const uint64_t g100 = lut3<MSJ_TT(TA & ~TB & ~TC)>(b2, b1, b0);           // xC
    const uint64_t g010 = lut3<MSJ_TT(~TA & TB & ~TC)>(b2, b1, b0);           // xA, x2
    const uint64_t g011_101 = lut3<MSJ_TT((TA ^ TB) & TC)>(b2, b1, b0);       // xB, xD
    const uint64_t g001_010_101 = lut3<MSJ_TT((~TA & (TB ^ TC)) | (TA & ~TB & TC))>(b2, b1, b0);  // x9 xA xD
    const uint64_t g000 = lut3<MSJ_TT(~TA & ~TB & ~TC)>(b2, b1, b0);          // x0
    Classes c;
    // operators {0C,2C} {1A,3A} {5B,5D,7B,7D}: b3 = 1, b7 = 0, b5 free
    {
        const uint64_t x = lut3<MSJ_TT((TC & TB) | (~TC & TA))>(g010, g011_101, b6);    // (xA & ~b6) | (xB/xD & b6)
        const uint64_t y = lut3<MSJ_TT(TA & ~TB & ~TC)>(g100, b6, b4);        // xC with hi nibble 0 / 2
        const uint64_t z = lut3<MSJ_TT((TA & TB) | TC)>(x, b4, y);
        c.op = lut3<MSJ_TT(TA & TB & ~TC)>(z, b3, b7);
    }
    // whitespace 09 0A 0D (hi nibble 0, b3 = 1) and 20 (hi nibble 2, low nibble 0)
    {
        const uint64_t w1 = lut3<MSJ_TT(~TA & TB & TC)>(b5, b3, g001_010_101);
        const uint64_t w2 = lut3<MSJ_TT(TA & ~TB & TC)>(b5, b3, g000);
        const uint64_t w3 = lut3<MSJ_TT((TA | TB) & ~TC)>(w1, w2, b4);
        c.ws = lut3<MSJ_TT(TA & ~TB & ~TC)>(w3, b7, b6);
    }
    // backslash 5C = 0101 1100, quote 22 = 0010 0010, control 000x xxxx
    {
        const uint64_t h5 = lut3<MSJ_TT(~TA & TB & ~TC)>(b7, b6, b5);
        const uint64_t h5b = lut3<MSJ_TT(TA & TB & TC)>(h5, b4, b3);
        c.backslash = h5b & g100;
        const uint64_t h2 = lut3<MSJ_TT(~TA & ~TB & TC)>(b7, b6, b5);
        const uint64_t h2b = lut3<MSJ_TT(TA & ~TB & ~TC)>(h2, b4, b3);
        c.quote_chr = h2b & g010;
        c.ctrl = lut3<MSJ_TT(~TA & ~TB & ~TC)>(b7, b6, b5);
    }
    if (valid != ~0ull) {
        c.backslash &= valid;
        c.quote_chr &= valid;
        c.op &= valid;
        c.ws = (c.ws & valid) | ~valid;
        c.ctrl &= valid;
    }
    return c;
}

// Character classes the token-span kernel reads from a block (tokens_kernel.hip, SURVEY.md section 8 rows
// f2 / f4), from the same bit-planes:
//   digit      30..39                        (is_integer, include/generic/number_parsing.mojo:22-30)
//   sow        09 0A 0D 20 , : [ ] { }       (structural_or_whitespace, internal/jsoncharutils_tables.mojo:5-16)
//   backslash  5C
//   blank      09 0A 0D 20
// No `valid` mask: the kernel presents bytes past the end of the buffer as 0x20.
struct SpanClasses {
    uint64_t digit, sow, backslash, blank;
};
MSJ_HD SpanClasses span_classes(const uint64_t p[8]) {
    const uint64_t b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    const uint64_t b4 = p[4], b5 = p[5], b6 = p[6], b7 = p[7];
    const uint64_t g100 = lut3<MSJ_TT(TA & ~TB & ~TC)>(b2, b1, b0);      // xC
    const uint64_t g010 = lut3<MSJ_TT(~TA & TB & ~TC)>(b2, b1, b0);      // xA
    const uint64_t g011_101 = lut3<MSJ_TT((TA ^ TB) & TC)>(b2, b1, b0);  // xB, xD
    const uint64_t g001_010_101 = lut3<MSJ_TT((~TA & (TB ^ TC)) | (TA & ~TB & TC))>(b2, b1, b0);  // x9 xA xD
    const uint64_t g000 = lut3<MSJ_TT(~TA & ~TB & ~TC)>(b2, b1, b0);     // x0
    SpanClasses c;
    {   // 09 0A 0D: high nibble 0, b3 = 1;  20: high nibble 2, low nibble 0
        const uint64_t w1 = lut3<MSJ_TT(~TA & TB & TC)>(b5, b3, g001_010_101);
        const uint64_t w2 = lut3<MSJ_TT(TA & ~TB & TC)>(b5, b3, g000);
        const uint64_t w3 = lut3<MSJ_TT((TA | TB) & ~TC)>(w1, w2, b4);
        c.blank = lut3<MSJ_TT(TA & ~TB & ~TC)>(w3, b7, b6);
    }
    {   // 2C, 3A, 5B 5D 7B 7D: b3 = 1, b7 = 0; (b6 = 0: b5 = 1, b4 picks xC / xA) (b6 = 1: b4 = 1, xB / xD, b5 free)
        const uint64_t x = lut3<MSJ_TT((TC & TB) | (~TC & TA))>(g010, g011_101, b6);  // (xA & ~b6) | (xB/xD & b6)
        const uint64_t y = lut3<MSJ_TT(TA & ~TB & ~TC)>(g100, b6, b4);                // xC, b6 = 0, b4 = 0
        const uint64_t z = lut3<MSJ_TT((TA & TB) | TC)>(x, b4, y);
        const uint64_t op = lut3<MSJ_TT(TA & TB & ~TC)>(z, b3, b7);                   // 0C 2C 1A 3A 5B 5D 7B 7D
        const uint64_t h = lut3<MSJ_TT(TA & (TB | TC))>(op, b5, b6);                  // drop 0C and 1A
        c.sow = h | c.blank;
    }
    {
        const uint64_t h5 = lut3<MSJ_TT(~TA & TB & ~TC)>(b7, b6, b5);
        const uint64_t h5b = lut3<MSJ_TT(TA & TB & TC)>(h5, b4, b3);
        c.backslash = h5b & g100;
    }
    {   // 3x with x <= 9: b3 = 0, or b2 = b1 = 0
        const uint64_t h3 = lut3<MSJ_TT(~TA & ~TB & TC)>(b7, b6, b5);
        const uint64_t lo = lut3<MSJ_TT(~TA | (~TB & ~TC))>(b3, b2, b1);
        c.digit = lut3<MSJ_TT(TA & TB & TC)>(h3, b4, lo);
    }
    return c;
}

// Inclusive prefix XOR over 64 bits (stuff.mojo:21-28 prefix_xor, here 6
// doubling steps instead of 64 popcounts).
MSJ_HD uint64_t prefix_xor(uint64_t x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    lo ^= lo << 1;  hi ^= hi << 1;
    lo ^= lo << 2;  hi ^= hi << 2;
    lo ^= lo << 4;  hi ^= hi << 4;
    lo ^= lo << 8;  hi ^= hi << 8;
    lo ^= lo << 16; hi ^= hi << 16;
    hi ^= (uint32_t)(-(int32_t)(lo >> 31));
    return u64(lo, hi);
}

// JsonEscapeScanner.next (json_escape_scanner.mojo:18-45) for one block with a
// known carry-in; returns the `escaped` mask, *escape_out = next_is_escaped.
MSJ_HD uint64_t escaped_mask(uint64_t backslash, uint32_t next_is_escaped, uint32_t *escape_out) {
    const uint64_t ODD = 0xAAAAAAAAAAAAAAAAull;
    const uint64_t nie = next_is_escaped;
    const uint64_t pe = backslash & ~nie;
    const uint64_t t = (((pe << 1) | ODD) - pe) ^ ODD;
    const uint64_t escaped = t ^ (backslash | nie);
    const uint64_t escape = t & backslash;
    *escape_out = (uint32_t)(escape >> 63);
    return escaped;
}

// Length of the run of set bits at the top of m (bit 63 downwards), 0..64.
MSJ_HD uint32_t top_run(uint64_t m) {
    const uint64_t inv = ~m;
    if (inv == 0) return 64;
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clzll((long long)inv);
#else
    return (uint32_t)__builtin_clzll(inv);
#endif
}

// JsonEscapeScanner.next (json_escape_scanner.mojo:18-45) for next_is_escaped = 0, in the pieces the
// kernel uses (a block that starts escaped is rare: the kernel takes escaped_mask() for those tiles).
//   escape_tt0      tt = t ^ ODD, t = (((bs << 1) | ODD) - bs) ^ ODD  (:39-45); the shift by one is an
//                   add and a funnel shift (a 64-bit shift issues at a quarter of the rate)
//   escape_out0     next_is_escaped of the next block = bit 63 of escape = t & bs  (:30-31)
//   unescaped_quotes0  eq['"'] & ~escaped (json_string_scanner.mojo:58): on a quote byte backslash is 0,
//                   so escaped = t ^ backslash = t there
MSJ_HD uint64_t escape_tt0(uint64_t backslash) {
    const uint64_t ODD = 0xAAAAAAAAAAAAAAAAull;
    const uint32_t lo = (uint32_t)backslash, hi = (uint32_t)(backslash >> 32);
    uint32_t sl;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_add_u32_e32 %0, %1, %1" : "=v"(sl) : "v"(lo));  // lo << 1 at full rate
    const uint32_t sh = __builtin_amdgcn_alignbit(hi, lo, 31);
#else
    sl = lo << 1;
    const uint32_t sh = (hi << 1) | (lo >> 31);
#endif
    return (u64(sl, sh) | ODD) - backslash;
}
MSJ_HD uint32_t escape_out0(uint64_t tt, uint64_t backslash) {
    return lut3<MSJ_TT((TA ^ TB) & TC)>((uint32_t)(tt >> 32), 0xAAAAAAAAu, (uint32_t)(backslash >> 32)) >> 31;
}
MSJ_HD uint64_t unescaped_quotes0(uint64_t quote_chr, uint64_t tt) {
    return lut3<MSJ_TT(TA & ~(TB ^ TC))>(quote_chr, tt, 0xAAAAAAAAAAAAAAAAull);
}

// ---------------------------------------------------------------------------
// UTF-8 (strict, Unicode Table 3-7) on bit-planes.  The reference's checker is
// an empty stub (json_structural_indexer.mojo:16-30); this implements what
// upstream simdjson's checker accepts/rejects.
//
// Two kinds of rules, checked from two sides so that only five planes have to move across
// byte positions (each move is a cross-lane shift of three half-rate operations on the device):
//   * structure -- a byte is a continuation (10xxxxxx) exactly where a lead byte 1, 2 or 3
//     positions back asks for one: the three lead planes are shifted FORWARD onto the bytes they
//     constrain (exp1 / exp2 / exp3);
//   * the four second-byte ranges (E0 A0..BF, ED 80..9F, F0 90..BF, F4 80..8F) -- checked AT THE
//     LEAD byte against the NEXT byte's bits b5 and (b5 | b4), shifted BACKWARD by one (nb5, nb54:
//     two planes instead of the four lead classes moving forward).  The pair (last byte of a
//     block, first byte of the next) is the caller's: `pairmask` clears positions whose next byte
//     it has not supplied, and utf8_boundary_error() checks such a pair from the carry word.
//
// Carry word of the bytes in front of a block (bits of the planes at their top end):
//   bit 0      : byte[-1] is a lead byte (2-, 3- or 4-byte)
//   bits 1..2  : byte[-2], byte[-1] is a 3/4-byte lead        (bit1 = byte[-2])
//   bits 3..5  : byte[-3..-1] is a 4-byte lead                (bit3 = byte[-3])
//   bit 6 / 7 / 8 / 9 : byte[-1] is E0 / ED / F0 / F4
struct Utf8Leads {
    uint64_t lead234, lead34, lead4;  // C0..F7 / E0..F7 / F0..F7
    uint64_t l3, hi2, f;              // 1110xxxx / 11xxxxxx / 1111xxxx (shared with the error terms)
};

MSJ_HD Utf8Leads utf8_leads(const uint64_t p[8]) {
    const uint64_t b3 = p[3], b4 = p[4], b5 = p[5], b6 = p[6], b7 = p[7];
    Utf8Leads u;
    u.hi2 = b7 & b6;                                                       // 11xxxxxx
    u.f = lut3<MSJ_TT(TA & TB & TC)>(u.hi2, b5, b4);                       // 1111xxxx
    u.l3 = lut3<MSJ_TT(TA & TB & ~TC)>(u.hi2, b5, b4);                     // 1110xxxx
    u.lead4 = lut3<MSJ_TT(TA & ~TB)>(u.f, b3, b3);                         // 11110xxx
    u.lead234 = lut3<MSJ_TT(TA & ~(TB & TC))>(u.hi2, u.f, b3);             // C0..F7
    u.lead34 = u.l3 | u.lead4;                                             // E0..F7
    return u;
}

// bits 6..9 of the carry word: the block's last byte is E0 / ED / F0 / F4
MSJ_HD uint32_t utf8_special_top(const uint64_t p[8], const Utf8Leads &u) {
    const uint32_t t = 63;
    const uint32_t b0 = (uint32_t)(p[0] >> t) & 1u, b1 = (uint32_t)(p[1] >> t) & 1u, b2 = (uint32_t)(p[2] >> t) & 1u;
    const uint32_t b3 = (uint32_t)(p[3] >> t) & 1u, l3 = (uint32_t)(u.l3 >> t) & 1u, l4 = (uint32_t)(u.lead4 >> t) & 1u;
    const uint32_t z210 = !(b2 | b1 | b0), n101 = b2 & !b1 & b0, n100 = b2 & !b1 & !b0;
    return ((l3 & !b3 & z210) << 6) | ((l3 & b3 & n101) << 7) | ((l4 & z210) << 8) | ((l4 & n100) << 9);
}

MSJ_HD uint32_t utf8_carry_out(const uint64_t p[8], const Utf8Leads &u) {
    return (uint32_t)(u.lead234 >> 63) | ((uint32_t)(u.lead34 >> 62) << 1) | ((uint32_t)(u.lead4 >> 61) << 3) |
           utf8_special_top(p, u);
}

// The pair (byte[-1], byte[0]) across a block boundary: byte[-1]'s class from the carry word (bits 6..9),
// byte[0]'s bits b5 and b4.  Non-zero = the second byte is outside the range its lead byte allows.
MSJ_HD uint32_t utf8_boundary_error(uint32_t carry_in, uint32_t first_b5, uint32_t first_b4) {
    const uint32_t n5 = first_b5 & 1u, n54 = (first_b5 | first_b4) & 1u;
    return (((carry_in >> 6) & 1u) & (n5 ^ 1u)) |   // E0 80..9F  (overlong)
           (((carry_in >> 7) & 1u) & n5) |          // ED A0..BF  (surrogates)
           (((carry_in >> 8) & 1u) & (n54 ^ 1u)) |  // F0 80..8F  (overlong)
           (((carry_in >> 9) & 1u) & n54);          // F4 90..BF  (> U+10FFFF)
}

// (x << K) with `low` (K bits) shifted in at the bottom: three operations on the halves
// (bit-field extract by the caller, v_lshl_or_b32, v_alignbit_b32) instead of a 64-bit shift.
template <int K>
MSJ_HD uint64_t shl_in(uint64_t x, uint32_t low) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t nh = __builtin_amdgcn_alignbit(hi, lo, 32 - K);
#else
    const uint32_t nh = (hi << K) | (lo >> (32 - K));
#endif
    return u64((lo << K) | low, nh);
}

// What moves across byte positions: the lead planes forward onto the bytes they constrain (with the
// previous bytes' top bits shifted in at the bottom), b5 and b5 | b4 of the NEXT byte backward.
struct Utf8Shifted {
    uint64_t exp1, exp2, exp3;  // a continuation byte is expected here (lead 1 / 2 / 3 bytes back)
    uint64_t nb5, nb54;         // the next byte's b5 / b5 | b4
};

// Error masks of one block.  Planes must already be masked with `valid` (bytes past the end read as
// 0x00 = ASCII, so a sequence truncated at EOF shows up as a missing continuation).  Returns the errors that
// stand as they are; *pair_bad receives the second-byte violations, which the caller keeps only at positions
// whose next byte s.nb5 / s.nb54 really describe (everywhere but a block's last byte without a successor).
MSJ_HD uint64_t utf8_errors_shifted(const uint64_t p[8], const Utf8Leads &u, const Utf8Shifted &s, uint64_t *pair_bad) {
    const uint64_t b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    const uint64_t b4 = p[4], b5 = p[5], b6 = p[6], b7 = p[7];
    // expected ^ continuation (10xxxxxx): a missing or a stray continuation byte
    const uint64_t cont = lut3<MSJ_TT(TA & ~TB)>(b7, b6, b6);
    const uint64_t e12 = s.exp1 | s.exp2;
    uint64_t err = lut3<MSJ_TT((TA | TB) ^ TC)>(e12, s.exp3, cont);
    // F5..FF: 1111xxxx with b3, or with b2 and one of b1, b0
    const uint64_t x567 = lut3<MSJ_TT(TA & (TB | TC))>(b2, b1, b0);
    const uint64_t f5ff = lut3<MSJ_TT(TA & (TB | TC))>(u.f, b3, x567);
    // C0, C1: 1100000x
    const uint64_t z432 = lut3<MSJ_TT(~TA & ~TB & ~TC)>(b4, b3, b2);
    const uint64_t w = lut3<MSJ_TT(TA & ~TB & ~TC)>(z432, b1, b5);
    const uint64_t c0c1 = u.hi2 & w;
    err = lut3<MSJ_TT(TA | TB | TC)>(err, f5ff, c0c1);
    // the second byte of E0 / ED / F0 / F4, judged at the lead byte from the next byte's b5, b5 | b4
    const uint64_t z210 = lut3<MSJ_TT(~TA & ~TB & ~TC)>(b2, b1, b0);       // xxxxx000: E0, F0
    const uint64_t m10 = lut3<MSJ_TT(TA & ~TB)>(b2, b1, b1);               // xxxxx10x: ED (b0 = 1), F4 (b0 = 0)
    const uint64_t t1 = lut3<MSJ_TT(TA & ~TB & ~TC)>(u.l3, b3, s.nb5);     // E0 (given z210) followed by 80..9F
    const uint64_t t2 = lut3<MSJ_TT((TA & ~TB) | TC)>(u.lead4, s.nb54, t1); // ... or F0 followed by 80..8F
    const uint64_t u1 = lut3<MSJ_TT(TA & TB & TC)>(u.l3, b3, s.nb5);       // ED (given m10, b0) followed by A0..BF
    const uint64_t v1 = u.lead4 & s.nb54;                                  // F4 (given m10, ~b0) followed by 90..BF
    const uint64_t sel = lut3<MSJ_TT((TC & TA) | (~TC & TB))>(u1, v1, b0);
    const uint64_t lo_bad = z210 & t2;
    *pair_bad = lut3<MSJ_TT(TA | (TB & TC))>(lo_bad, m10, sel);
    return err;
}

// One block with the bytes around it given explicitly (host harness, CPU tests): carry word of the
// bytes in front, b5 / b4 of the byte behind (0 / 0 at the end of the input: the pair is not judged, a
// lead byte there is a truncated sequence and flagged as such).
MSJ_HD uint64_t utf8_errors(const uint64_t p[8], const Utf8Leads &u, uint32_t carry_in, uint32_t next_b5,
                            uint32_t next_b4, bool have_next) {
    const uint32_t c = carry_in;
    Utf8Shifted s;
    s.exp1 = shl_in<1>(u.lead234, c & 1u);
    s.exp2 = shl_in<2>(u.lead34, (c >> 1) & 3u);
    s.exp3 = shl_in<3>(u.lead4, (c >> 3) & 7u);
    s.nb5 = (p[5] >> 1) | ((uint64_t)(next_b5 & 1u) << 63);
    s.nb54 = ((p[5] | p[4]) >> 1) | ((uint64_t)((next_b5 | next_b4) & 1u) << 63);
    const uint64_t pairmask = have_next ? ~0ull : ~(1ull << 63);
    uint64_t bad2;
    uint64_t err = utf8_errors_shifted(p, u, s, &bad2);
    err |= bad2 & pairmask;
    // the pair across the block's front edge
    err |= utf8_boundary_error(c, (uint32_t)p[5] & 1u, (uint32_t)p[4] & 1u);
    return err;
}

// ---------------------------------------------------------------------------
// One whole block with explicit carries: the per-lane equivalent of
// JsonScanner.next + JsonStructuralIndexer.next for one 64-byte sub-block
// (json_scanner.mojo:64-70, json_string_scanner.mojo:55-69,
//  json_structural_indexer.mojo:129-145).  Used directly by the CPU unit test;
// the kernel inlines the same steps with the cross-lane carry resolution in
// between.
struct BlockCarry {
    uint32_t next_is_escaped, in_string, prev_scalar;
};
struct BlockOut {
    uint64_t structural;  // JsonBlock.structural_start (json_scanner.mojo:24-26)
    uint64_t unescaped;   // ctrl & in_string (json_structural_indexer.mojo:143-145)
};

MSJ_HD BlockOut block_step(const uint32_t x[16], uint64_t valid, BlockCarry &cy) {
    uint64_t p[8];
    bitplanes(x, p);
#pragma unroll
    for (int k = 0; k < 8; k++) p[k] &= valid;
    const Classes c = classify(p, valid);
    // the kernel's two escape paths: carry-in 0 (pieces above), carry-in 1 (escaped_mask)
    uint32_t e_out;
    uint64_t quote;
    if (cy.next_is_escaped == 0) {
        const uint64_t tt = escape_tt0(c.backslash);
        e_out = escape_out0(tt, c.backslash);
        quote = unescaped_quotes0(c.quote_chr, tt);
    } else {
        const uint64_t escaped = escaped_mask(c.backslash, cy.next_is_escaped, &e_out);
        quote = c.quote_chr & ~escaped;
    }
    const uint64_t in_string = prefix_xor(quote) ^ (uint64_t)(-(int64_t)cy.in_string);
    // scalar = ~(op | ws); nonquote_scalar = scalar & ~quote  (json_scanner.mojo:64-70)
    const uint64_t nonquote_scalar = lut3<MSJ_TT(~TA & ~TB & ~TC)>(c.op, c.ws, quote);
    const uint64_t follows = (nonquote_scalar << 1) | cy.prev_scalar;
    // potential_structural_start = op | (scalar & ~follows) = op | (~ws & ~follows)  (:40-49)
    const uint64_t potential = lut3<MSJ_TT(TA | (~TB & ~TC))>(c.op, c.ws, follows);
    BlockOut o;
    o.structural = potential & ~(in_string ^ quote);
    o.unescaped = c.ctrl & in_string;
    cy.next_is_escaped = e_out;
    cy.in_string = (uint32_t)(in_string >> 63);
    cy.prev_scalar = (uint32_t)(nonquote_scalar >> 63);
    return o;
}

}  // namespace msj
