// Test-only harness: compiles mojo_simdjson_amd/csrc/lane_math.h for the host
// (g++) and chains its per-block functions sequentially, so the per-lane bit
// math of the HIP kernel (bit-plane transpose, classification, escape,
// in-string, UTF-8 planes) can be checked against the oracle on a CPU-only box.
// NOT part of the product: the product path is the HIP kernel.
#include <cstring>
#include "../mojo_simdjson_amd/csrc/lane_math.h"
#include "../mojo_simdjson_amd/csrc/token_math.h"

extern "C" {

// returns reference code; same contract as the oracle entry points
int32_t lane_stage1(const uint8_t *buf, uint64_t len, uint32_t *idx, uint64_t cap,
                    uint64_t *n_out, int32_t *utf8_out) {
    if (len + 3 > cap) return 1;
    if (len == 0) return 13;
    msj::BlockCarry cy = {0, 0, 0};
    uint64_t n = 0, bad = 0, u8err = 0;
    uint32_t ucarry = 0;
    for (uint64_t off = 0; off < len; off += 64) {
        uint8_t blk[64];
        std::memset(blk, 0xEE, 64);  // garbage past the end: must be masked by `valid`
        uint64_t nv = len - off < 64 ? len - off : 64;
        std::memcpy(blk, buf + off, nv);
        uint32_t x[16];
        std::memcpy(x, blk, 64);
        uint64_t valid = nv == 64 ? ~0ull : ((1ull << nv) - 1);
        msj::BlockOut o = msj::block_step(x, valid, cy);
        bad |= o.unescaped;
        uint64_t m = o.structural;
        while (m) {
            idx[n++] = (uint32_t)(off + __builtin_ctzll(m));
            m &= m - 1;
        }
        uint64_t p[8];
        msj::bitplanes(x, p);
        for (int k = 0; k < 8; k++) p[k] &= valid;
        msj::Utf8Leads u = msj::utf8_leads(p);
        // the byte behind the block (the kernel's lanes see it through a backward cross-lane shift)
        const bool have_next = off + 64 < len;
        const uint32_t nb = have_next ? buf[off + 64] : 0u;
        u8err |= msj::utf8_errors(p, u, ucarry, nb >> 5, nb >> 4, have_next);
        ucarry = msj::utf8_carry_out(p, u);
    }
    if (len % 64 == 0 && (ucarry & 0x3F)) u8err |= 1;  // sequence truncated exactly at EOF
    *utf8_out = u8err ? 11 : 0;
    if (cy.in_string) return 15;
    if (bad) return 14;
    idx[n] = (uint32_t)len; idx[n + 1] = (uint32_t)len; idx[n + 2] = 0;
    *n_out = n;
    if (n == 0) return 13;
    return 0;
}

void lane_bitplanes(const uint8_t *blk64, uint64_t *planes8) {
    uint32_t x[16];
    std::memcpy(x, blk64, 64);
    msj::bitplanes(x, planes8);
}

// digit, sow, backslash, blank masks of one 64-byte block (the span kernel's classes)
void lane_span_classes(const uint8_t *blk64, uint64_t *out4) {
    uint32_t x[16];
    std::memcpy(x, blk64, 64);
    uint64_t p[8];
    msj::bitplanes(x, p);
    msj::SpanClasses c = msj::span_classes(p);
    out4[0] = c.digit; out4[1] = c.sow; out4[2] = c.backslash; out4[3] = c.blank;
}

// quote, dote masks of one 64-byte block (the tile kernel's extra classes)
void lane_tile_classes(const uint8_t *blk64, uint64_t *out2) {
    uint32_t x[16];
    std::memcpy(x, blk64, 64);
    uint64_t p[8];
    msj::bitplanes(x, p);
    msj::TileClasses c = msj::tile_classes(p);
    out2[0] = c.quote; out2[1] = c.dote;
}

uint32_t lane_top_run(uint64_t m) { return msj::top_run(m); }
uint64_t lane_prefix_xor(uint64_t m) { return msj::prefix_xor(m); }
}
