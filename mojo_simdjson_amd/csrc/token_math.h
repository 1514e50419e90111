// token_math.h -- per-lane mask arithmetic of the token kernels (tokens_kernel.hip) that the stage-1 kernel does not
// share: host + device like lane_math.h, unit-tested on the CPU (tests/test_lane_math.py).  Kept apart from
// lane_math.h because that file is part of the stage-1 kernel's source hash (msj_version(), profiles/traffic.json).
#pragma once
#include "lane_math.h"

namespace msj {

// Two classes beside span_classes() for the kernel organised by tiles (token_tiles), which resolves escapes per block
// the way stage 1 does and answers "is the byte behind the digits one of . e E" from a bitmap instead of a byte read:
//   quote      22 (raw: before escape resolution)
//   dote       2E 45 65                      (what makes parse_number take the float branch, number_parsing.mojo:50-53)
struct TileClasses {
    uint64_t quote, dote;
};
MSJ_HD TileClasses tile_classes(const uint64_t p[8]) {
    const uint64_t b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3];
    const uint64_t b4 = p[4], b5 = p[5], b6 = p[6], b7 = p[7];
    TileClasses c;
    const uint64_t lo3 = lut3<MSJ_TT(~TA & ~TB)>(b4, b3, b3);                    // b4 = b3 = 0
    {   // 22 = 0010 0010
        const uint64_t h2 = lut3<MSJ_TT(~TA & ~TB & TC)>(b7, b6, b5);
        const uint64_t g010 = lut3<MSJ_TT(~TA & TB & ~TC)>(b2, b1, b0);
        c.quote = lut3<MSJ_TT(TA & TB & TC)>(h2, lo3, g010);
    }
    {   // 2E = 0010 1110;  45 / 65 = 01x0 0101
        const uint64_t h2 = lut3<MSJ_TT(~TA & ~TB & TC)>(b7, b6, b5);
        const uint64_t g110 = lut3<MSJ_TT(TA & TB & ~TC)>(b2, b1, b0);
        const uint64_t dot = lut3<MSJ_TT(TA & TB & TC)>(h2, g110, lut3<MSJ_TT(~TA & TB)>(b4, b3, b3));
        const uint64_t g101 = lut3<MSJ_TT(TA & ~TB & TC)>(b2, b1, b0);
        const uint64_t ee = lut3<MSJ_TT(TA & TB & TC)>(lut3<MSJ_TT(~TA & TB)>(b7, b6, b6), lo3, g101);
        c.dote = dot | ee;
    }
    return c;
}

}  // namespace msj
