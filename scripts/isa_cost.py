#!/usr/bin/env python3
"""Weighted VALU issue cost of a gfx950 .s file (or of the part between two markers),
using the per-instruction rates measured by scripts/ubench/valu_ops.hip:
full rate = 1 unit (~2 cycles at 4 waves/SIMD), half rate = 2 units."""
import re, sys, collections
FULL = {"v_and_b32","v_or_b32","v_xor_b32","v_not_b32","v_add_u32","v_sub_u32","v_subrev_u32","v_mov_b32",
        "v_lshrrev_b32","v_bitop3_b32","v_cmp","v_cmpx","v_add_co_u32","v_addc_co_u32","v_sub_co_u32","v_subb_co_u32"}
def cost(m):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    if base in FULL or base.startswith("v_cmp"): return 1
    if base.endswith("_b64") or base.endswith("_u64"): return 4
    return 2
path = sys.argv[1]
tot = collections.Counter(); units = collections.Counter()
for line in open(path):
    m = re.match(r"\s+(v_[a-z0-9_]+)", line)
    if m:
        tot[m.group(1)] += 1; units[m.group(1)] += cost(m.group(1))
print("VALU instructions:", sum(tot.values()), " cost units:", sum(units.values()))
for k, v in units.most_common(18):
    print(f"  {k:28s} n={tot[k]:5d} units={v:5d}")
