#!/usr/bin/env python3
"""Static instruction counts per section of token_tiles<true, true>'s chunk loop: compiles a copy of tokens_kernel.hip
with `; MARK` comments at the section borders and counts the instructions between them in the ISA.
    python3 scripts/span_marks.py   (needs hipcc; no GPU)"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "mojo_simdjson_amd/csrc/tokens_kernel.hip")).read()
i = src.index("void token_tiles(")
t = src[i:]
def mark(after, name):
    global t
    j = t.index(after)
    t = t[:j] + 'asm volatile("; MARK %s");\n' % name + t[j:]
mark("        // the next chunk's indices: requested now", "LOOPTOP")
mark("        uint32_t e0 = 0, f0 = 0, c0 = 0, e1 = 0, f1 = 0, c1 = 0;\n        if (kSpans && staged && allhere) {", "PRE")
mark("            const uint32_t rs0 = i0 - base, rs1 = i1 - base, rn1 = i2 - base;", "HOT")
mark("            if (__ballot((again | (s0 & s1)) != 0u) != 0ull) {", "RARE")
mark("            e0 = s0 ? e : 0u, f0 = s0 ? f : 0u;\n            if (!(s0 & s1))", "HOTEND")
mark("        } else if (!staged) {", "GENERIC")
mark("        if (allhere && wide) {", "STORES")
mark("        if (kFused) {\n            // the chunk.s bracket counts".replace(".", chr(39), 1), "AGG")
mark("        i0 = n_i0, i1 = n_i1, nxt = n_nxt;", "LOOPEND")
src = (src[:i] + t).replace('#include "../../include/msj_stage1.h"', '#include "%s/include/msj_stage1.h"' % ROOT)
d = tempfile.mkdtemp()
for h in ("lane_math.h", "stage1_kernel.h", "token_math.h"):
    open(os.path.join(d, h), "w").write(open(os.path.join(ROOT, "mojo_simdjson_amd/csrc", h)).read())
open(os.path.join(d, "tokens_kernel.hip"), "w").write(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps=obj", "-c",
                       "tokens_kernel.hip", "-o", "tok.o"], cwd=d, stderr=subprocess.DEVNULL)
lines = open(os.path.join(d, "tokens_kernel-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
a = next(k for k, l in enumerate(lines) if l.startswith("_ZN10msj_tokens11token_tilesILb1ELb1EEE"))
b = next(k for k in range(a, len(lines)) if "Lfunc_end" in lines[k])
lines = lines[a:b]
open(os.path.join(d, "t11.s"), "w").write("\n".join(lines))
def count(a, b):
    c = dict(valu=0, salu=0, lds=0, vmem=0, branches=0)
    for x in lines[a:b]:
        x = x.strip()
        if x.startswith("v_"): c["valu"] += 1
        elif x.startswith("s_cbranch") or x.startswith("s_branch"): c["branches"] += 1; c["salu"] += 1
        elif x.startswith("s_"): c["salu"] += 1
        elif x.startswith("ds_"): c["lds"] += 1
        elif x.startswith(("global_", "flat_", "buffer_")): c["vmem"] += 1
    return c
marks = [(k, l.split("MARK")[1].strip()) for k, l in enumerate(lines) if "MARK" in l]
for (a, na), (b, nb) in zip(marks, marks[1:]):
    print(f"{a:5d} {na:8s} -> {nb:8s} {count(a, b)}")
print("behind the last mark", count(marks[-1][0], len(lines)))
print("ISA:", os.path.join(d, "t11.s"))
