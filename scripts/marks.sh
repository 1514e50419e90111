#!/bin/bash
# Static per-section instruction counts of the hot loop (-DMSJ_MARKS analysis build, CPU only).
cd "$(dirname "$0")/../mojo_simdjson_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMSJ_MARKS -S --cuda-device-only stage1_kernel.hip -o /tmp/marks.s 2>/dev/null
python3 - <<'PY'
import re,collections
FULL = {"v_and_b32","v_or_b32","v_xor_b32","v_not_b32","v_add_u32","v_sub_u32","v_subrev_u32","v_mov_b32",
        "v_lshrrev_b32","v_bitop3_b32","v_add_co_u32","v_addc_co_u32","v_sub_co_u32","v_subb_co_u32"}
def cost(m):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", m)
    if base in FULL or base.startswith("v_cmp"): return 1
    if base.endswith("_b64") or base.endswith("_u64"): return 4
    return 2
lines=open('/tmp/marks.s').read().split('\n')
cur=None; sec=[]
for i,l in enumerate(lines):
    m=re.search(r"; MSJ_MARK (\w+)",l)
    if m:
        cur=[m.group(1),i,0,0,0,0,0,collections.Counter()]; sec.append(cur); continue
    if cur is None: continue
    m=re.match(r"\s+(v_[a-z0-9_]+)",l)
    if m: cur[2]+=1; cur[3]+=cost(m.group(1)); cur[7][m.group(1)]+=1
    elif re.match(r"\s+s_(cbranch|branch)",l): cur[6]+=1; cur[4]+=1
    elif re.match(r"\s+s_",l): cur[4]+=1
    elif re.match(r"\s+(ds_|global_|buffer_|flat_)",l): cur[5]+=1
tot=[0,0,0]
for s in sec[:40]:
    print(f"mark {s[0]:>3s} line {s[1]:6d}: valu {s[2]:4d} units {s[3]:4d} salu {s[4]:4d} (br {s[6]:3d}) mem {s[5]:3d}  {s[7].most_common(5)}")
PY
