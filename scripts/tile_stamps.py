#!/usr/bin/env python3
"""Where a workgroup of token_tiles spends its time: real-time stamps (100 MHz) per wave from the diagnostic build
(make -C mojo_simdjson_amd/csrc tile_stamps), one msj_stage2_prep_device call on a 1 GiB workload.
    python3 scripts/tile_stamps.py [workload]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mojo_simdjson_amd import _lib, synth  # noqa: E402
_lib.LIB_PATH = os.path.join(ROOT, "scripts", "libmsj_stage1_tile_stamps.so")
from mojo_simdjson_amd.device import Stage1Device, _ptr  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "minified"
dev = Stage1Device(0)
u = synth.workload(w, 64 << 20)
d_buf = torch.from_numpy(u).to(dev.device).repeat((1 << 30) // u.size)
nbytes = d_buf.numel()
d_idx = torch.empty(int(nbytes * 0.3), dtype=torch.int32, device=dev.device)
d_carry = dev.new_carry()
dev.index(d_buf, d_idx, d_carry)
n = int(dev.fetch(d_carry).count)
dv = dev.device
outs = [torch.empty(n, dtype=torch.uint8, device=dv), torch.empty(n, dtype=torch.int32, device=dv), torch.empty(n, dtype=torch.int32, device=dv),
        torch.empty(n, dtype=torch.uint8, device=dv)]
d_res = torch.zeros(24, dtype=torch.uint8, device=dv)
GB = int(dev.lib.msj_debug_tile_group(0))
WAVES = GB // 4096 + 1
groups = (nbytes + GB - 1) // GB
d_st = torch.zeros(groups * WAVES * 8, dtype=torch.int64, device=dv)
def call():
    rc = dev.lib.msj_stage2_prep_device(dev.ctx, _ptr(d_buf), nbytes, _ptr(d_idx), n, _ptr(outs[0]), _ptr(outs[1]), None, _ptr(outs[2]),
                                        _ptr(outs[3]), _ptr(d_res), dev._stream())
    assert rc == 0
for _ in range(20):
    call()
dev.lib.msj_debug_set_tile_stamps.argtypes = [ctypes.c_void_p]
assert dev.lib.msj_debug_set_tile_stamps(ctypes.c_void_p(d_st.data_ptr())) == 0
call()
torch.cuda.synchronize()
st = d_st.cpu().numpy().reshape(groups, WAVES, 8).astype(np.int64)
ok = st[:, :, 0] != 0
t0 = st[:, :, 0][ok].min()
end = st[:, :, 5].max()
print(f"{w}: kernel span {(end - t0) / 100:.1f} us, {groups} workgroups")
names = ["entry -> table known, loads issued", "loads issued -> bytes staged", "staged -> classified", "classified -> barrier passed", "chunk loop"]
for k in range(5):
    d = (st[:, :, k + 1] - st[:, :, k])[ok] / 100.0
    print(f"  {names[k]:38s} median {np.median(d):6.2f} us   mean {d.mean():6.2f}   p90 {np.percentile(d, 90):6.2f}")
life = (st[:, :, 5].max(axis=1) - st[:, :, 0].min(axis=1)) / 100.0
print(f"  workgroup lifetime (first entry -> last exit): median {np.median(life):.2f} us, mean {life.mean():.2f}; "
      f"resident workgroups on average {life.sum() / ((end - t0) / 100.0):.0f} (slots: see the first entry of the last line)")
wl = ((st[:, :, 5] - st[:, :, 0])[ok] / 100.0)
print(f"  wave lifetime mean {wl.mean():.2f} us; resident waves on average {wl.sum() / ((end - t0) / 100.0):.0f}")
# gap between a workgroup's exit and the next entry: start times sorted, how many start within the kernel's first microseconds
starts = np.sort(st[:, :, 0].min(axis=1) - t0) / 100.0
print("  workgroups started after 1 / 2 / 5 / 10 / 20 us:", [int((starts <= x).sum()) for x in (1, 2, 5, 10, 20)])
dev.close()
