#!/usr/bin/env python3
"""Where a workgroup of match_compact (the pairs form's in-block pairing on the compact bracket list) spends a block:
real-time stamps (100 MHz) per wave and block from the diagnostic build (make -C mojo_simdjson_amd/csrc tile_stamps), one
msj_stage2_prep_pairs_device call on a 1 GiB workload.
    python3 scripts/compact_stamps.py [workload]"""
import ctypes
import os
import sys

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
d_type = torch.empty(n, dtype=torch.uint8, device=dv)
d_depth = torch.empty(n, dtype=torch.int32, device=dv)
d_pairs = torch.empty((n, 2), dtype=torch.int32, device=dv)
d_end = torch.empty(n, dtype=torch.int32, device=dv)
d_flags = torch.empty(n, dtype=torch.uint8, device=dv)
d_res = torch.zeros(24, dtype=torch.uint8, device=dv)
GB = int(dev.lib.msj_debug_tile_group(0))
slots = ((nbytes + GB - 1) // GB) * (GB // 4096 + 1)  # token_tiles stamps the same buffer first: room for both
d_st = torch.zeros(slots * 8, dtype=torch.int64, device=dv)


def call():
    rc = dev.lib.msj_stage2_prep_pairs_device(dev.ctx, _ptr(d_buf), nbytes, _ptr(d_idx), n, _ptr(d_type), _ptr(d_depth), _ptr(d_pairs),
                                              _ptr(d_end), _ptr(d_flags), _ptr(d_res), None, dev._stream())
    assert rc == 0


for _ in range(20):
    call()
dev.lib.msj_debug_set_tile_stamps.argtypes = [ctypes.c_void_p]
assert dev.lib.msj_debug_set_tile_stamps(ctypes.c_void_p(d_st.data_ptr())) == 0
call()
torch.cuda.synchronize()
res = _lib.MsjTokensResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
nbrk = 2 * res.reserved - res.final_depth
ncb = (nbrk + 2047) // 2048
st = d_st.cpu().numpy()[: ncb * 4 * 8].reshape(ncb, 4, 8).astype(np.int64)
full = st[: ncb - 1]  # (the last block is partial)
t0, end = full[:, :, 0].min(), full[:, :, 7].max()
print(f"{w}: {nbrk} brackets, {ncb} blocks of 2 048; match_compact spans {(end - t0) / 100:.1f} us")
names = ["loop top -> own loads arrived, bitmaps zeroed", "wait at barrier 1", "(a) opening brackets' bits (LDS atomics) + barrier 2",
         "(b) summary words (ballots) + barrier 3", "(c) closing brackets look up, records out + barrier 4",
         "(d) survivors: returning atomic, lists", "(e) tree levels + last barrier"]
for k in range(7):
    d = (full[:, :, k + 1] - full[:, :, k]) / 100.0
    print(f"  {names[k]:56s} median {np.median(d):6.2f} us   mean {d.mean():6.2f}   p90 {np.percentile(d, 90):6.2f}")
life = (full[:, :, 7].max(axis=1) - full[:, :, 0].min(axis=1)) / 100.0
print(f"  a block in its workgroup: median {np.median(life):.2f} us, mean {life.mean():.2f}; blocks in flight on average "
      f"{life.sum() / ((end - t0) / 100.0):.0f}")
dev.close()
