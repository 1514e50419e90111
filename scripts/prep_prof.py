#!/usr/bin/env python3
"""Sustained rate of msj_stage2_prep_device (rows f1 + f2 + f4) on one 1 GiB workload, outputs allocated once and
nothing waited for between calls.  Under `rocprofv3 --kernel-trace --stats -- python3 scripts/prep_prof.py minified`
the kernel stats give the split between its kernels.
    python3 scripts/prep_prof.py [workload] [--match] [--iters N] [--warm N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import synth  # noqa: E402
from mojo_simdjson_amd.device import Stage1Device, _ptr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="minified")
ap.add_argument("--match", action="store_true")
ap.add_argument("--pairs", action="store_true", help="the partners as a compact list (msj_stage2_prep_pairs_device)")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--warm", type=int, default=200)
ap.add_argument("--mib", type=int, default=1024)
ap.add_argument("--mode", type=int, default=0, help="0 = by density (the product), 1 = kernel organised by tokens, 2 = by tiles")
ap.add_argument("--lib", default=None, help="A/B: load this build of libmsj_stage1.so")
a = ap.parse_args()
if a.lib:
    from mojo_simdjson_amd import _lib

    _lib.LIB_PATH = os.path.abspath(a.lib)

dev = Stage1Device(0)
dev.lib.msj_debug_set_span_mode(dev.ctx, a.mode)
torch.cuda.set_device(0)
u = synth.workload(a.workload, 64 << 20)
d_buf = torch.from_numpy(u).to(dev.device).repeat((a.mib << 20) // u.size)
nbytes = d_buf.numel()
d_idx = torch.empty(int(nbytes * 0.3), dtype=torch.int32, device=dev.device)
d_carry = dev.new_carry()
dev.index(d_buf, d_idx, d_carry)
n = int(dev.fetch(d_carry).count)
dv = dev.device
d_type = torch.empty(n, dtype=torch.uint8, device=dv)
d_depth = torch.empty(n, dtype=torch.int32, device=dv)
d_match = torch.empty(n, dtype=torch.int32, device=dv) if a.match else None
d_pairs = torch.empty((n, 2), dtype=torch.int32, device=dv) if a.pairs else None
d_end = torch.empty(n, dtype=torch.int32, device=dv)
d_flags = torch.empty(n, dtype=torch.uint8, device=dv)
d_res = torch.zeros(24, dtype=torch.uint8, device=dv)


def call():
    if a.pairs:
        rc = dev.lib.msj_stage2_prep_pairs_device(dev.ctx, _ptr(d_buf), nbytes, _ptr(d_idx), n, _ptr(d_type), _ptr(d_depth), _ptr(d_pairs),
                                                  _ptr(d_end), _ptr(d_flags), _ptr(d_res), None, dev._stream())
        assert rc == 0, rc
        return
    rc = dev.lib.msj_stage2_prep_device(dev.ctx, _ptr(d_buf), nbytes, _ptr(d_idx), n, _ptr(d_type), _ptr(d_depth),
                                        _ptr(d_match) if a.match else None, _ptr(d_end), _ptr(d_flags), _ptr(d_res), dev._stream())
    assert rc == 0, rc


for _ in range(a.warm):
    call()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(a.iters):
    call()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print(f"{a.workload}{' +match' if a.match else ''}{' +pairs' if a.pairs else ''} mode {a.mode}: {n} structurals, {nbytes} bytes: {ms:.4f} ms per call "
      f"({ms * (1 << 30) / nbytes:.4f} ms per GiB, {n / ms / 1e6:.1f} G structurals/s), {a.iters} calls after {a.warm} warm-up")
dev.close()
