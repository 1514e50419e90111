#!/usr/bin/env python3
"""stage 1 that writes the type bytes itself (msj_stage1_types_device) + the depth pass on them
(msj_depth_from_types_device) against stage 1 + msj_tokens_device / msj_stage2_prep_device, 1 GiB, one box, alternating,
settled clocks.  The types pair gives row f1 only (type, depth; no string / number spans): compared with both.
    python3 scripts/fused_types.py [workload]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import synth  # noqa: E402
from mojo_simdjson_amd.device import Stage1Device, _ptr  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "minified"
dev = Stage1Device(0)
dv = dev.device
u = synth.workload(wl, 64 << 20)
d_buf = torch.from_numpy(u).to(dv).repeat((1 << 30) // u.size)
nbytes = d_buf.numel()
d_idx = torch.empty(int(nbytes * 0.3), dtype=torch.int32, device=dv)
d_res = dev.new_carry()
dev.index(d_buf, d_idx, d_res)
n = int(dev.fetch(d_res).count)
d_types = torch.empty(d_idx.numel(), dtype=torch.uint8, device=dv)
d_type2 = torch.empty(n, dtype=torch.uint8, device=dv)
d_depth = torch.empty(n, dtype=torch.int32, device=dv)
d_depth2 = torch.empty(n, dtype=torch.int32, device=dv)
d_end = torch.empty(n, dtype=torch.int32, device=dv)
d_flags = torch.empty(n, dtype=torch.uint8, device=dv)
d_tr = torch.zeros(24, dtype=torch.uint8, device=dv)
d_tr2 = torch.zeros(24, dtype=torch.uint8, device=dv)


def stage1():
    dev.index(d_buf, d_idx, d_res)


def stage1_types():
    dev.index_types(d_buf, d_idx, d_types, d_res)


def depth_from_types():
    dev.depth_from_types(d_types, n, d_depth=d_depth, d_result=d_tr)


def prep():
    rc = dev.lib.msj_stage2_prep_device(dev.ctx, _ptr(d_buf), nbytes, _ptr(d_idx), n, _ptr(d_type2), _ptr(d_depth2), None, _ptr(d_end),
                                        _ptr(d_flags), _ptr(d_tr2), dev._stream())
    assert rc == 0


def tokens():
    rc = dev.lib.msj_tokens_device(dev.ctx, _ptr(d_buf), nbytes, _ptr(d_idx), n, _ptr(d_type2), _ptr(d_depth2), None, _ptr(d_tr2), dev._stream())
    assert rc == 0


def timed(fns, iters=150, warm=100):
    for _ in range(warm):
        for f in fns:
            f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# correctness first: the type bytes and the depths equal what the separate calls give
stage1_types()
depth_from_types()
tokens()
torch.cuda.synchronize()
assert int(dev.fetch(d_res).count) == n
assert torch.equal(d_types[:n], d_type2[:n]), "type bytes differ"
assert torch.equal(d_depth[:n], d_depth2[:n]), "depths differ"
print(f"{wl}: {n} structurals in {nbytes} bytes; types and depths of the fused pair equal msj_tokens_device's")
scale = (1 << 30) / nbytes
for rnd in range(2):
    rows = [("stage 1", [stage1]), ("stage 1 + types", [stage1_types]), ("depth from types", [depth_from_types]),
            ("stage 1 + types, depth from types", [stage1_types, depth_from_types]),
            ("msj_tokens_device (type + depth)", [tokens]), ("stage 1, msj_tokens_device", [stage1, tokens]),
            ("msj_stage2_prep_device (type + depth + spans)", [prep]), ("stage 1, msj_stage2_prep_device", [stage1, prep])]
    for name, fns in rows:
        t_end = time.perf_counter() + 0.2
        while time.perf_counter() < t_end:
            for f in fns:
                f()
            torch.cuda.synchronize()
        print(f"  round {rnd}: {name:48s} {timed(fns) * scale:.4f} ms per GiB", flush=True)
dev.close()
