#!/usr/bin/env python3
"""Rate of the token pre-pass (row f1) on the 1 GiB workloads: type byte + nesting depth per structural.
    python3 scripts/tokens_rate.py [span mode]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import synth  # noqa: E402
from mojo_simdjson_amd.device import Stage1Device  # noqa: E402

dev = Stage1Device(0)
if len(sys.argv) > 1:  # 1 = the kernel organised by tokens, 2 = by tiles; default: the product's choice by density
    dev.lib.msj_debug_set_span_mode(dev.ctx, int(sys.argv[1]))
    print(f"span mode {sys.argv[1]}")
torch.cuda.set_device(0)
for name in ("minified", "utf8", "pretty4"):
    u = synth.workload(name, 64 << 20)
    d_buf = torch.from_numpy(u).to(dev.device).repeat((1 << 30) // u.size)
    nbytes = d_buf.numel()
    d_idx = torch.empty(int(nbytes * 0.3), dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n = int(dev.fetch(d_res).count)
    d_type = torch.empty(n, dtype=torch.uint8, device=dev.device)
    d_depth = torch.empty(n, dtype=torch.int32, device=dev.device)
    d_match = torch.empty(n, dtype=torch.int32, device=dev.device)
    for _ in range(3):
        dev.tokens(d_buf, nbytes, d_idx, n, d_type, d_depth)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        t, d, res = dev.tokens(d_buf, nbytes, d_idx, n, d_type, d_depth)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    alg = nbytes + 10 * n  # the buffer + 4 B index in, 1 B type out; then 1 B type in, 4 B depth out
    print(f"{name:9s}: {n} structurals, {ms:.3f} ms, {n / ms / 1e6:.1f} G structurals/s, "
          f"{alg / ms / 1e6:.0f} GB/s of buffer + 10 B/structural ({alg / ms / 1e6 / 8000:.3f} of 8 TB/s), "
          f"max depth {res.max_depth}, final {res.final_depth}; as input rate {nbytes / ms / 1e6:.0f} GB/s of JSON")
    dev.tokens(d_buf, nbytes, d_idx, n, d_type, d_depth, d_match)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        dev.tokens(d_buf, nbytes, d_idx, n, d_type, d_depth, d_match)
    e1.record()
    torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / 10
    print(f"           with bracket matching: {ms2:.3f} ms ({n / ms2 / 1e6:.1f} G structurals/s)")
    dev.token_spans(d_buf, nbytes, d_idx, n)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        dev.token_spans(d_buf, nbytes, d_idx, n)
    e1.record()
    torch.cuda.synchronize()
    ms3 = e0.elapsed_time(e1) / 10
    print(f"           token spans (strings, numbers): {ms3:.3f} ms ({n / ms3 / 1e6:.1f} G structurals/s, {nbytes / ms3 / 1e6:.0f} GB/s of JSON)")
    del d_type, d_depth, d_match
    dev.stage2_prep(d_buf, nbytes, d_idx, n)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        dev.stage2_prep(d_buf, nbytes, d_idx, n)
    e1.record()
    torch.cuda.synchronize()
    ms4 = e0.elapsed_time(e1) / 10
    print(f"           type + depth + spans in one go (msj_stage2_prep_device): {ms4:.3f} ms ({n / ms4 / 1e6:.1f} G structurals/s, "
          f"{nbytes / ms4 / 1e6:.0f} GB/s of JSON) vs {ms + ms3:.3f} ms for the two calls")
dev.close()
