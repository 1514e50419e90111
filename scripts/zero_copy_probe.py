#!/usr/bin/env python3
"""Probe: the stage-1 kernel reading its input from / writing its indices to pinned HOST memory over PCIe
(no DMA copies at all), against device-resident buffers.  256 MiB minified."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import synth  # noqa: E402
from mojo_simdjson_amd.device import Stage1Device  # noqa: E402

dev = Stage1Device(0)
torch.cuda.set_device(0)
u = synth.workload("minified", 64 << 20)
h_buf = torch.from_numpy(u).repeat(4).pin_memory()
nbytes = h_buf.numel()
d_buf = h_buf.to(dev.device)
cap = int(nbytes * 0.3)
d_idx = torch.empty(cap, dtype=torch.int32, device=dev.device)
h_idx = torch.empty(cap, dtype=torch.int32).pin_memory()
d_carry = dev.new_carry()


def run(name, buf, idx, reps=5):
    dev.index(buf, idx, d_carry)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.index(buf, idx, d_carry)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n = int(dev.fetch(d_carry).count)
    print(f"{name:34s}: {dt * 1e3:7.3f} ms, {nbytes / dt / 1e9:7.2f} GB/s of JSON, n = {n}", flush=True)


run("input device, indices device", d_buf, d_idx, 20)
run("input HOST (pinned), indices device", h_buf, d_idx)
run("input device, indices HOST (pinned)", d_buf, h_idx)
run("input HOST, indices HOST", h_buf, h_idx)
ref = d_idx.cpu()
dev.index(h_buf, h_idx, d_carry)
torch.cuda.synchronize()
n = int(dev.fetch(d_carry).count)
print("host-side index array equals the device one:", bool(torch.equal(ref[:n + 3], h_idx[:n + 3])))
dev.close()
