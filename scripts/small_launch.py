#!/usr/bin/env python3
"""Short launches: ms per GiB of a 1/16 .. 1/2 GiB minified input, (a) the same buffers launch
after launch -- what scripts/size_sweep.sh and bench.py --gib-per-gpu do: a 0.25 GiB input and its indices fit the 256 MiB
Infinity Cache, so a replay partly measures the cache -- and (b) cycling through distinct inputs and index buffers of >= 2 GiB
in total, which is what a caller with a stream of documents sees.      scripts/small_launch.py [lib.so ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mojo_simdjson_amd import _lib, synth  # noqa: E402


def run(lib):
    if lib:
        _lib.LIB_PATH = os.path.abspath(lib)
    from mojo_simdjson_amd.device import Stage1Device
    dev = Stage1Device(0)
    torch.cuda.set_device(0)
    unit = torch.from_numpy(synth.workload("minified", 64 << 20)).to(dev.device)
    print(f"library {lib or 'mojo_simdjson_amd/libmsj_stage1.so'}  ({dev.version() if hasattr(dev, 'version') else ''})")
    for gib in (0.0625, 0.125, 0.25, 0.5):
        n = int(gib * (1 << 30)) // unit.numel() * unit.numel() or unit.numel()
        copies = max(2, int((2 << 30) / (1.8 * n)))
        bufs = [unit.repeat(n // unit.numel()).clone() for _ in range(copies)]
        idxs = [torch.empty(int(0.25 * n) + 16, dtype=torch.int32, device=dev.device) for _ in range(copies)]
        res = dev.new_carry()
        out = []
        for cyc in (1, copies):
            for i in range(600):  # settle
                dev.index(bufs[i % cyc], idxs[i % cyc], res)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(600):
                dev.index(bufs[i % cyc], idxs[i % cyc], res)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 600
            r = dev.fetch(res)
            assert r.internal_error == 0 and int(r.count) > 0
            alg = n + 4 * int(r.count)
            out.append(f"{ms:.4f} ms  {ms / (n / 2**30):.4f} ms/GiB  frac {alg / ms / 1e6 / 8000:.3f}")
        print(f"  {n / 2**30:.4f} GiB  same buffers: {out[0]}   |  {copies} buffer pairs in turn: {out[1]}", flush=True)
        del bufs, idxs
    dev.close()


if __name__ == "__main__":
    for lib in (sys.argv[1:] or [None]):
        run(lib)
