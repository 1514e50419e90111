#!/usr/bin/env python3
"""A/B of builds of libmsj_stage1.so on the density extremes and the BASELINE workloads in ONE process per build,
alternating builds, settled clocks: python scripts/ab_density.py "<cases>" a.so b.so ...   (cases: names of
tests/density_sweep.py's CASES, comma separated; default: the d ~ 0 rows and the three BASELINE workloads).
Each (build, case): 0.3 s of untimed launches, then 200 timed ones (HIP events); two alternating rounds."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import os, sys, time, ctypes
import numpy as np, torch
sys.path.insert(0, %(root)r)
from mojo_simdjson_amd import _lib, synth
_lib.LIB_PATH = %(lib)r
from mojo_simdjson_amd.device import Stage1Device
UNIT = 64 << 20
CASES = {"blanks": lambda: synth.extreme(UNIT - 52, 3), "one string": lambda: synth.extreme(UNIT - 52, 2),
         "pretty8": lambda: synth.workload("pretty8", UNIT), "pretty4": lambda: synth.workload("pretty4", UNIT),
         "utf8": lambda: synth.workload("utf8", UNIT), "minified": lambda: synth.workload("minified", UNIT),
         "d0.5": lambda: synth.extreme(UNIT - 52, 4), "d1.0": lambda: synth.extreme(UNIT - 52, 0)}
dev = Stage1Device(0)
for name in %(cases)r:
    u = CASES[name]()
    d_buf = torch.from_numpy(u).to(dev.device).repeat((1 << 30) // u.size)
    n = d_buf.numel()
    d_idx = torch.empty(n + 16 if name.startswith("d1") else n // 2 + 1024, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(25):
            dev.index(d_buf, d_idx, d_res)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        dev.index(d_buf, d_idx, d_res)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    res = dev.fetch(d_res)
    alg = n + 4 * int(res.count)
    print(f"%(tag)s {name:12s} {ms:.4f} ms  ingest {n / ms / 1e6:7.1f} GB/s  (N+4S)/t {alg / ms / 1e6:7.1f} GB/s = {alg / ms / 1e6 / 8000:.4f} of 8 TB/s  count {int(res.count)} code {int(res.code)}", flush=True)
    del d_buf, d_idx
dev.close()
'''


def main():
    cases = [c.strip() for c in (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else "blanks,one string,pretty4,utf8,minified").split(",")]
    libs = sys.argv[2:] or [os.path.join(ROOT, "mojo_simdjson_amd", "libmsj_stage1.so")]
    for rnd in range(2):
        for lib in libs:
            code = CHILD % {"root": ROOT, "lib": os.path.abspath(lib), "cases": cases, "tag": os.path.basename(lib)}
            subprocess.run([sys.executable, "-c", code], check=False)


if __name__ == "__main__":
    main()
