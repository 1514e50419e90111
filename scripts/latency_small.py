#!/usr/bin/env python3
"""Latency of the host-pointer entry point msj_stage1 (H2D + kernel + D2H of n + 3 indices + sync) against the
input size, called through ctypes directly (no Python facade): where the GPU path crosses a CPU stage 1."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import _lib, synth  # noqa: E402

lib = _lib.load()
n_out = ctypes.c_uint64(0)
verdict = ctypes.c_int32(0)
print("bytes        us/call    GB/s   (reference port on the host: 0.37 GB/s = 2.7 us per KB)")
for size in (64, 1 << 10, 4 << 10, 16 << 10, 64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20):
    data = np.ascontiguousarray(synth.unit(size)) if size >= 4096 else np.frombuffer((b'{"a":[1,2,3],"b":"xyz"},' * 64)[:size - 1] + b"]", dtype=np.uint8).copy()
    size = int(data.size)
    idx = np.zeros(size + 3, dtype=np.uint32)
    args = (data.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(size), idx.ctypes.data_as(ctypes.c_void_p),
            ctypes.c_uint64(idx.size), ctypes.byref(n_out), ctypes.byref(verdict), ctypes.c_uint32(0))
    for _ in range(5):
        rc = lib.msj_stage1(*args)
    reps = 200 if size <= (1 << 20) else 30
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = lib.msj_stage1(*args)
    dt = (time.perf_counter() - t0) / reps
    print(f"{size:9d} {dt * 1e6:10.1f} {size / dt / 1e9:7.2f}   rc {rc}, {n_out.value} structurals")
