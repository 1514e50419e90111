#!/usr/bin/env python3
"""msj_stage1 (host pointers, pageable buffers) on valid documents of 8 .. 256 MiB: ms and GB/s.  Run once as is and
once with MSJ_PIPE_DISABLE=1 (plain staging) to see where the chunked pipeline starts to pay.  The knob exists in the
measurement build only (make -C mojo_simdjson_amd/csrc knobs), which this script loads."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import _lib, synth  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmsj_stage1_knobs.so")
lib = _lib.load()
lib.msj_debug_set_pipeline_min_bytes(None, 24 << 20)
mode = "plain staging" if os.environ.get("MSJ_PIPE_DISABLE") else "pipeline from 24 MiB"
for mib in [int(x) for x in os.environ.get("MSJ_SIZES_MIB", "8,16,23,24,32,48,64,96,128,256").split(",")]:
    data = synth.workload("minified", mib << 20).tobytes()
    idx = np.zeros(len(data) + 3, dtype=np.uint32)
    n = ctypes.c_uint64(0)

    def call():
        return lib.msj_stage1(data, len(data), idx.ctypes.data_as(ctypes.c_void_p), idx.size, ctypes.byref(n), None, 0)

    call()
    reps = 8 if mib <= 64 else 4
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = call()
    dt = (time.perf_counter() - t0) / reps
    print(f"{mode:22s} {len(data) / (1 << 20):7.1f} MiB: rc {rc}, {dt * 1e3:7.3f} ms, {len(data) / dt / 1e9:6.2f} GB/s", flush=True)
