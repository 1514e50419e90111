"""Host-pointer entry point (msj_stage1): rate including H2D of the input and D2H of the indices.

Two figures: the C entry point with caller-owned, already touched buffers (what a Mojo shim
with a reused parser sees), and the Python mirror of the reference facade, which like the
reference (dom_parser_implementation.mojo:85-89) allocates and zero-fills the index array on
every call."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import DomParserImplementation, _lib, synth  # noqa: E402

u = synth.workload("minified", 256 << 20)
data = u.tobytes()
lib = _lib.load()
idx = np.zeros(len(data) + 3, dtype=np.uint32)  # touched once, reused
n = ctypes.c_uint64(0)
verdict = ctypes.c_int32(0)


def call():
    return lib.msj_stage1(data, len(data), idx.ctypes.data_as(ctypes.c_void_p), idx.size, ctypes.byref(n),
                          ctypes.byref(verdict), 0)


call()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    rc = call()
dt = (time.perf_counter() - t0) / reps
print(f"msj_stage1 (C entry point, reused buffers): {len(data)} B, rc {rc}, n {n.value}, {dt*1e3:.1f} ms, "
      f"{len(data)/dt/1e9:.2f} GB/s of JSON (H2D {len(data)/1e6:.0f} MB + D2H {(n.value+3)*4/1e6:.0f} MB over PCIe)")
p = DomParserImplementation()
p.stage1(data)
t0 = time.perf_counter()
for _ in range(3):
    rc = p.stage1(data)
dt = (time.perf_counter() - t0) / 3
print(f"DomParserImplementation.stage1 (Python mirror, allocates like the reference): {dt*1e3:.1f} ms, {len(data)/dt/1e9:.2f} GB/s")
