"""Host-pointer entry point (msj_stage1): rate including H2D of the input and D2H of the indices.

The C entry point with caller-owned, already touched pageable buffers; the same with the index array
(and then the input as well) pinned once through msj_host_register; and the Python mirror of the
reference facade with a reused parser: like the reference's list (dom_parser_implementation.mojo:85-89,
reserve + resize) its index array is allocated where a document is larger than any before, and pinned there."""
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
# the same with the index array pinned once (msj_host_register: what the shim does in allocate(), the reference
# allocates structural_indexes once per parser), and with the input buffer pinned as well (a host that reuses it)
lib.msj_host_register.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
lib.msj_host_unregister.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
inbuf = np.frombuffer(data, dtype=np.uint8)
t0 = time.perf_counter()
assert lib.msj_host_register(None, idx.ctypes.data_as(ctypes.c_void_p), idx.nbytes) == 0
t_reg = time.perf_counter() - t0
for label, with_input in (("index array registered", False), ("index array and input registered", True)):
    if with_input:
        assert lib.msj_host_register(None, inbuf.ctypes.data_as(ctypes.c_void_p), inbuf.nbytes) == 0
    call()
    t0 = time.perf_counter()
    for _ in range(reps):
        rc = call()
    dt = (time.perf_counter() - t0) / reps
    print(f"msj_stage1, {label}: rc {rc}, n {n.value}, {dt*1e3:.1f} ms, {len(data)/dt/1e9:.2f} GB/s of JSON"
          + (f" (registering {idx.nbytes/1e6:.0f} MB took {t_reg*1e3:.1f} ms, once)" if not with_input else ""))
assert lib.msj_host_unregister(None, inbuf.ctypes.data_as(ctypes.c_void_p)) == 0
assert lib.msj_host_unregister(None, idx.ctypes.data_as(ctypes.c_void_p)) == 0
p = DomParserImplementation()
p.stage1(data)
t0 = time.perf_counter()
for _ in range(3):
    rc = p.stage1(data)
dt = (time.perf_counter() - t0) / 3
print(f"DomParserImplementation.stage1 (Python mirror, reused parser: index array allocated and pinned once): {dt*1e3:.1f} ms, {len(data)/dt/1e9:.2f} GB/s")
