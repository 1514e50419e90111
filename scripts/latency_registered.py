import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from mojo_simdjson_amd import _lib, synth
lib = _lib.load()
lib.msj_host_register.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
lib.msj_host_unregister.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
big = synth.workload("minified", 64 << 20)
for nbytes in (258381, 1045197, 4190541, 16773581, 33554432, 67108864 - 4096):
    data = np.ascontiguousarray(big[:nbytes]).copy()
    # make it a valid-ish prefix: fine either way, codes are not checked here
    idx = np.zeros(nbytes + 3, dtype=np.uint32)
    n = ctypes.c_uint64(0)
    def call():
        return lib.msj_stage1(data.ctypes.data_as(ctypes.c_void_p), nbytes, idx.ctypes.data_as(ctypes.c_void_p), idx.size, ctypes.byref(n), None, 0)
    def timeit():
        call(); reps = 20 if nbytes < (8 << 20) else 6
        t0 = time.perf_counter()
        for _ in range(reps): rc = call()
        return (time.perf_counter() - t0) / reps, rc
    t_plain, rc = timeit()
    assert lib.msj_host_register(None, idx.ctypes.data_as(ctypes.c_void_p), idx.nbytes) == 0
    assert lib.msj_host_register(None, data.ctypes.data_as(ctypes.c_void_p), data.nbytes) == 0
    t_reg, rc2 = timeit()
    lib.msj_host_unregister(None, data.ctypes.data_as(ctypes.c_void_p)); lib.msj_host_unregister(None, idx.ctypes.data_as(ctypes.c_void_p))
    print(f"{nbytes:10d} B: pageable {t_plain*1e6:8.1f} us ({nbytes/t_plain/1e9:5.2f} GB/s)   registered {t_reg*1e6:8.1f} us ({nbytes/t_reg/1e9:5.2f} GB/s)  rc {rc}/{rc2}", flush=True)
