#!/usr/bin/env python3
"""msj_stage1 (host pointers, pageable caller memory, 256 MiB minified) against WHERE the caller's buffers lie and how
many copy workers the pipeline runs: the caller's pages are first-touched from a CPU of the GPU's NUMA node or of the
other one (sched_setaffinity around the allocation), the worker count comes from MSJ_PIPE_THREADS of the measurement
build (make -C mojo_simdjson_amd/csrc knobs; the product reads no environment variable).  One process per setting.
    python3 scripts/e2e_placement.py            the sweep (starts the children)
    python3 scripts/e2e_placement.py child <near|far>"""
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def node_cpus():
    nodes = {}
    base = "/sys/devices/system/node"
    for d in sorted(os.listdir(base)):
        if d.startswith("node") and d[4:].isdigit():
            cpus = set()
            for part in open(os.path.join(base, d, "cpulist")).read().strip().split(","):
                if part:
                    a, _, b = part.partition("-")
                    cpus |= set(range(int(a), int(b or a) + 1))
            nodes[int(d[4:])] = cpus
    return nodes


def child(where):
    import numpy as np

    from mojo_simdjson_amd import _lib, synth

    _lib.LIB_PATH = os.path.join(ROOT, "scripts", "libmsj_stage1_knobs.so")
    lib = _lib.load()
    text = ctypes.create_string_buffer(1024)
    assert lib.msj_host_placement(None, text, 1024) == 0
    gpu_node = json.loads(text.value.decode())["gpu_numa_node"]
    nodes = node_cpus()
    allowed = os.sched_getaffinity(0)
    want = gpu_node if where == "near" else next((k for k in nodes if k != gpu_node), gpu_node)
    cpus = (nodes.get(want, allowed) & allowed) or allowed
    os.sched_setaffinity(0, cpus)  # the caller's thread, and with it the first touch of its buffers
    u = synth.workload("minified", 256 << 20)
    data = np.ascontiguousarray(u).copy()
    idx = np.zeros(data.size + 3, dtype=np.uint32)
    n, verdict = ctypes.c_uint64(0), ctypes.c_int32(0)

    def call():
        return lib.msj_stage1(data.ctypes.data_as(ctypes.c_void_p), data.size, idx.ctypes.data_as(ctypes.c_void_p), idx.size, ctypes.byref(n),
                              ctypes.byref(verdict), 0)

    assert call() == 0
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        assert call() == 0
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({"caller_buffers": where, "caller_node": int(lib.msj_debug_numa_node_of(data.ctypes.data + data.size // 2)),
                      "index_node": int(lib.msj_debug_numa_node_of(idx.ctypes.data + idx.nbytes // 4)), "gpu_node": gpu_node,
                      "threads": os.environ.get("MSJ_PIPE_THREADS", "8"), "parts": os.environ.get("MSJ_PIPE_PARTS", "4"),
                      "median_gbps": round(data.size / ts[len(ts) // 2] / 1e9, 2), "best_gbps": round(data.size / ts[0] / 1e9, 2)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2])
    else:
        for where in ("near", "far"):
            for threads, parts in ((8, 4), (12, 4), (16, 4), (16, 8), (24, 8)):
                env = dict(os.environ, MSJ_PIPE_THREADS=str(threads), MSJ_PIPE_PARTS=str(parts))
                subprocess.run([sys.executable, os.path.abspath(__file__), "child", where], env=env, check=False, timeout=300)
