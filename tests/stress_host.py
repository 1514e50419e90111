#!/usr/bin/env python3
"""Randomised parity stress of the host-pointer entry point on the GPU box (not part of pytest: runs for minutes).

Byte soups of tests/stress.py at sizes on both sides of every switch of msj_stage1 (the zero-copy pinned path up to 1 MiB,
plain staging, the chunked pipeline -- set to start at 24 MiB here -- and its 16 MiB chunks), through ONE reused DomParserImplementation -- its
index array grows, gets pinned (msj_host_register) and is reused, like the reference's list -- and through the C entry
point with fresh pageable arrays.  Code, count, every index and the trailer against the oracle.
usage: tests/stress_host.py [seconds] [seed]
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from tests.stress import soup  # noqa: E402
from mojo_simdjson_amd import _lib  # noqa: E402
from mojo_simdjson_amd.dom_parser_implementation import DomParserImplementation  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    oracle = helpers.load_oracle()
    lib = _lib.load()
    lib.msj_debug_set_pipeline_min_bytes(None, 24 << 20)  # (default 64 MiB) both paths within the sizes below
    parser = DomParserImplementation()
    mib = 1 << 20
    sizes = [1, 63, 4097, 65535, 65536, 65537, mib - 1, mib, mib + 1, 3 * mib + 5, 16 * mib - 1, 24 * mib - 1, 24 * mib, 24 * mib + 1,
             32 * mib, 32 * mib + 4099, 40 * mib + 17, 48 * mib]
    t0 = time.time()
    cases = nbytes = 0
    while time.time() - t0 < budget:
        n = int(rng.choice(sizes)) if rng.random() < 0.8 else int(rng.integers(1, 30 * mib))
        data = soup(rng, n)
        if rng.random() < 0.3:  # mostly valid-looking: long stretches without an error
            data = (b'{"k":["v\\\\"w",1.5,true],"s":"' + data[: n // 3].replace(b'"', b"a").replace(b"\\", b"b") + b'"} ') * 2
            data = data[:n] if len(data) >= n else data
            n = len(data)
        idx = np.full(n + 3, helpers.SENTINEL, dtype=np.uint32)
        nn = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
        code = oracle.msj_oracle_stage1(data, n, idx.ctypes.data, idx.size, ctypes.byref(nn))
        tag = f"case {cases} (seed {seed}, len {n}, code {code})"
        # (a) the reused parser
        rc = parser.stage1(data)
        assert rc == code, f"{tag}: parser code {rc}"
        if code in (0, 13):
            k = int(nn.value)
            assert parser.n_structural_indexes == k, tag
            assert np.array_equal(parser.structural_indexes[: k + 3], idx[: k + 3]), f"{tag}: parser indices"
        # (b) the C entry point, fresh pageable arrays
        cap = n + 3
        out = np.full(cap, helpers.SENTINEL, dtype=np.uint32)
        got = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
        want = np.full(cap, helpers.SENTINEL, dtype=np.uint32)
        wn = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
        wcode = oracle.msj_oracle_stage1(data, n, want.ctypes.data, want.size, ctypes.byref(wn))
        rc = lib.msj_stage1(data, n, out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(got), None, 0)
        assert rc == wcode, f"{tag}: cap {cap}: code {rc} != {wcode}"
        if wcode in (0, 13):
            k = int(wn.value)
            assert got.value == k, f"{tag}: cap {cap}"
            assert np.array_equal(out[: k + 3], want[: k + 3]), f"{tag}: cap {cap}: indices"
        cases += 1
        nbytes += n
        if cases % 10 == 0:
            print(f"{cases} cases, {nbytes / 1e6:.0f} MB, {time.time() - t0:.0f} s", flush=True)
    print(f"stress_host ok: seed {seed}, {cases} cases, {nbytes / 1e6:.0f} MB through msj_stage1 twice each")


if __name__ == "__main__":
    main()
