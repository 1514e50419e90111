"""oracle/stage1_fast.c (the "best-case CPU" figures of bench.py) gives exactly what the
reference-faithful oracle gives: same indices, count, trailer and return code."""
import ctypes
import functools
import os

import numpy as np
import pytest

from tests import helpers


@pytest.fixture(scope="module")
def libs():
    return helpers.load_oracle(), helpers.load_oracle_fast()


def _same(libs, data, threads=(1, 2, 3, 7, 16)):
    oracle, fast = libs
    want = helpers.run_oracle(oracle.msj_oracle_stage1, data)
    got = helpers.run_oracle(fast.msj_fast_stage1, data)
    assert got[0] == want[0] and got[1] == want[1]
    if want[1] is not None:
        assert np.array_equal(got[2], want[2])
    for t in threads:
        fn = functools.partial(fast.msj_fast_stage1_mt)
        got = helpers.run_oracle(lambda b, n, i, c, o, t=t: fn(b, n, i, c, o, t), data)
        assert got[0] == want[0] and got[1] == want[1], (t, got[:2], want[:2])
        if want[1] is not None:
            assert np.array_equal(got[2], want[2]), t


def test_golden_fixtures(libs):
    for f in helpers.golden_valid_files():
        js, _ = helpers.read_fixture(f)
        _same(libs, js)


def test_every_byte_class(libs):
    oracle, fast = libs
    for b in range(256):
        _same(libs, bytes([b]) * 3 + b" 1", threads=(1, 2))
        _same(libs, b'"' + bytes([b]) + b'" ', threads=(1,))


def test_fuzz(libs):
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b'{}[],: \n"\\\\ab1\x01\x0c\x1a\xc3\xa9', dtype=np.uint8)
    for k in range(400):
        n = int(rng.integers(1, 700))
        _same(libs, alphabet[rng.integers(0, len(alphabet), n)].tobytes(), threads=(1, 2, 5, 11))
    # chunk boundaries inside backslash runs and right after quotes
    for run in (1, 2, 63, 64, 65, 127, 128, 129, 200):
        _same(libs, b'["' + b"\\" * run + b'"x", 1]' + b" " * 70, threads=(2, 3, 4, 5, 6))
        _same(libs, b" " * 62 + b'"' + b"\\" * run + b'\\"' + b'", 2' + b" " * 200, threads=(2, 3, 4, 5, 6))


def test_workloads(libs):
    from mojo_simdjson_amd import synth

    for name in ("minified", "utf8", "pretty4"):
        _same(libs, synth.workload(name, 2 << 20).tobytes(), threads=(1, 8))
