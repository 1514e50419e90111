"""CPU check of the kernel's per-lane bit math (mojo_simdjson_amd/csrc/lane_math.h).

The header is compiled for the host with g++ (tests/lane_math_host.cpp chains
its per-block functions sequentially) and compared with the oracle: this
validates the bit-plane transpose, the plane-logic character classes, the
escape / in-string / follows formulas and the UTF-8 plane validator without a
GPU.  The cross-lane / cross-tile carry resolution only exists on the device
and is covered by the -m gpu tests.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import helpers

BUILD = os.path.join(helpers.ROOT, "tests", "_build")


@pytest.fixture(scope="module")
def lane():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "liblane_math_host.so")
    src = os.path.join(helpers.ROOT, "tests", "lane_math_host.cpp")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.lane_stage1.restype = ctypes.c_int32
    lib.lane_stage1.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int32)]
    lib.lane_bitplanes.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    lib.lane_top_run.restype = ctypes.c_uint32
    lib.lane_top_run.argtypes = [ctypes.c_uint64]
    lib.lane_prefix_xor.restype = ctypes.c_uint64
    lib.lane_prefix_xor.argtypes = [ctypes.c_uint64]
    return lib


def run_lane(lib, d):
    idx = np.full(len(d) + 3, helpers.SENTINEL, dtype=np.uint32)
    n = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
    u = ctypes.c_int32(-1)
    rc = lib.lane_stage1(d, len(d), idx.ctypes.data, idx.size, ctypes.byref(n), ctypes.byref(u))
    if n.value == 0xFFFFFFFFFFFFFFFF:
        return rc, None, None, u.value
    return rc, int(n.value), idx[: n.value + 3].copy(), u.value


def test_bitplanes_exhaustive(lane):
    allb = bytes(range(256))
    for s in range(0, 256, 64):
        b = allb[s:s + 64]
        pl = (ctypes.c_uint64 * 8)()
        lane.lane_bitplanes(b, pl)
        for k in range(8):
            assert pl[k] == sum(((b[i] >> k) & 1) << i for i in range(64))
    rng = np.random.default_rng(3)
    for _ in range(200):
        b = rng.integers(0, 256, 64, dtype=np.uint8).tobytes()
        pl = (ctypes.c_uint64 * 8)()
        lane.lane_bitplanes(b, pl)
        for k in range(8):
            assert pl[k] == sum(((b[i] >> k) & 1) << i for i in range(64))


def test_span_classes_exhaustive(lane):
    """the token-span kernel's classes against the reference's tables, every byte value at every position parity"""
    lane.lane_span_classes.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    lane.lane_tile_classes.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64)]
    sow = set(b"\t\n\r ,:[]{}")
    blank = set(b"\t\n\r ")
    want = lambda b, cls: sum(1 << i for i in range(64) if b[i] in cls)
    rng = np.random.default_rng(5)
    blocks = [bytes(range(s, s + 64)) for s in range(0, 256, 64)]
    blocks += [rng.integers(0, 256, 64, dtype=np.uint8).tobytes() for _ in range(300)]
    blocks += [bytes(rng.choice(np.frombuffer(b'0123456789-.eE,:[]{}"\\ \t\n\r\x0c\x1a;Z', dtype=np.uint8), 64)) for _ in range(300)]
    for b in blocks:
        out = (ctypes.c_uint64 * 4)()
        lane.lane_span_classes(b, out)
        assert out[0] == want(b, set(b"0123456789"))
        assert out[1] == want(b, sow)
        assert out[2] == want(b, set(b"\\"))
        assert out[3] == want(b, blank)
        out2 = (ctypes.c_uint64 * 2)()
        lane.lane_tile_classes(b, out2)
        assert out2[0] == want(b, set(b'"'))
        assert out2[1] == want(b, set(b".eE"))


def test_small_helpers(lane):
    assert lane.lane_top_run(0) == 0
    assert lane.lane_top_run(0xFFFFFFFFFFFFFFFF) == 64
    assert lane.lane_top_run(0xE000000000000000) == 3
    assert lane.lane_top_run(0x7FFFFFFFFFFFFFFF) == 0
    rng = np.random.default_rng(4)
    for v in rng.integers(0, 2**63, 200, dtype=np.uint64):
        v = int(v) * 2 + int(v & 1)
        want, acc = 0, 0
        for i in range(64):
            acc ^= (v >> i) & 1
            want |= acc << i
        assert lane.lane_prefix_xor(v & (2**64 - 1)) == want


def test_golden(lane):
    for f in helpers.golden_valid_files():
        js, mask = helpers.read_fixture(f)
        rc, n, idx, _ = run_lane(lane, js)
        assert rc == 0
        assert helpers.mask_from_indices(idx[:n], len(mask)) == mask
        assert list(idx[n:n + 3]) == [len(js), len(js), 0]


def test_fuzz_vs_oracle(lane, oracle):
    for d in helpers.fuzz_inputs(99, 12000):
        a = helpers.run_oracle(oracle.msj_oracle_stage1, d)
        b = run_lane(lane, d)
        assert a[0] == b[0], d
        assert a[1] == b[1], d
        if a[1] is not None:
            assert np.array_equal(a[2], b[2]), d
        assert b[3] == oracle.msj_oracle_utf8(d, len(d)), d


def test_utf8_text_and_truncations(lane, oracle):
    import random

    rng = random.Random(11)
    chars = "aé中😀߿ࠀ￿\U00010000\U0010ffff\"\\ "
    for _ in range(3000):
        s = "".join(rng.choice(chars) for _ in range(rng.randint(1, 90))).encode()
        k = rng.randint(1, len(s))
        d = s[:k]
        assert run_lane(lane, d)[3] == oracle.msj_oracle_utf8(d, len(d)), d


def test_utf8_every_pair_at_every_block_position(lane, oracle):
    """The second-byte rules (E0 / ED / F0 / F4) are judged at the LEAD byte from the next byte's bits, the structure
    from lead planes moved forward: every (lead, second) pair and every 3- / 4-byte shape with its second byte at the
    extremes, placed so that each of its bytes in turn is the first byte of a 64-byte block, the last one, or the last
    byte of the input -- against the oracle's validator (itself pinned to CPython's strict decoder)."""
    seqs = []
    for a in range(0x80, 0x100):
        for b in (0x00, 0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC2, 0xE0, 0xED, 0xF0, 0xF4, 0xFF):
            seqs.append(bytes([a, b]))
            seqs.append(bytes([a, b, 0x80]))
            seqs.append(bytes([a, b, 0xBF, 0x80]))
            seqs.append(bytes([a, b, 0x80, 0x41]))
    for a in (0xE0, 0xE1, 0xEC, 0xED, 0xEE, 0xEF, 0xF0, 0xF1, 0xF3, 0xF4, 0xF5):
        for b in range(0x70, 0xD0):
            seqs += [bytes([a, b, 0x80, 0x80]), bytes([a, b, 0x80]), bytes([a, b])]
    bad = 0
    for s in seqs:
        for start in (61, 62, 63, 64, 127, 0):
            d = b"a" * start + s + b"bc"
            for cut in {len(d), start + len(s), start + len(s) - 1, start + 1}:
                dd = d[:cut]
                want = oracle.msj_oracle_utf8(dd, len(dd))
                got = run_lane(lane, dd)[3]
                if got != want:
                    bad += 1
                    assert bad < 5, (s.hex(), start, cut, got, want)
    assert bad == 0
