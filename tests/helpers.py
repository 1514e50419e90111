"""Shared test plumbing.  The oracle (oracle/libmsj_oracle.so) is loaded HERE,
in tests/, and nowhere in the product package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "jsons_for_test")
ORACLE_SO = os.path.join(ROOT, "oracle", "libmsj_oracle.so")

_ARGS = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
         ctypes.POINTER(ctypes.c_uint64)]


def load_oracle():
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(ORACLE_SO)
    for name in ("msj_oracle_stage1", "msj_oracle_stage1_serial"):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int32
        fn.argtypes = _ARGS
    lib.msj_oracle_utf8.restype = ctypes.c_int32
    lib.msj_oracle_utf8.argtypes = [ctypes.c_char_p, ctypes.c_uint64]
    lib.msj_oracle_classify_byte.restype = ctypes.c_int
    lib.msj_oracle_classify_byte.argtypes = [ctypes.c_uint8, ctypes.c_int]
    return lib


FAST_SO = os.path.join(ROOT, "oracle", "libmsj_oracle_fast.so")


def load_oracle_fast():
    """The optimised CPU variants (oracle/stage1_fast.c): measurement infrastructure, not the reference."""
    if not os.path.exists(FAST_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(FAST_SO)
    lib.msj_fast_stage1.restype = ctypes.c_int32
    lib.msj_fast_stage1.argtypes = _ARGS
    lib.msj_fast_stage1_mt.restype = ctypes.c_int32
    lib.msj_fast_stage1_mt.argtypes = _ARGS + [ctypes.c_int32]
    return lib


TOKENS_SO = os.path.join(ROOT, "oracle", "libmsj_oracle_tokens.so")


def oracle_tokens(data, idx):
    """Definition of the token pre-pass (oracle/tokens_oracle.c): (type uint8[n], depth int32[n], (final, min, max))."""
    if not os.path.exists(TOKENS_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(TOKENS_SO)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    n = idx.size
    typ = np.zeros(max(n, 1), dtype=np.uint8)
    dep = np.zeros(max(n, 1), dtype=np.int32)
    res = (ctypes.c_uint8 * 24)()
    lib.msj_oracle_tokens(ctypes.c_char_p(bytes(data)), idx.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n),
                          typ.ctypes.data_as(ctypes.c_void_p), dep.ctypes.data_as(ctypes.c_void_p), res)
    r = np.frombuffer(bytes(res), dtype=np.int32)
    return typ[:n], dep[:n], (int(r[2]), int(r[3]), int(r[4]))


def oracle_match(typ):
    """Partner index of every bracket (oracle/tokens_oracle.c: msj_oracle_match), uint32, 0xFFFFFFFF = none."""
    lib = ctypes.CDLL(TOKENS_SO)
    typ = np.ascontiguousarray(typ, dtype=np.uint8)
    m = np.zeros(max(typ.size, 1), dtype=np.uint32)
    assert lib.msj_oracle_match(typ.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(typ.size),
                                m.ctypes.data_as(ctypes.c_void_p)) == 0
    return m[:typ.size]


def oracle_token_spans(data, idx):
    """(end uint32[n], flags uint8[n]) per oracle/tokens_oracle.c: msj_oracle_token_spans."""
    lib = ctypes.CDLL(TOKENS_SO)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    end = np.zeros(max(idx.size, 1), dtype=np.uint32)
    flags = np.zeros(max(idx.size, 1), dtype=np.uint8)
    lib.msj_oracle_token_spans(ctypes.c_char_p(bytes(data)), ctypes.c_uint64(len(data)), idx.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_uint64(idx.size), end.ctypes.data_as(ctypes.c_void_p), flags.ctypes.data_as(ctypes.c_void_p))
    return end[:idx.size], flags[:idx.size]


def ref_parse_number_scan(data, start, pad=0x20):
    """oracle/tokens_oracle.c msj_ref_parse_number_scan: the reference's parse_number scan
    (number_parsing.mojo:41-59) -> (code 0 / 9, end offset, is_float)."""
    lib = ctypes.CDLL(TOKENS_SO)
    end, flt = ctypes.c_uint64(0), ctypes.c_int(0)
    lib.msj_ref_parse_number_scan.restype = ctypes.c_int
    rc = lib.msj_ref_parse_number_scan(ctypes.c_char_p(bytes(data)), ctypes.c_uint64(len(data)), ctypes.c_uint64(start),
                                       ctypes.c_uint8(pad), ctypes.byref(end), ctypes.byref(flt))
    return rc, int(end.value), bool(flt.value)


def ref_parse_string_end(data, body_start, bytes_processed=8, pad=0x20):
    """oracle/tokens_oracle.c msj_ref_parse_string_end: the reference's parse_string terminator search
    (string_parsing.mojo:334-386) -> (offset of the closing quote or -1, escaped)."""
    lib = ctypes.CDLL(TOKENS_SO)
    esc = ctypes.c_int(0)
    lib.msj_ref_parse_string_end.restype = ctypes.c_int64
    e = lib.msj_ref_parse_string_end(ctypes.c_char_p(bytes(data)), ctypes.c_uint64(len(data)), ctypes.c_uint64(body_start),
                                     ctypes.c_uint8(pad), ctypes.c_int(bytes_processed), ctypes.byref(esc))
    return int(e), bool(esc.value)


def oracle_documents(data, idx, typ, dep, open_string=False, capacity=None, is_final=False):
    """Definition of the document split (oracle/tokens_oracle.c: msj_oracle_documents):
    (doc_first uint32[min(n_documents, capacity)], (n_documents, n_complete, tokens_complete, resume_offset))."""
    lib = ctypes.CDLL(TOKENS_SO)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    typ = np.ascontiguousarray(typ, dtype=np.uint8)
    dep = np.ascontiguousarray(dep, dtype=np.int32)
    n = idx.size
    cap = n if capacity is None else capacity
    first = np.zeros(max(cap, 1), dtype=np.uint32)
    res = (ctypes.c_uint64 * 4)()
    lib.msj_oracle_documents(ctypes.c_char_p(bytes(data)), ctypes.c_uint64(len(data)), ctypes.c_int(int(is_final)),
                             idx.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n), typ.ctypes.data_as(ctypes.c_void_p),
                             dep.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(int(open_string)),
                             first.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(cap), res)
    r = tuple(int(v) for v in res)
    return first[: min(r[0], cap)], r


def oracle_find_next_document_index(data, idx, open_string=False):
    """Upstream simdjson's backward scan, restated (oracle/tokens_oracle.c): (kept structurals, error)."""
    lib = ctypes.CDLL(TOKENS_SO)
    lib.msj_oracle_find_next_document_index.restype = ctypes.c_uint64
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    err = ctypes.c_int(0)
    keep = lib.msj_oracle_find_next_document_index(ctypes.c_char_p(bytes(data)), idx.ctypes.data_as(ctypes.c_void_p),
                                                   ctypes.c_uint64(idx.size), ctypes.c_int(int(open_string)), ctypes.byref(err))
    return int(keep), int(err.value)


def oracle_window(fn, data):
    """Structural indices of a WINDOW of a stream (nothing is an error yet at its end): the stage-1 oracle
    on the window, with the string it may end in closed by hand (bytes inside a string are never structural,
    the closing quote is not either).  -> (idx uint32[n], open_string)."""
    data = bytes(data)
    for k, tail in enumerate((b"", b'"', b'\\"')):
        rc, n, idx = run_oracle(fn, data + tail)
        if rc == 15:
            continue
        if rc == 13:
            return np.zeros(0, dtype=np.uint32), k > 0
        assert rc == 0, rc
        keep = idx[:n]
        assert n == 0 or keep[-1] < len(data)
        return keep.copy(), k > 0
    raise AssertionError("window does not close")


SENTINEL = 0xDEADBEEF


def run_oracle(fn, data):
    """-> (code, n or None, idx array incl. trailer or None)."""
    data = bytes(data)
    idx = np.full(len(data) + 3, SENTINEL, dtype=np.uint32)
    n = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
    rc = fn(data, len(data), idx.ctypes.data, idx.size, ctypes.byref(n))
    if n.value == 0xFFFFFFFFFFFFFFFF:
        return rc, None, None
    return rc, int(n.value), idx[: n.value + 3].copy()


def read_fixture(path):
    """tests/test_stage_1.mojo:85-89: line 0 = JSON text, line 1 = expected mask."""
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    return lines[0], lines[1]


def mask_from_indices(indices, width):
    """tests/test_stage_1.mojo:52-58."""
    m = bytearray(b" " * width)
    for i in indices:
        m[int(i)] = ord("1")
    return bytes(m)


def golden_valid_files():
    d = os.path.join(GOLDEN, "valid")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".json"))


def fuzz_inputs(seed, count, max_len=700):
    """JSON-punctuation-heavy adversarial byte strings."""
    import random

    rng = random.Random(seed)
    alpha = (b'\\\\\\"""[]{}:, \n\tabc019-\x0c\x1a\x00\x01\x1f\x20\xc3\xa9\xe4\xb8\xad'
             b'\xf0\x9f\x98\x80tfn\x7f\x80\xff')
    lens = [1, 2, 5, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 300]
    for it in range(count):
        n = rng.choice(lens + [rng.randint(1, max_len)])
        if it % 3 == 0:
            yield bytes(rng.getrandbits(8) for _ in range(n))
        else:
            yield bytes(rng.choice(alpha) for _ in range(n))
