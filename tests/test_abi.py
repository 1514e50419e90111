"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/msj_stage1.h declares; without a GPU the product fails loudly
(no CPU fallback, and nothing in the package reaches for oracle/)."""
import ctypes
import os
import re

import pytest

from tests import helpers

HEADER = os.path.join(helpers.ROOT, "include", "msj_stage1.h")
PKG = os.path.join(helpers.ROOT, "mojo_simdjson_amd")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(msj_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from mojo_simdjson_amd import _lib

    lib = _lib.load()
    names = declared_functions()
    assert {"msj_stage1", "msj_stage1_device", "msj_stage1_shard_device", "msj_ctx_create",
            "msj_ctx_destroy", "msj_carry_fetch", "msj_device_count"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/msj_stage1.h but not exported"
    assert lib.msj_tile_bytes() == 4096
    assert b"gfx950" in lib.msj_version()
    # ... and the Python binding declares the arguments of every one of them: an undeclared Python int goes over as a
    # 32-bit C int (a 3.94 GiB buffer length arrived truncated at the round-5 entry points until they were declared)
    from mojo_simdjson_amd import sharded

    sharded.lib()  # (the N-GPU entry points are declared there)
    called = set()
    for f in os.listdir(PKG):
        if f.endswith(".py"):
            called |= set(re.findall(r"\.(msj_[a-z0-9_]+)\(", open(os.path.join(PKG, f)).read()))
    undeclared = sorted(n for n in called & set(names) if getattr(lib, n).argtypes is None)
    assert not undeclared, f"the package calls these without declared argtypes: {undeclared}"


def test_token_workspace_sizes():
    """The token calls' workspace (host arithmetic, no GPU): grows with the token count, the match[] form adds the min
    tree and the lists of brackets left to it, the pairs form the compact bracket list on top (a token and a depth word
    per token of capacity, csrc/tokens_kernel.hip)."""
    from mojo_simdjson_amd import _lib

    lib = _lib.load()
    f = lib.msj_stage2_prep_workspace_bytes
    f.restype = ctypes.c_uint64
    f.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
    prev = (0, 0, 0)
    for n in (0, 1, 7, 2047, 2048, 2049, 1 << 20, (1 << 27) + 5, (1 << 31) - 1):
        plain, match, pairs = (f(n, 6 * n + 1, m) for m in (0, 1, 2))
        assert 0 < plain <= match <= pairs and plain % 4 == 0 and match % 4 == 0 and pairs % 4 == 0
        assert pairs - match >= 8 * n  # the compact list
        assert match - plain >= 4 * n  # lists of one word per bracket a block may leave over
        assert (plain, match, pairs) >= prev
        prev = (plain, match, pairs)
    assert f((1 << 31) - 1, (1 << 32) - 1, 2) < 40 * (1 << 31)  # (a bound a caller can budget with: < 40 bytes per token)


def test_struct_layouts():
    from mojo_simdjson_amd import _lib

    assert ctypes.sizeof(_lib.MsjCarry) == 64
    assert _lib.MsjCarry.count.offset == 0 and _lib.MsjCarry.in_string.offset == 16
    assert _lib.MsjCarry.code.offset == 40 and _lib.MsjCarry.capacity_error.offset == 44
    assert ctypes.sizeof(_lib.MsjSegment) == 32
    from mojo_simdjson_amd import sharded

    assert ctypes.sizeof(sharded.MsjShardPlacement) == 32 and ctypes.sizeof(sharded.MsjShardedStats) == 96
    assert ctypes.sizeof(sharded.MsjShardReport) == 128  # what one rank contributes to the all-gather
    # the ctypes mirrors against the header itself: a C program prints sizeof / offsetof of what the bindings restate
    import subprocess

    out = os.path.join(helpers.ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    src = os.path.join(out, "layout.c")
    probes = [("msj_carry", _lib.MsjCarry, ["count", "in_string", "code", "capacity_error"]),
              ("msj_segment", _lib.MsjSegment, ["count"]),
              ("msj_tokens_result", _lib.MsjTokensResult, ["max_depth"]),
              ("msj_documents_result", _lib.MsjDocumentsResult, ["resume_offset"]),
              ("msj_shard_report", sharded.MsjShardReport, ["out"]),
              ("msj_exchange", sharded.MsjExchange, ["allgather", "rank", "owns_comm"]),
              ("msj_shard_placement", sharded.MsjShardPlacement, ["bytes"]),
              ("msj_sharded_stats", sharded.MsjShardedStats, ["kernel_device_ns", "last_stitch_ns", "reruns_behind_queue"]),
              ("msj_sharded_ops", sharded.MsjShardedOps, ["run_shard", "event_create", "stream_wait", "event_elapsed_ns",
                                                          "side_stream"])]
    with open(src, "w") as f:
        f.write('#include <stddef.h>\n#include <stdio.h>\n#include "msj_stage1.h"\nint main(void) {\n')
        for name, _, fields in probes:
            f.write(f'printf("%zu", sizeof({name}));')
            for fld in fields:
                f.write(f'printf(" %zu", offsetof({name}, {fld}));')
            f.write('printf("\\n");\n')
        f.write("return 0; }\n")
    exe = os.path.join(out, "layout")
    subprocess.check_call(["gcc", "-std=c11", "-I" + os.path.join(helpers.ROOT, "include"), src, "-o", exe])
    lines = subprocess.check_output([exe], text=True).split("\n")
    for (name, cls, fields), line in zip(probes, lines):
        want = [ctypes.sizeof(cls)] + [getattr(cls, fld).offset for fld in fields]
        assert [int(x) for x in line.split()] == want, (name, line, want)


def test_no_environment_switches():
    """the shipped package must not be able to load another kernel, or change its data path,
    because of an environment variable.  The library imports no getenv at all (the host pipeline's tuning knobs
    exist only in the -DMSJ_DEBUG_KNOBS measurement build) and the Python side never looks at os.environ."""
    import subprocess

    from mojo_simdjson_amd import _lib

    und = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH], text=True)
    assert "getenv" not in und, [l for l in und.splitlines() if "getenv" in l]
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(root, f)).read()
                assert "os.environ" not in text and "getenv(" not in text, os.path.join(root, f)
    assert _lib.LIB_PATH == os.path.join(PKG, "libmsj_stage1.so")
    assert re.search(rb"src:[0-9a-f]{12}$", _lib.load().msj_version())


def test_product_never_touches_oracle():
    """The product package must not import, load, link or mention the test oracle."""
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")) or f == "Makefile":
                text = open(os.path.join(root, f), errors="ignore").read().lower()
                assert "oracle" not in text, os.path.join(root, f)


def test_fails_loudly_without_gpu():
    from mojo_simdjson_amd import DomParserImplementation, _lib, errors

    lib = _lib.load()
    if lib.msj_device_count() > 0:
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    assert lib.msj_ctx_create(0, ctypes.byref(h)) == errors.ERR_NO_DEVICE
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DomParserImplementation().stage1("[1, 2]")


def test_error_values_match_reference():
    from mojo_simdjson_amd import errors

    # src/mojo_simdjson/errors.mojo:2-26
    assert (errors.SUCCESS, errors.CAPACITY, errors.UTF8_ERROR, errors.EMPTY,
            errors.UNESCAPED_CHARS, errors.UNCLOSED_STRING, errors.UNEXPECTED_ERROR) == \
        (0, 1, 11, 13, 14, 15, 24)


def test_cpp_mirror_compiles():
    """The C++ mirror of the reference facade and the C++ twin of its test build against the C ABI
    (no GPU needed to compile and link; tests/test_stage1_gpu.py runs it)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    libdir = os.path.join(root, "mojo_simdjson_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "test_stage_1.cpp"), "-o", os.path.join(out, "test_stage_1"),
                           "-L" + libdir, "-lmsj_stage1", "-Wl,-rpath," + libdir])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "test_document_stream.cpp"), "-o",
                           os.path.join(out, "test_document_stream"), "-L" + libdir, "-lmsj_stage1", "-Wl,-rpath," + libdir])
