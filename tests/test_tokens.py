"""Token-stream pre-pass: type byte and nesting depth per structural.

Derived quantities (the reference has no such arrays, oracle/tokens_oracle.c is their definition);
the CPU part checks that definition by hand, the GPU part the HIP kernels against it."""
import ctypes

import numpy as np
import pytest

from tests import helpers


def _stage1(oracle, data):
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, data)
    assert n is not None
    return idx[:n]


def test_definition_by_hand():
    oracle = helpers.load_oracle()
    doc = b'{"a":[1,{"b":[]}],"c":{}}'
    idx = _stage1(oracle, doc)
    typ, dep, (final, mn, mx) = helpers.oracle_tokens(doc, idx)
    assert bytes(typ) == b'{":[1,{":[]}],":{}}'
    #                    {  "  :  [  1  ,  {  "  :  [  ]  }  ]  ,  "  :  {  }  }
    assert list(dep) == [0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 2, 1, 1, 1, 1, 1, 1, 0]
    assert (final, mn, mx) == (0, 0, 4)
    typ, dep, (final, mn, mx) = helpers.oracle_tokens(b"]]1[", np.array([0, 1, 2, 3], dtype=np.uint32))
    assert list(dep) == [-1, -2, -2, -2] and (final, mn, mx) == (-1, -2, -1)
    typ, dep, res = helpers.oracle_tokens(b"", np.zeros(0, dtype=np.uint32))
    assert typ.size == 0 and res == (0, 0, 0)
    N = 0xFFFFFFFF
    assert list(helpers.oracle_match(np.frombuffer(b'{":[1,{":[]}],":{}}', dtype=np.uint8))) == \
        [18, N, N, 12, N, N, 11, N, N, 10, 9, 6, 3, N, N, N, 17, 16, 0]
    assert list(helpers.oracle_match(np.frombuffer(b"][]][", dtype=np.uint8))) == [N, 2, 1, N, N]


def test_span_definition_by_hand():
    oracle = helpers.load_oracle()
    doc = b'{"a\\"b":-12.5e3,"c":[7,"x"],"d":"\\\\"}'
    idx = _stage1(oracle, doc)
    end, flags = helpers.oracle_token_spans(doc, idx)
    toks = [(chr(doc[i]), int(e), int(f)) for i, e, f in zip(idx, end, flags)]
    #       0{ 1" 2a 3\ 4" 5b 6" 7: 8- .. 14"3" 15, 16" 17c 18" 19: 20[ 21"7" 22, 23" 24x 25" 26] 27, 28" 29d 30" 31: 32" 33\ 34\ 35" 36}
    assert toks == [("{", 0, 0), ('"', 6, 3), (":", 0, 0), ("-", 15, 12), (",", 0, 0), ('"', 18, 1), (":", 0, 0),
                    ("[", 0, 0), ("7", 22, 4), (",", 0, 0), ('"', 25, 1), ("]", 0, 0), (",", 0, 0), ('"', 30, 1),
                    (":", 0, 0), ('"', 35, 3), ("}", 0, 0)], toks
    # an unclosed string, a string and a number beyond the scan cap
    d2 = b'["' + b"a" * 2000 + b'",' + b"1" * 2000 + b',"zz'
    idx2 = np.array([0, 1, 2003, 2004, 4004, 4005], dtype=np.uint32)
    end, flags = helpers.oracle_token_spans(d2, idx2)
    assert list(zip(end.tolist(), flags.tolist())) == [(0, 0), (2002, 1), (0, 0), (0, 132), (0, 0), (len(d2), 17)]
    # the escape flag has no cap (round 5): a body of 70 000 bytes whose only backslash is its last-but-one byte
    d3 = b'["' + b"a" * 69998 + b'\\n","' + b"b" * 70000 + b'"]'
    idx3 = _stage1(oracle, d3)
    end, flags = helpers.oracle_token_spans(d3, idx3)
    assert list(zip(idx3.tolist(), end.tolist(), flags.tolist())) == [(0, 0, 0), (1, 70002, 3), (70003, 0, 0), (70004, 140005, 1), (140006, 0, 0)]



STAGE2_DOCS = ("simple_json.json", "simple_strings.json", "escaping.json", "escaping_very_long.json")


def _check_spans_against_reference_scans(data, idx, end, flags):
    """Every string / number token: the span equals what the reference's own scans find (restated literally in
    oracle/tokens_oracle.c), and lies inside the token's extent [idx[i], idx[i+1])."""
    literal_differs = 0
    for i, start in enumerate(idx.tolist()):
        nxt = int(idx[i + 1]) if i + 1 < len(idx) else len(data)
        f, e = int(flags[i]), int(end[i])
        c = data[start]
        if c == 0x22:
            assert f & 1
            want, esc = helpers.ref_parse_string_end(data, start + 1, 8)
            if f & 16:  # open string: the reference has no terminator either
                continue
            assert want == e, (i, start, want, e)
            assert start < e < nxt or (e < len(data) and nxt == len(data))
            assert not f & 128  # MSJ_SPAN_LONG is for numbers only
            assert bool(f & 2) == esc, (i, start)
            lit, _ = helpers.ref_parse_string_end(data, start + 1, 32)
            literal_differs += lit != want
        elif c == 0x2D or 0x30 <= c <= 0x39:
            assert f & 4
            if f & 128:
                continue
            rc, want, flt = helpers.ref_parse_number_scan(data, start)
            assert want == e and bool(f & 8) == flt and bool(f & 32) == (rc == 9), (i, start, rc, want, flt, e, f)
            assert start < e <= nxt or bool(f & 32) or flt  # a float may run over a quote up to the next structural/blank
        else:
            assert f == 0 and e == 0
    return literal_differs


def test_spans_follow_the_reference_scans():
    """f2 / f4 anchored: oracle/tokens_oracle.c's definition agrees with its literal restatements of parse_number's scan
    (number_parsing.mojo:41-59) and of parse_string's terminator search (string_parsing.mojo:334-386, windows advancing
    by the 8 bytes examined) on the four documents the reference's stage 2 accepts (tests/test_stage_2.mojo:47-67), on
    the synthetic workloads, and -- numbers -- on malformed text, where the flag MSJ_SPAN_BAD marks exactly the
    reference's NUMBER_ERROR."""
    import os

    from mojo_simdjson_amd import synth

    oracle = helpers.load_oracle()
    differs = {}
    for name in STAGE2_DOCS:
        js, _ = helpers.read_fixture(os.path.join(helpers.GOLDEN, "valid", name))
        idx = _stage1(oracle, js)
        end, flags = helpers.oracle_token_spans(js, idx)
        differs[name] = _check_spans_against_reference_scans(js, idx, end, flags)
        typ, dep, (final, mn, mx) = helpers.oracle_tokens(js, idx)
        assert final == 0 and mn == 0, name  # what walk_document needs (json_iterator.mojo:84-90,173-180)
    # the reference advances a quote-less window by 32 although it examined 8 bytes: on the long fixture the
    # literal restatement misses closing quotes (recorded, not followed)
    assert differs["simple_json.json"] == 0 and differs["simple_strings.json"] == 0
    for name in ("minified", "utf8", "pretty4"):
        u = synth.workload(name, 1 << 20).tobytes()
        idx = _stage1(oracle, u)
        end, flags = helpers.oracle_token_spans(u, idx)
        _check_spans_against_reference_scans(u, idx, end, flags)
    bad = b'[12a,-,--1,1+2,1.5x,1e5,-0.5E-3,0x10,1.,12 ,3\t,4\n,5:6,7"a",1.5"b" ,9]'
    idx = _stage1(oracle, bad)
    end, flags = helpers.oracle_token_spans(bad, idx)
    _check_spans_against_reference_scans(bad, idx, end, flags)
    got = [(bad[i:e].decode(), int(f)) for i, e, f in zip(idx.tolist(), end.tolist(), flags.tolist()) if f & 4]
    N, F, B = 4, 4 + 8, 4 + 32
    assert got == [("12", B), ("-", N), ("-", B), ("1", B), ("1.5x", F), ("1e5", F), ("-0.5E-3", F), ("0", B), ("1.", F),
                   ("12", N), ("3", N), ("4", N), ("5", N), ("6", N), ("7", B), ('1.5"b"', F), ("9", N)], got


@pytest.mark.gpu
def test_stage2_documents_through_f1_to_f4(dev):
    """The four documents the reference's stage 2 accepts (tests/test_stage_2.mojo:47-67) through the HIP kernels of
    rows f1-f4: depth ends at 0 and never goes below it, every bracket has its partner, every string / number span
    equals the reference's own scans and lies inside its token's extent."""
    import os

    import torch

    for name in STAGE2_DOCS:
        js, _ = helpers.read_fixture(os.path.join(helpers.GOLDEN, "valid", name))
        idx, t, d, res, m = _gpu_tokens(dev, js)
        assert (res.final_depth, res.min_depth) == (0, 0) and res.n == len(idx), name
        for i, c in enumerate(t.tolist()):
            if c in b"[{":
                j = int(m[i])
                assert j != 0xFFFFFFFF and t[j] == {0x5B: 0x5D, 0x7B: 0x7D}[c] and int(m[j]) == i and d[i] == d[j], (name, i)
        d_buf = torch.from_numpy(np.frombuffer(js, dtype=np.uint8).copy()).to(dev.device)
        d_idx = torch.from_numpy(idx.view(np.int32).copy()).to(dev.device)
        d_idx = torch.cat([d_idx, torch.zeros(8, dtype=torch.int32, device=dev.device)])
        end, flags = dev.token_spans(d_buf, len(js), d_idx, len(idx))
        _check_spans_against_reference_scans(js, idx, end.cpu().numpy().view(np.uint32), flags.cpu().numpy())


@pytest.fixture(scope="module")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mojo_simdjson_amd.device import Stage1Device

    d = Stage1Device(0)
    yield d
    d.close()


def _gpu_tokens(dev, data):
    import torch

    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(data) + 3 + 4, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    r = dev.fetch(d_res)
    n = int(r.count)
    t, d, res, m = dev.tokens(d_buf, len(data), d_idx, n, match=True)
    return d_idx[:n].cpu().numpy().view(np.uint32), t.cpu().numpy(), d.cpu().numpy(), res, m.cpu().numpy().view(np.uint32)


def _check(dev, data, where):
    idx, t, d, res, m = _gpu_tokens(dev, data)
    wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx)
    wm = helpers.oracle_match(wt)
    if not np.array_equal(m, wm):
        bad = int(np.argmax(m != wm))
        raise AssertionError(f"{where}: match[{bad}] = {m[bad]} != {wm[bad]}")
    assert np.array_equal(t, wt), where
    if not np.array_equal(d, wd):
        bad = int(np.argmax(d != wd))
        raise AssertionError(f"{where}: depth[{bad}] = {d[bad]} != {wd[bad]}")
    assert (res.n, res.final_depth, res.min_depth, res.max_depth) == (len(idx), final, mn, mx), where


@pytest.fixture(params=["tiles", "tokens"])
def span_mode(dev, request):
    """The token calls' two kernels, each forced whatever the density of the index (the product picks by density):
    organised by tiles of the buffer (msj_debug_set_span_mode(2)) and by tokens (1): the same tests, the same
    definitions."""
    dev.lib.msj_debug_set_span_mode(dev.ctx, 1 if request.param == "tokens" else 2)
    request.addfinalizer(lambda: dev.lib.msj_debug_set_span_mode(dev.ctx, 0))
    return request.param


@pytest.mark.gpu
def test_tokens_fixtures_and_shapes(dev, span_mode):
    for f in helpers.golden_valid_files():
        js, _ = helpers.read_fixture(f)
        _check(dev, js, f)
    rng = np.random.default_rng(5)
    alphabet = np.frombuffer(b'{}[]{}[],: "a1', dtype=np.uint8)
    for n in (1, 2, 7, 8, 9, 2047, 2048, 2049, 4096 * 3 + 5, 100000, 1 << 20):
        soup = alphabet[rng.integers(0, len(alphabet), n)].tobytes().replace(b'"', b"x")  # no strings: every bracket counts
        _check(dev, soup, f"bracket soup {n}")
    _check(dev, b"[" * 300000 + b"]" * 299999, "deep")
    _check(dev, b"]" * 5000 + b"[" * 7, "underflow")
    _check(dev, b" ", "no structurals")


def _check_spans(dev, data, where):
    import torch

    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(data) + 3 + 4, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n = int(dev.fetch(d_res).count)
    e, f = dev.token_spans(d_buf, len(data), d_idx, n)
    idx = d_idx[:n].cpu().numpy().view(np.uint32)
    we, wf = helpers.oracle_token_spans(data, idx)
    assert np.array_equal(f.cpu().numpy(), wf), where
    assert np.array_equal(e.cpu().numpy().view(np.uint32), we), where


@pytest.mark.gpu
@pytest.mark.parametrize("lds_limit", ["", "0", "4096"])
def test_token_spans(dev, lds_limit, request, span_mode):
    """Both paths of the kernel: the workgroup's stretch staged in LDS with per-byte class bitmaps (default),
    and the per-token path from global memory that long stretches take (forced by a limit of 0; 4096 mixes them)."""
    from mojo_simdjson_amd import synth

    dev.lib.msj_debug_set_span_limits(dev.ctx, int(lds_limit) if lds_limit else 0xFFFFFFFF, 0xFFFFFFFF)
    request.addfinalizer(lambda: dev.lib.msj_debug_set_span_limits(dev.ctx, 0xFFFFFFFF, 0xFFFFFFFF))

    for f in helpers.golden_valid_files():
        js, _ = helpers.read_fixture(f)
        _check_spans(dev, js, f)
    for name in ("minified", "utf8", "pretty4"):
        _check_spans(dev, synth.workload(name, 8 << 20).tobytes(), name)
    _check_spans(dev, b'["' + b"a" * 5000 + b'", ' + b"9" * 3000 + b', "tail\\', "long spans and an escape at the very end")
    _check_spans(dev, b'"' + b"\\" * 600 + b'"', "cap reached inside an escape")
    _check_spans(dev, b"-", "one byte")
    # numbers around the cap of 1 024 characters, at the end of the buffer, on and off the 16-byte grid
    for digits in (1, 15, 16, 31, 32, 33, 63, 64, 65, 1023, 1024, 1025, 1026, 2000):
        for tail in (b"", b" ", b",1]", b"e5 ", b"." + b"0" * 20):
            for pad in (0, 5, 13):
                _check_spans(dev, b" " * pad + b"[" + b"7" * digits + tail, f"number of {digits} digits + {tail!r} at {pad}")
    _check_spans(dev, b"[" + b"1.5e+3," * 4000 + b"-0.25E-7]", "floats over several workgroups")
    # a long string in the middle of many short tokens: its workgroup reads from global memory, the others stage
    _check_spans(dev, b"[" + b'"a\\b",12,' * 3000 + b'"' + b"x" * 40000 + b'",' + b'"cd",3.5,' * 3000 + b"0]", "mixed paths")
    _check_spans(dev, b'["a"  ,"b\\"" , "c\\\\"  ]  ', "blanks between the closing quote and the next structural")
    # round 5: the escape flag of a string is exact at ANY length (bodies over 1 024 bytes: a wave per string behind the
    # span kernel, over 1 MiB the whole grid).  The only backslash at the far end, at the near end, nowhere; bodies
    # around the old cap, over several tiles / tile groups, and an escaped quote as the body's last bytes
    for n in (1022, 1023, 1024, 1025, 1026, 4095, 4096, 4097, 12288, 16384, 70000):
        for body in (b"a" * n, b"a" * (n - 2) + b"\\n", b"\\t" + b"a" * (n - 2), b"a" * (n - 2) + b'\\"', b"a" * (n // 2) + b"\\\\" + b"a" * (n - n // 2 - 2)):
            assert len(body) == n
            for pad in (0, 7, 4090):
                _check_spans(dev, b" " * pad + b'["k",' + b"1," * 100 + b'"' + body + b'" , 5,"x"]', f"string of {n} bytes at {pad}: {body[:3]!r}..{body[-3:]!r}")
    _check_spans(dev, b"[" + (b'"' + b"a" * 1500 + b'","' + b"b" * 1400 + b"\\n" + b"b" * 98 + b'",') * 300 + b"0]", "600 long strings")
    big = b"x" * ((1 << 20) + 4097)
    _check_spans(dev, b'{"blob":"' + big + b'","next":"' + big[:-9] + b"\\u00e9abc" + b'","n":1}', "bodies over 1 MiB: the whole grid per string")
    _check_spans(dev, b'["' + big * 3 + b"\\\\" + b'"]', "3 MiB, the backslash in the last piece")
    _check_spans(dev, b'["' + b"\\/" + big * 2 + b'"]', "2 MiB, the backslash in the first piece")
    # the long-string list overflows (capacity lowered to 3 entries): the marker is found in flags[] itself
    dev.lib.msj_debug_set_span_limits(dev.ctx, int(lds_limit) if lds_limit else 0xFFFFFFFF, 3)
    _check_spans(dev, b"[" + (b'"' + b"a" * 1500 + b'","' + b"b" * 1400 + b"\\n" + b"b" * 98 + b'",') * 30 + b"0]", "the long-string list overflows")
    dev.lib.msj_debug_set_span_limits(dev.ctx, int(lds_limit) if lds_limit else 0xFFFFFFFF, 0xFFFFFFFF)
    _check_spans(dev, b"[" + (b'"' + b"a" * 1500 + b'","' + b"b" * 1400 + b"\\n" + b"b" * 98 + b'",') * 30 + b"0]", "the lists are clean again")
    # what sends a lane from the one-round evaluation to the general loop: more than 32 blanks / backslashes / digits
    # in a row, a body that crosses the 4 KiB one wave counts backslashes over (with and without an escape in it)
    _check_spans(dev, b'["x"' + b" " * 33 + b',"y"' + b" " * 32 + b',"z"' + b"\n" * 100 + b', 1' + b" " * 70 + b"]" + b" " * 50,
                 "long runs of blanks in front of the next structural")
    for body in (b"a" * 90 + b"\\n" + b"b" * 7, b"a" * 99, b"\\\\" * 17 + b"c" * 65, b"\\" * 33 + b"\\" + b"d"):
        _check_spans(dev, b"[" + (b'"' + body + b'",') * 700 + b"0]", f"bodies across 4 KiB edges: {body[:12]!r}...")
    _check_spans(dev, b"[" + b"1" * 40 + b"." + b"5" * 40 + b"e" + b"7" * 40 + b"," + b"-" + b"9" * 33 + b"x]", "long floats")
    # a float's scan (`while not structural-or-blank`) goes on past the next structural where that is a scalar that
    # follows a quote, and -- for the last token of a workgroup -- past the bytes the workgroup has staged
    for fill in (253, 254, 255, 256, 257):
        for tail in (300, 2000):
            _check_spans(dev, b"[" + b"1," * fill + b'1.5"xx"' + b"y" * tail + b" ,7]", f"float scan leaves the stretch ({fill}, {tail})")
    # ... many of them: the work list of the fix-up pass, and (capacity lowered to 3 entries) its overflow path
    many = b"[" + (b"1," * 255 + b'1.5"xx"' + b"y" * 300 + b" ,") * 40 + b"7]"
    _check_spans(dev, many, "forty tokens on the fix-up list")
    dev.lib.msj_debug_set_span_limits(dev.ctx, int(lds_limit) if lds_limit else 0xFFFFFFFF, 3)
    _check_spans(dev, many, "the fix-up list overflows")
    dev.lib.msj_debug_set_span_limits(dev.ctx, int(lds_limit) if lds_limit else 0xFFFFFFFF, 0xFFFFFFFF)
    _check_spans(dev, many, "the list is clean again")
    # the closing quote is found from the NEXT structural: byte soups (valid or not) must agree with the forward scan
    rng = np.random.default_rng(9)
    alphabets = [b'{}[],: \n"\\ab1', b'""\\\\ a,', b'"abc\\" \t:1e5-', b'"\\" \r\n"x']
    for k in range(300):
        a = np.frombuffer(alphabets[k % len(alphabets)], dtype=np.uint8)
        n = int(rng.integers(1, 3000))
        _check_spans(dev, a[rng.integers(0, len(a), n)].tobytes(), f"soup {k}")


@pytest.mark.gpu
def test_token_spans_arrays_off_the_wide_grid(dev, span_mode):
    """The kernel writes a pair of tokens per access when the arrays allow it (8-byte aligned ends, 2-byte aligned
    flags): off that grid, and for odd token counts, it must write the same values one by one."""
    import torch
    from mojo_simdjson_amd import synth
    from mojo_simdjson_amd.device import _ptr

    data = synth.workload("minified", 1 << 20).tobytes() + b' [1, "a\\"b", -2.5e3 ]'
    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(data) + 8, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n_all = int(dev.fetch(d_res).count)
    for n in (n_all, n_all - 1, 513, 512, 2, 1):
        # the oracle on the whole index array: a token's span does not depend on how many tokens follow it
        full_e, full_f = helpers.oracle_token_spans(data, d_idx[:n_all].cpu().numpy().view(np.uint32))
        for end_off, flag_off in ((1, 1), (0, 1), (1, 0), (0, 0)):
            d_end = torch.full((n + 4,), -1, dtype=torch.int32, device=dev.device)
            d_flags = torch.full((n + 4,), 0xEE, dtype=torch.uint8, device=dev.device)
            rc = dev.lib.msj_token_spans_device(dev.ctx, _ptr(d_buf), len(data), _ptr(d_idx), n, _ptr(d_end[end_off:]),
                                                _ptr(d_flags[flag_off:]), dev._stream())
            assert rc == 0
            e = d_end.cpu().numpy().view(np.uint32)
            f = d_flags.cpu().numpy()
            # tokens in front of the last one do not depend on n; the last one ends where the buffer ends when n < n_all
            assert np.array_equal(e[end_off:end_off + n - 1], full_e[:n - 1]), (n, end_off, flag_off)
            assert np.array_equal(f[flag_off:flag_off + n - 1], full_f[:n - 1]), (n, end_off, flag_off)
            assert (e[end_off + n:] == 0xFFFFFFFF).all() and (f[flag_off + n:] == 0xEE).all(), "nothing behind the arrays"
            assert (e[:end_off] == 0xFFFFFFFF).all() and (f[:flag_off] == 0xEE).all(), "nothing in front of the arrays"
            if n == n_all:
                assert e[end_off + n - 1] == full_e[n - 1] and f[flag_off + n - 1] == full_f[n - 1]


def _check_prep(dev, data, where):
    import torch

    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(data) + 3 + 4, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n = int(dev.fetch(d_res).count)
    idx = d_idx[:n].cpu().numpy().view(np.uint32)
    t, d, res, m, e, f = dev.stage2_prep(d_buf, len(data), d_idx, n, match=True)
    wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx)
    we, wf = helpers.oracle_token_spans(data, idx)
    assert np.array_equal(t.cpu().numpy(), wt), where
    assert np.array_equal(d.cpu().numpy(), wd), where
    assert (res.n, res.final_depth, res.min_depth, res.max_depth) == (n, final, mn, mx), where
    assert np.array_equal(m.cpu().numpy().view(np.uint32), helpers.oracle_match(wt)), where
    assert np.array_equal(f.cpu().numpy(), wf), where
    assert np.array_equal(e.cpu().numpy().view(np.uint32), we), where


@pytest.mark.gpu
@pytest.mark.parametrize("lds_limit", ["", "0", "4096"])
def test_stage2_prep_equals_the_separate_calls(dev, lds_limit, request, span_mode):
    """msj_stage2_prep_device = token pre-pass + token spans from one pass over the buffer."""
    from mojo_simdjson_amd import synth

    dev.lib.msj_debug_set_span_limits(dev.ctx, int(lds_limit) if lds_limit else 0xFFFFFFFF, 0xFFFFFFFF)
    request.addfinalizer(lambda: dev.lib.msj_debug_set_span_limits(dev.ctx, 0xFFFFFFFF, 0xFFFFFFFF))
    for f in helpers.golden_valid_files():
        js, _ = helpers.read_fixture(f)
        _check_prep(dev, js, f)
    for name in ("minified", "utf8", "pretty4"):
        _check_prep(dev, synth.workload(name, 8 << 20).tobytes(), name)
    rng = np.random.default_rng(12)
    alphabet = np.frombuffer(b'{}[]{}[],: "a1\\\n', dtype=np.uint8)
    for n in (1, 2, 255, 256, 257, 511, 512, 513, 2047, 2048, 2049, 4096 * 3 + 5, 100000, 1 << 20):
        _check_prep(dev, alphabet[rng.integers(0, len(alphabet), n)].tobytes(), f"soup {n}")
    _check_prep(dev, b"[" * 300000 + b"]" * 299999, "deep")
    _check_prep(dev, b"]" * 5000 + b"[" * 7, "underflow")
    _check_prep(dev, b'[' + b'"a\\\\b",12,' * 3000 + b'"' + b"x" * 40000 + b'",' + b'"cd",3.5,' * 3000 + b"0]", "mixed paths")
    _check_prep(dev, b" ", "no structurals")


@pytest.mark.gpu
@pytest.mark.parametrize("lds_limit", ["", "4096"])
def test_prep_around_the_tile_groups(dev, lds_limit, request):
    """The kernel organised by tiles stages a group of tiles of the buffer + a 2 KiB halo per workgroup and hands out tokens in
    chunks of 128 that belong to the group their first token lies in: tokens, chunk borders, long strings, floats whose
    scan runs past the next structural and numbers at the cap are moved across the group border, the end of the halo
    and the end of the buffer byte by byte."""
    dev.lib.msj_debug_set_span_limits(dev.ctx, int(lds_limit) if lds_limit else 0xFFFFFFFF, 0xFFFFFFFF)
    dev.lib.msj_debug_set_span_mode(dev.ctx, 2)
    request.addfinalizer(lambda: (dev.lib.msj_debug_set_span_limits(dev.ctx, 0xFFFFFFFF, 0xFFFFFFFF), dev.lib.msj_debug_set_span_mode(dev.ctx, 0)))
    G, H = int(dev.lib.msj_debug_tile_group(0)), int(dev.lib.msj_debug_tile_group(1))  # 12 KiB of the buffer per workgroup + 2 KiB
    rng = np.random.default_rng(77)
    # (a) a dense run of short tokens across two borders, shifted byte by byte (chunk grid against byte grid)
    for shift in list(range(0, 9)) + [63, 64, 65, 127, 128, 129]:
        _check_prep(dev, b" " * shift + b"[" + b'1,"a\\",{"k":-2.5e3},' * 2100 + b"0]", f"dense run shifted by {shift}")
    # (b) one token placed around the border / the end of the halo, its extent crossing either
    for edge in (G, G + H, 2 * G, 2 * G + H):
        for off in (-70, -64, -33, -2, -1, 0, 1, 31, 64):
            for tok in (b'"' + b"s" * 90 + b'\n"', b"-" + b"8" * 70 + b".5e+7", b"3" * 1030, b'"' + b"\\" * 40 + b'"', b"true"):
                head = b"[" + b"0," * 1500
                pad = edge + off - len(head)
                doc = head + b" " * pad + tok + b" ," + b"1," * 600 + b"2]"
                _check_prep(dev, doc, f"{tok[:6]!r} at {edge}{off:+d}")
    # (c) a float whose scan passes the next structural (a scalar behind a quote) and leaves the staged range: the
    #     chunk's last token sits at the end of the halo
    for fill in range(1020, 1030):
        doc = b"[" + b"1," * 100 + b" " * (G - 1024 - 201) + b'"",' * 340 + b" " * fill + b'1.5"xx"' + b"y" * 3000 + b" ,7]"
        _check_prep(dev, doc, f"float scan leaves the staged range ({fill})")
    # (d) sparse stretches: groups that own no chunk, chunks far longer than a group, then dense again
    doc = b"[" + b'"' + b"x" * 70000 + b'",' + b" " * 40000 + b"1," * 5000 + b'"' + b"y" * 20000 + b'"' + b" " * 20000 + b",[]]"
    _check_prep(dev, doc, "sparse and dense")
    # ... and indices far from linear in the byte offset (the group table closes in on its entries by interpolation
    # before it bisects): all tokens in the first / the last percent of the buffer, a dense island between deserts
    _check_prep(dev, b"[" + b"1," * 20000 + b" " * 4000000 + b"2]", "all tokens in front")
    _check_prep(dev, b"[" + b" " * 4000000 + b"1," * 20000 + b"2]", "all tokens behind")
    _check_prep(dev, b"[" + b" " * 1500000 + b'"a",' * 30000 + b" " * 2500000 + b"3," * 10 + b" " * 700000 + b"4]", "an island")
    # (e) byte soups of every size around the group size
    alphabet = np.frombuffer(b'{}[],: "a1\\\n-.e', dtype=np.uint8)
    for n in (G - 1, G, G + 1, G + H - 1, G + H, G + H + 1, 2 * G + 5, 5 * G + 77):
        _check_prep(dev, alphabet[rng.integers(0, len(alphabet), n)].tobytes(), f"soup {n}")
    # (g) runs of backslashes that cross a group border (or a tile border inside a group) and end right in front of a
    #     quote, outside a string and inside one.  The tile kernel resolves escapes per block with the carry into a GROUP
    #     taken as 0: a run that comes in from the bytes in front of the group belongs to a token of an earlier group --
    #     inside a string to that string, outside one to the scalar the run starts (stage 1 makes the first backslash
    #     the structural, never the quote behind the run) -- and the quotes the group's own tokens look at lie behind an
    #     opening quote inside the group
    for border in (G, G + 4096, 2 * G):
        for run in (1, 2, 3, 63, 64, 65, 127, 128, 129, 4095, 4096, 4097, 4200, 9001):
            for off in (0, 1, 2, 33):
                # 1 408 = 11 x 128 tokens in front: what follows the blanks starts a chunk, which then belongs to the group
                # behind the border
                head = b"0," * 704
                pad = border + off - run - len(head)
                if pad < 0:
                    continue
                for body in (b"ab\\n", b"ab"):  # with and without a backslash of its own
                    outside = head + b" " * pad + b"\\" * run + b'"' + body + b'" ,"c",' + b"1," * 300 + b"2]"
                    _check_prep(dev, outside, f"run of {run} in front of a quote at {border}+{off}, outside a string, body {body!r}")
                inside = head + b" " * (pad - 1) + b'"' + b"\\" * run + b'" ,"d\\"e",' + b"1," * 300 + b"2]"
                _check_prep(dev, inside, f"run of {run} in front of a quote at {border}+{off}, inside a string")
    # (f) the wide stores need aligned arrays: the same through the span call with arrays off the grid
    _check_spans(dev, b"[" + b'"ab",' * 9000 + b"1]", "spans only")


@pytest.mark.gpu
def test_tokens_workloads(dev):
    from mojo_simdjson_amd import synth

    for name in ("minified", "utf8", "pretty4"):
        _check(dev, synth.workload(name, 8 << 20).tobytes(), name)


@pytest.mark.gpu
def test_tokens_1gib_replicated(dev):
    """1 GiB: every unit is a complete document, so the depth pattern repeats unit by unit."""
    import torch

    from mojo_simdjson_amd import synth

    u = synth.workload("minified", 64 << 20)
    oracle = helpers.load_oracle()
    idx_u = _stage1(oracle, u.tobytes())
    wt, wd, (final, mn, mx) = helpers.oracle_tokens(u.tobytes(), idx_u)
    assert final == 0 and mn == 0
    reps = 16
    d_buf = torch.from_numpy(u).to(dev.device).repeat(reps)
    d_idx = torch.empty(len(idx_u) * reps + 16, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n = int(dev.fetch(d_res).count)
    assert n == len(idx_u) * reps
    t, d, res, m = dev.tokens(d_buf, d_buf.numel(), d_idx, n, match=True)
    assert (res.n, res.final_depth, res.min_depth, res.max_depth) == (n, 0, 0, mx)
    wm = helpers.oracle_match(wt).astype(np.int64)
    wm_d = torch.from_numpy(wm).to(dev.device)
    k = torch.arange(reps, device=dev.device, dtype=torch.int64)[:, None] * len(idx_u)
    want_m = torch.where(wm_d[None, :] == 0xFFFFFFFF, wm_d[None, :], wm_d[None, :] + k)  # partners stay inside their unit
    assert torch.equal(m.view(reps, -1).to(torch.int64) & 0xFFFFFFFF, want_m)
    wt_d = torch.from_numpy(wt).to(dev.device)
    wd_d = torch.from_numpy(wd).to(dev.device)
    assert torch.equal(t.view(reps, -1), wt_d.expand(reps, -1))
    assert torch.equal(d.view(reps, -1), wd_d.expand(reps, -1))


@pytest.mark.gpu
@pytest.mark.parametrize("workload,mode,reps", [("minified", 0, 16), ("minified", 1, 16), ("utf8", 0, 16), ("utf8", 2, 16), ("pretty8", 0, 16),
                                                ("pretty8", 2, 16), ("minified", 0, 63), ("pretty8", 0, 63)])
def test_stage2_prep_1gib_replicated(dev, workload, mode, reps, request):
    """msj_stage2_prep_device at full size (BASELINE configs 2-4): every unit is a complete document, so type, depth and
    span flags repeat unit by unit and span ends repeat shifted by the unit's length -- every token of the 1 GiB stream is
    compared on the device with the definition's result for ONE unit.  Mode 0 is the product's choice of kernel by the
    density of the index (tiles for minified and UTF-8-heavy, tokens for indent 8), 1 / 2 force the other one.  63 units
    = 3.94 GiB: offsets and byte positions at the top of what one uint32 segment holds."""
    import torch

    from mojo_simdjson_amd import synth

    dev.lib.msj_debug_set_span_mode(dev.ctx, mode)
    request.addfinalizer(lambda: dev.lib.msj_debug_set_span_mode(dev.ctx, 0))
    u = synth.workload(workload, 64 << 20)
    oracle = helpers.load_oracle()
    data = u.tobytes()
    idx_u = _stage1(oracle, data)
    wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx_u)
    we, wf = helpers.oracle_token_spans(data, idx_u)
    assert final == 0 and mn == 0
    nu = len(idx_u)
    d_buf = torch.from_numpy(u).to(dev.device).repeat(reps)
    d_idx = torch.empty(nu * reps + 16, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n = int(dev.fetch(d_res).count)
    assert n == nu * reps
    t, d, res, _, e, f = dev.stage2_prep(d_buf, d_buf.numel(), d_idx, n)
    assert (res.n, res.final_depth, res.min_depth, res.max_depth) == (n, 0, 0, mx)
    assert torch.equal(t.view(reps, -1), torch.from_numpy(wt).to(dev.device).expand(reps, -1))
    assert torch.equal(d.view(reps, -1), torch.from_numpy(wd).to(dev.device).expand(reps, -1))
    assert torch.equal(f.view(reps, -1), torch.from_numpy(wf).to(dev.device).expand(reps, -1))
    we_d = torch.from_numpy(we.astype(np.int64)).to(dev.device)
    # a string / number token's end is an offset into the stream; every other token's is 0.  (The stream's very last
    # token is the closing bracket of the last unit: no token's span reaches the end of the buffer.)
    ev = e.view(reps, -1)
    for k in range(reps):
        want_e = torch.where(we_d == 0, we_d, we_d + k * len(data))
        assert torch.equal(ev[k].to(torch.int64) & 0xFFFFFFFF, want_e), k


# ---- round 4: bracket partners inside a block (apply_depth), the depth carried from call to call, segments ----
@pytest.mark.gpu
def test_bracket_partners_around_the_block_levels(dev, span_mode):
    """The partner of a bracket is settled inside its block of 2 048 tokens where its container closes there and lies
    inside the 16 depth levels the block keeps bitmaps of (4 below the depth at its start .. 11 above); everything
    else goes through the min tree.  Containers placed across block borders, nests that leave the levels on either
    side, the same level re-used many times inside one block, empty containers, stray brackets."""
    cases = {
        "many small containers at one level": b"[" + b"[1,2],{},[[]]," * 3000 + b"0]",
        "a nest of 11 / 12 / 13 inside a block": b"[" + (b"[" * 11 + b"]" * 11 + b",") * 50 + (b"[" * 12 + b"]" * 12 + b",") * 50 +
                                                   (b"[" * 13 + b"]" * 13 + b",") * 50 + b"0]",
        "a block that starts deep and climbs 6 below its start": b"[" * 40 + b"1," * 2100 + b"2" + b"]" * 6 + b",[3]" * 700 + b"]" * 34,
        "containers over block borders": b"[" + (b"[" + b"7," * 500 + b"8],") * 30 + b"0]",
        "closing brackets in front of every opening one": b"]" * 3000 + b"[" * 10 + b"]" * 10 + b"[" * 3000,
        "alternating at the top level": b"[]" * 5000 + b"{}" * 5000,
        "one long flat array": b"[" + b"1," * 50000 + b"1]",
    }
    for where, data in cases.items():
        _check(dev, data, where)
    rng = np.random.default_rng(11)
    for k in range(40):  # nests of random depth and width around the block size
        parts = []
        for _ in range(int(rng.integers(50, 400))):
            dpt = int(rng.integers(1, 24))
            parts.append(b"[" * dpt + b"1," * int(rng.integers(0, 60)) + b"1" + b"]" * dpt)
        _check(dev, b"[" + b",".join(parts) + b"]", f"random nests {k}")


@pytest.mark.gpu
def test_depth_is_carried_from_call_to_call(dev, span_mode):
    """msj_tokens_chain_device / msj_stage2_prep_chain_device: a token array cut anywhere -- inside containers, at
    negative depths -- and handed over in pieces gives the depths, the final / minimum / maximum of the whole; the
    partners are those of each piece alone (a container cut by the border keeps 0xFFFFFFFF at both ends)."""
    import torch
    from mojo_simdjson_amd import synth

    rng = np.random.default_rng(12)
    docs = [synth.workload("minified", 2 << 20).tobytes(), b"]" * 700 + b"[" * 300 + b"1," * 5000 + b"]" * 100 + b"[" * 9000,
            b"[" + (b"[" * 9 + b"1" + b"]" * 9 + b",") * 3000 + b"0]"]
    for which, data in enumerate(docs):
        d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
        d_idx = torch.empty(len(data) + 8, dtype=torch.int32, device=dev.device)
        d_res = dev.new_carry()
        dev.index(d_buf, d_idx, d_res)
        n = int(dev.fetch(d_res).count)
        idx = d_idx[:n].cpu().numpy().view(np.uint32)
        wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx)
        for trial in range(4):
            cuts = sorted(set([0, n] + [int(c) // 4 * 4 for c in rng.integers(1, n, int(rng.integers(1, 4)))]))  # 16-byte aligned slices
            prev = None
            for a, b in zip(cuts[:-1], cuts[1:]):
                res_buf = torch.zeros(24, dtype=torch.uint8, device=dev.device)
                if trial % 2 == 0:
                    t, d, _, m = dev.tokens(d_buf, len(data), d_idx[a:], b - a, match=True, d_result=res_buf, sync=False, d_prev=prev)
                else:
                    t, d, _, m, _, _ = dev.stage2_prep(d_buf, len(data), d_idx[a:], b - a, match=True, d_prev=prev, d_result=res_buf)
                where = f"doc {which}, tokens [{a}, {b}) of {n}"
                assert np.array_equal(t.cpu().numpy(), wt[a:b]), where
                assert np.array_equal(d.cpu().numpy(), wd[a:b]), where
                assert np.array_equal(m.cpu().numpy().view(np.uint32), helpers.oracle_match(wt[a:b])), where
                prev = res_buf
            from mojo_simdjson_amd import _lib

            r = _lib.MsjTokensResult.from_buffer_copy(prev.cpu().numpy().tobytes())
            assert (r.n, r.final_depth, r.min_depth, r.max_depth) == (cuts[-1] - cuts[-2], final, mn, mx), (which, cuts, r.final_depth, r.min_depth, r.max_depth)


@pytest.mark.gpu
def test_prep_segments_of_one_shard(dev):
    """msj_stage2_prep_segments: the shard call's segment table (here 1 MiB segments forced by the test hook, so that
    an 8 MiB document is a chain of eight) -> type, depth, spans and partners of every token, the depth handed from
    segment to segment on the device; a later segment's index slice starts wherever the one in front ended (not on the
    16-byte grid: the library copies it)."""
    import torch
    from mojo_simdjson_amd import _lib, synth

    data = synth.workload("minified", 8 << 20).tobytes()
    L = dev.lib
    L.msj_debug_set_segment_bytes.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    assert L.msj_debug_set_segment_bytes(dev.ctx, 1 << 20) == 0
    try:
        d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
        d_idx = torch.empty(len(data) // 2, dtype=torch.int32, device=dev.device)
        d_seg = torch.zeros(16 * 32, dtype=torch.uint8, device=dev.device)
        cin, cout = dev.make_carry(0, 0, 0), dev.new_carry()
        _, nseg = dev.shard(d_buf, len(data), d_idx, cin, cout, segments=d_seg, is_final=True, trailer_len=len(data))
        assert nseg == 8 and dev.fetch(cout).code == 0
        table = np.frombuffer(d_seg.cpu().numpy().tobytes(), dtype=np.uint64).reshape(16, 4)[:nseg]
        segs = [tuple(int(x) for x in row) for row in table]
        assert any(s[2] % 4 for s in segs[1:]), "the test wants a slice off the 16-byte grid"
        offs, t, d, m, e, f, results = dev.stage2_prep_segments(d_buf, segs, d_idx, match=True)
        idx_all = d_idx.cpu().numpy().view(np.uint32)
        # partners over the WHOLE shard (round 5): the definition on the document's token stream, every index mapped to
        # its position in the shard's output arrays (the document's root object spans all eight segments)
        n_all = sum(sg[3] for sg in segs)
        pos_of = np.concatenate([offs[s] + np.arange(sg[3], dtype=np.int64) for s, sg in enumerate(segs)])
        wt_all, _, _ = helpers.oracle_tokens(data, np.concatenate([idx_all[sg[2]:sg[2] + sg[3]] + np.uint32(sg[0]) for sg in segs]))
        wm_all = helpers.oracle_match(wt_all).astype(np.int64)
        want_pos = np.where(wm_all == 0xFFFFFFFF, 0xFFFFFFFF, pos_of[np.minimum(wm_all, n_all - 1)])
        depth0, mn_all, mx_all, stitched = 0, None, None, 0
        for s, (bb, bl, ib, cnt) in enumerate(segs):
            piece = data[bb:bb + bl]
            idx = idx_all[ib:ib + cnt]
            wt, wd, (final, mn, mx) = helpers.oracle_tokens(piece, idx)
            we, wf = helpers.oracle_token_spans(piece, idx)
            o = offs[s]
            where = f"segment {s}"
            assert np.array_equal(t[o:o + cnt].cpu().numpy(), wt), where
            assert np.array_equal(d[o:o + cnt].cpu().numpy(), wd + depth0), where
            got_m = m[o:o + cnt].cpu().numpy().view(np.uint32).astype(np.int64)
            assert np.array_equal(got_m, want_pos[ib - segs[0][2]:ib - segs[0][2] + cnt]), where
            stitched += int(((helpers.oracle_match(wt).astype(np.int64) == 0xFFFFFFFF) & (got_m != 0xFFFFFFFF)).sum())
            assert np.array_equal(f[o:o + cnt].cpu().numpy(), wf) and np.array_equal(e[o:o + cnt].cpu().numpy().view(np.uint32), we), where
            mn_all = depth0 + mn if mn_all is None else min(mn_all, depth0 + mn)
            mx_all = depth0 + mx if mx_all is None else max(mx_all, depth0 + mx)
            depth0 += final
            r = results[s]
            assert (r.n, r.final_depth, r.min_depth, r.max_depth) == (cnt, depth0, mn_all, mx_all), where
        assert depth0 == 0  # the document is closed at the end of the shard
        assert int((want_pos != 0xFFFFFFFF).sum()) == int((wm_all != 0xFFFFFFFF).sum()) and results[-1].reserved >> 31 == 0
        assert stitched >= 2 * 7 and stitched % 2 == 0, stitched  # brackets whose partner lies in another segment
        # ... and a shard that starts and ends in the middle of things: segments 2 .. 6 alone (closing brackets whose
        # partners lie in front of the shard and opening ones never closed keep 0xFFFFFFFF; everything between pairs up)
        sub = segs[2:7]
        d_prev = torch.zeros(24, dtype=torch.uint8, device=dev.device)
        offs2, t2, d2, m2, e2, f2, res2 = dev.stage2_prep_segments(d_buf[sub[0][0]:], sub, d_idx[sub[0][2]:], match=True, d_prev=None)
        lo_tok, hi_tok = sub[0][2] - segs[0][2], sub[-1][2] + sub[-1][3] - segs[0][2]
        wm_sub = helpers.oracle_match(wt_all[lo_tok:hi_tok]).astype(np.int64)
        pos2 = np.concatenate([offs2[s] + np.arange(sg[3], dtype=np.int64) for s, sg in enumerate(sub)])
        want2 = np.where(wm_sub == 0xFFFFFFFF, 0xFFFFFFFF, pos2[np.minimum(wm_sub, hi_tok - lo_tok - 1)])
        got2 = np.concatenate([m2[offs2[s]:offs2[s] + sg[3]].cpu().numpy().view(np.uint32).astype(np.int64) for s, sg in enumerate(sub)])
        assert np.array_equal(got2, want2)
        assert int((want2 == 0xFFFFFFFF).sum()) > int((wm_all[lo_tok:hi_tok] == 0xFFFFFFFF).sum())  # some really are cut off
    finally:
        assert L.msj_debug_set_segment_bytes(dev.ctx, 0xFFFF0000) == 0


@pytest.mark.gpu
def test_prep_segments_over_4gib(dev):
    """config 5's own data through the rows marked "next".  A 4.5 GiB minified stream = one
    shard of TWO uint32 segments (cut inside a unit, inside a container) -> msj_stage2_prep_segments: type, depth (carried
    over the cut on the device), spans, partners of 0.9 G tokens, compared ON THE DEVICE with the definition by the
    replication property: a unit is a complete document, so token j of repetition k has the unit's type / depth / flags,
    the unit's end + k * unit_len (relative to its segment's first byte) and the unit's partner + k * unit_tokens
    (as a position in the shard's output arrays: round 5 -- a container opened in segment 0 and closed in segment 1 has
    its partner like any other)."""
    import torch
    from mojo_simdjson_amd import synth

    oracle = helpers.load_oracle()
    SEG = 0xFFFF0000
    u = synth.workload("minified", 64 << 20)
    b = u.tobytes()
    L = len(b)
    idx_u = _stage1(oracle, b)
    nu = int(idx_u.size)
    wt, wd, (final, _, _) = helpers.oracle_tokens(b, idx_u)
    assert final == 0
    wm = helpers.oracle_match(wt)
    we, wf = helpers.oracle_token_spans(b, idx_u)
    reps = SEG // L + 4
    dv = dev.device
    d_buf = torch.from_numpy(u).to(dv).repeat(reps)
    total = int(d_buf.numel())
    d_idx = torch.empty(nu * reps + 8, dtype=torch.int32, device=dv)
    d_seg = torch.zeros(4 * 32, dtype=torch.uint8, device=dv)
    cin, cout = dev.new_carry(), dev.new_carry()
    rc, nseg = dev.shard(d_buf, total, d_idx, cin, cout, segments=d_seg, is_final=True, trailer_len=total)
    assert rc == 0 and nseg == 2 and dev.fetch(cout).code == 0
    table = np.frombuffer(d_seg.cpu().numpy().tobytes(), dtype=np.uint64).reshape(4, 4)[:2]
    segs = [tuple(int(x) for x in row) for row in table]
    c0, c1 = segs[0][3], segs[1][3]
    assert c0 + c1 == nu * reps and segs[1][0] == SEG
    offs, t, d, m, e, f, results = dev.stage2_prep_segments(d_buf, segs, d_idx, match=True)
    assert (results[1].final_depth, results[1].min_depth) == (0, 0) and results[0].final_depth > 0  # the cut is inside a container
    g_t = torch.from_numpy(wt).to(dv)
    g_d = torch.from_numpy(wd).to(dv)
    g_f = torch.from_numpy(wf).to(dv)
    g_e = torch.from_numpy(we.astype(np.int64)).to(dv)
    g_m = torch.from_numpy(wm.astype(np.int64)).to(dv)       # 0xFFFFFFFF = none
    g_start = torch.from_numpy(idx_u.astype(np.int64)).to(dv)
    g_next = torch.cat([g_start[1:], torch.tensor([L], dtype=torch.int64, device=dv)])  # where the token's extent ends
    NONE = 0xFFFFFFFF
    crossing = 0
    for k in range(reps):
        J0 = k * nu  # global token index of the repetition's first token
        for s, (bb, bl, ib, cnt) in enumerate(segs):
            lo, hi = max(J0, ib), min(J0 + nu, ib + cnt)  # the repetition's tokens inside this segment
            if lo >= hi:
                continue
            r0, r1 = lo - J0, hi - J0
            o = offs[s] + (lo - ib)
            where = f"repetition {k}, segment {s}"
            assert torch.equal(t[o:o + hi - lo], g_t[r0:r1]), where
            assert torch.equal(d[o:o + hi - lo], g_d[r0:r1]), where
            # spans: tokens whose extent [start, next structural) lies inside the segment's bytes
            start_g = g_start[r0:r1] + k * L
            inside = (g_next[r0:r1] + k * L <= bb + bl) & (start_g >= bb)
            got_f, got_e = f[o:o + hi - lo], e[o:o + hi - lo].to(torch.int64) & 0xFFFFFFFF
            want_e = torch.where(g_e[r0:r1] != 0, g_e[r0:r1] + k * L - bb, torch.zeros_like(start_g))
            assert torch.equal(got_f[inside], g_f[r0:r1][inside]) and torch.equal(got_e[inside], want_e[inside]), where
            assert int((~inside).sum()) <= 1, where
            # partners (round 5: valid over the whole shard): the position in the shard's output arrays of the unit's
            # partner -- also for the containers the segment border cuts (the unit's root array among them)
            pg = g_m[r0:r1] + J0
            has = g_m[r0:r1] != NONE
            pos = torch.full_like(pg, NONE)
            for s2, (_, _, ib2, cnt2) in enumerate(segs):
                in2 = has & (pg >= ib2) & (pg < ib2 + cnt2)
                pos = torch.where(in2, pg - ib2 + offs[s2], pos)
            got_m = m[o:o + hi - lo].to(torch.int64) & 0xFFFFFFFF
            assert torch.equal(got_m, pos), where
            crossing += int((has & ((pg < ib) | (pg >= ib + cnt))).sum())
    assert crossing >= 4 and crossing % 2 == 0, crossing  # the cut lies inside a unit: its root object and array at least
    assert results[1].reserved >> 31 == 0


@pytest.mark.gpu
def test_types_from_stage1_prototype(dev):
    """PROTOTYPE: msj_stage1_types_device writes d_types[k] = buf[idx[k]] beside every index
    from the same emission, msj_depth_from_types_device turns them into depths (and partners): both equal what the
    separate calls and the definitions give -- fixtures, workloads (every emission path: staged, two rounds, block-wise,
    clipped), soups, a 1 GiB stream."""
    import torch
    from mojo_simdjson_amd import synth

    oracle = helpers.load_oracle()
    docs = []
    for f in helpers.golden_valid_files():
        docs.append((f, helpers.read_fixture(f)[0]))
    for name in ("minified", "utf8", "pretty4"):
        docs.append((name, synth.workload(name, 4 << 20).tobytes()))
    for kind in (0, 1, 4, 5, 6, 7):
        docs.append((f"extreme {kind}", synth.extreme((1 << 20) - 52, kind).tobytes()))
    rng = np.random.default_rng(17)
    alphabet = np.frombuffer(b'{}[]{}[],: "a1\\tn', dtype=np.uint8)
    for n in (1, 2, 7, 63, 64, 65, 4095, 4096, 4097, 70001):
        docs.append((f"soup {n}", alphabet[rng.integers(0, len(alphabet), n)].tobytes()))
    for where, data in docs:
        if isinstance(data, str):
            data = data.encode()
        d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
        d_idx = torch.zeros(len(data) + 3 + 4, dtype=torch.int32, device=dev.device)
        d_types = torch.zeros(len(data) + 3 + 4, dtype=torch.uint8, device=dev.device)
        d_res = dev.new_carry()
        dev.index_types(d_buf, d_idx, d_types, d_res)
        r = dev.fetch(d_res)
        n = int(r.count)
        idx = d_idx[:n].cpu().numpy().view(np.uint32)
        code, wn, widx = helpers.run_oracle(oracle.msj_oracle_stage1, data)
        if code in (0, 13):
            assert n == wn and np.array_equal(idx, widx[:n]), where
        arr = np.frombuffer(data, dtype=np.uint8)
        assert np.array_equal(d_types[:n].cpu().numpy(), arr[idx]), where
        if n:
            d_depth, d_match, d_tr = dev.depth_from_types(d_types, n, match=True)
            wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx)
            assert np.array_equal(d_depth[:n].cpu().numpy(), wd), where
            assert np.array_equal(d_match[:n].cpu().numpy().view(np.uint32), helpers.oracle_match(wt)), where
    # an index buffer that is too small: the clipped emission writes the types it writes indices for
    data = b"[" * 5000 + b"]" * 5000
    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.zeros(6000, dtype=torch.int32, device=dev.device)
    d_types = torch.zeros(6000, dtype=torch.uint8, device=dev.device)
    d_res = dev.new_carry()
    dev.index_types(d_buf, d_idx, d_types, d_res)
    assert dev.fetch(d_res).code == 1
    assert d_types[:5000].cpu().numpy().tobytes() == b"[" * 5000 and d_types[5000:6000].cpu().numpy().tobytes() == b"]" * 1000
    # full size: 1 GiB minified, replicated unit
    u = synth.workload("minified", 64 << 20)
    b = u.tobytes()
    idx_u = _stage1(oracle, b)
    wt, wd, _ = helpers.oracle_tokens(b, idx_u)
    nu = len(idx_u)
    for reps in (16, 63):  # 1 GiB; 3.94 GiB = the top of what one uint32 segment holds
        d_buf = torch.from_numpy(u).to(dev.device).repeat(reps)
        d_idx = torch.empty(nu * reps + 16, dtype=torch.int32, device=dev.device)
        d_types = torch.empty(nu * reps + 16, dtype=torch.uint8, device=dev.device)
        d_res = dev.new_carry()
        dev.index_types(d_buf, d_idx, d_types, d_res)
        assert int(dev.fetch(d_res).count) == nu * reps
        assert torch.equal(d_types[:nu * reps].view(reps, -1), torch.from_numpy(wt).to(dev.device).expand(reps, -1)), reps
        d_depth, _, _ = dev.depth_from_types(d_types, nu * reps)
        assert torch.equal(d_depth[:nu * reps].view(reps, -1), torch.from_numpy(wd).to(dev.device).expand(reps, -1)), reps
        del d_buf, d_idx, d_types, d_depth
        torch.cuda.empty_cache()


def _check_pairs(dev, data, where, spans):
    import torch
    from mojo_simdjson_amd import _lib

    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(data) + 3 + 4, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n = int(dev.fetch(d_res).count)
    t, d, pairs, e, f, d_tr = dev.stage2_prep_pairs(d_buf, len(data), d_idx, n, spans=spans)
    r = _lib.MsjTokensResult.from_buffer_copy(d_tr.cpu().numpy().tobytes())
    idx = d_idx[:n].cpu().numpy().view(np.uint32)
    wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx)
    wm = helpers.oracle_match(wt)
    opens = np.nonzero((wt == ord("{")) | (wt == ord("[")))[0]
    want = np.stack([opens.astype(np.uint32), wm[opens]], axis=1) if len(opens) else np.zeros((0, 2), dtype=np.uint32)
    assert r.reserved == len(opens), (where, r.reserved, len(opens))
    got = pairs[:len(opens)].cpu().numpy().view(np.uint32)
    if not np.array_equal(got, want):
        bad = int(np.argmax((got != want).any(axis=1)))
        raise AssertionError(f"{where}: pair {bad} = {got[bad].tolist()} != {want[bad].tolist()}")
    assert np.array_equal(t.cpu().numpy(), wt) and np.array_equal(d.cpu().numpy(), wd), where
    assert (r.n, r.final_depth, r.min_depth, r.max_depth) == (n, final, mn, mx) if n else True, where
    if spans and n:
        we, wf = helpers.oracle_token_spans(data, idx)
        assert np.array_equal(f[:n].cpu().numpy(), wf) and np.array_equal(e[:n].cpu().numpy().view(np.uint32), we), where


@pytest.mark.gpu
@pytest.mark.parametrize("spans", [True, False], ids=["prep", "tokens"])
def test_bracket_pairs_compact_list(dev, span_mode, spans):
    """Round 5: the partners as one {open, close} record per container, in the order of the opening brackets
    (msj_stage2_prep_pairs_device / msj_tokens_pairs_device) = the definition's match[] read at the opening brackets:
    fixtures, workloads, soups (stray and unclosed brackets), deep nests, containers across block borders and outside
    the sixteen levels a block keeps, a 1 GiB stream."""
    import torch
    from mojo_simdjson_amd import synth

    for f in helpers.golden_valid_files():
        js, _ = helpers.read_fixture(f)
        _check_pairs(dev, js.encode() if isinstance(js, str) else js, f, spans)
    for name in ("minified", "utf8", "pretty4"):
        _check_pairs(dev, synth.workload(name, 8 << 20).tobytes(), name, spans)
    rng = np.random.default_rng(23)
    alphabet = np.frombuffer(b'{}[]{}[],: "a1', dtype=np.uint8)
    # (2 048 tokens = a block of the depth pass = a wave; four blocks a workgroup; 2 048 brackets a block of the pairing)
    for n in (1, 2, 7, 8, 9, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 8191, 8192, 8193, 4096 * 3 + 5, 24577, 100000, 1 << 20):
        soup = alphabet[rng.integers(0, len(alphabet), n)].tobytes().replace(b'"', b"x")
        _check_pairs(dev, soup, f"bracket soup {n}", spans)
    for n in (2048, 8192, 8193, 70000):  # every token a bracket: the compact list as long as the token stream
        only = np.frombuffer(b"[]{}[[]]", dtype=np.uint8)[rng.integers(0, 8, n)].tobytes()
        _check_pairs(dev, only, f"brackets only {n}", spans)
    _check_pairs(dev, b"[" * 300000 + b"]" * 299999, "deep", spans)
    _check_pairs(dev, b"]" * 5000 + b"[" * 7, "underflow", spans)
    _check_pairs(dev, b"[" + b"[1]," * 3000 + b"[" * 20 + b"1" + b"]" * 20 + b",[[2]]" * 3000 + b"]", "mixed nests over many blocks", spans)
    _check_pairs(dev, b" ", "no structurals", spans)
    _check_pairs(dev, b'"a" 1 true "b" 2.5 null ' * 700, "tokens but no bracket at all", spans)
    _check_pairs(dev, b'[1,{"a":' * 3000, "opening brackets only", spans)
    _check_pairs(dev, b'1],"a"},' * 3000, "closing brackets only", spans)
    # full size
    u = synth.workload("minified", 64 << 20)
    oracle = helpers.load_oracle()
    b = u.tobytes()
    idx_u = _stage1(oracle, b)
    wt, wd, _ = helpers.oracle_tokens(b, idx_u)
    wm = helpers.oracle_match(wt).astype(np.int64)
    opens = np.nonzero((wt == ord("{")) | (wt == ord("[")))[0]
    nu, no = len(idx_u), len(opens)
    want = torch.from_numpy(np.stack([opens.astype(np.int64), wm[opens]], axis=1)).to(dev.device)
    # 1 GiB; and (the token call only) 63 units = 3.94 GiB, 818 M tokens, 98.6 M brackets: token indices, list slots and
    # record ranks at the top of what a uint32 segment holds
    for reps in (16,) if spans else (16, 63):
        d_buf = torch.from_numpy(u).to(dev.device).repeat(reps)
        d_idx = torch.empty(nu * reps + 16, dtype=torch.int32, device=dev.device)
        d_res = dev.new_carry()
        dev.index(d_buf, d_idx, d_res)
        assert int(dev.fetch(d_res).count) == nu * reps
        t, d, pairs, e, f, d_tr = dev.stage2_prep_pairs(d_buf, d_buf.numel(), d_idx, nu * reps, spans=spans)
        got = pairs[:no * reps].to(torch.int64) & 0xFFFFFFFF
        k = (torch.arange(reps, device=dev.device, dtype=torch.int64) * nu)[:, None, None]
        assert torch.equal(got.view(reps, no, 2), want[None, :, :] + k), reps
        assert torch.equal(d.view(reps, -1), torch.from_numpy(wd).to(dev.device).expand(reps, -1)), reps
        del d_buf, d_idx, t, d, pairs, got, k
        torch.cuda.empty_cache()
