"""Multi-document mode.

CPU part: the definition of the document split (oracle/tokens_oracle.c: msj_oracle_documents) checked by
hand, and against a restatement of upstream simdjson's backward scan (find_next_document_index) on
well-formed streams cut at arbitrary points.  GPU part: csrc/documents_kernel.hip against the definition,
and the windowed DocumentStream against the split of the whole stream.
PARITY UNPINNED: the reference has no streaming mode (tape_builder.mojo:25 "TODO: add streaming").
"""
import json
import random

import numpy as np
import pytest

from tests import helpers


def _rand_value(rng, depth=0):
    k = rng.random()
    if depth > 3 or k < 0.35:
        return rng.choice([0, 1, -12, 3.5, 1e10, True, False, None, "", "a", "x y", 'q"uo\\te', "café 中", "[not]{a}bracket"])
    if k < 0.7:
        return {f"k{j}": _rand_value(rng, depth + 1) for j in range(rng.randrange(0, 4))}
    return [_rand_value(rng, depth + 1) for _ in range(rng.randrange(0, 4))]


def _stream(rng, ndocs, scalars=True):
    """Concatenated documents with NDJSON newlines, blanks or nothing in between; -> (bytes, start offsets)."""
    out = bytearray()
    starts = []
    prev_scalar = False
    for _ in range(ndocs):
        v = _rand_value(rng) if scalars else _rand_value(rng, 0) if rng.random() < 0.0 else {"v": _rand_value(rng, 1)}
        text = json.dumps(v, ensure_ascii=rng.random() < 0.5, separators=(",", ":") if rng.random() < 0.7 else (", ", ": ")).encode()
        scalar = not text.startswith((b"{", b"["))
        sep = rng.choice([b"\n", b" ", b"\r\n", b"  \n", b""])
        if (scalar or prev_scalar) and sep == b"" and out:
            sep = b"\n"  # two scalars (or a scalar and a bracket) need a separator to be two tokens
        out += sep if out else b""
        starts.append(len(out))
        out += text
        prev_scalar = scalar
    return bytes(out), starts


def _split(oracle, data, is_final=False):
    idx, open_string = helpers.oracle_window(oracle.msj_oracle_stage1, data)
    typ, dep, _ = helpers.oracle_tokens(data, idx)
    first, res = helpers.oracle_documents(data, idx, typ, dep, open_string, is_final=is_final)
    return idx, typ, dep, open_string, first, res


def test_definition_by_hand(oracle):
    data = b'{"a":1} [1,2]\n3 "x" {"b":{"c":[]}} tru'
    idx, typ, dep, open_string, first, res = _split(oracle, data)
    assert not open_string
    assert [int(idx[i]) for i in first] == [0, 8, 14, 16, 20, 35]
    # a literal that touches the end of the window may go on in the next one ...
    assert res == (6, 5, idx.size - 1, 35)
    # ... unless the window is the end of the stream, or ends in a blank
    assert _split(oracle, data, is_final=True)[5] == (6, 6, idx.size, len(data))
    assert _split(oracle, data + b"e\n")[5] == (6, 6, idx.size, len(data) + 2)
    assert _split(oracle, data[:19])[5] == (4, 4, 12, 19)  # ... "x" : a closed string is complete
    data = b'{"a":1} {"b":[1,'
    idx, typ, dep, open_string, first, res = _split(oracle, data)
    assert [int(idx[i]) for i in first] == [0, 8]
    assert res == (2, 1, int(first[1]), 8)
    data = b'[1] "abc'
    idx, typ, dep, open_string, first, res = _split(oracle, data)
    assert open_string and res == (2, 1, 3, 4)
    data = b'[1] {"k":"abc'
    idx, typ, dep, open_string, first, res = _split(oracle, data)
    assert open_string and res == (2, 1, 3, 4)
    idx, typ, dep, open_string, first, res = _split(oracle, b"   ")
    assert res == (0, 0, 0, 3)
    # capacity clips the list, not the count
    idx, typ, dep, _, _, _ = _split(oracle, b"1 2 3 4")
    first, res = helpers.oracle_documents(b"1 2 3 4", idx, typ, dep, False, capacity=2)
    assert list(first) == [0, 1] and res[0] == 4


def test_definition_agrees_with_upstream_backward_scan(oracle):
    rng = random.Random(5)
    checked = 0
    for case in range(300):
        data, starts = _stream(rng, rng.randrange(1, 9), scalars=case % 2 == 0)
        cuts = sorted({len(data)} | {rng.randrange(1, len(data) + 1) for _ in range(12)})
        for cut in cuts:
            win = data[:cut]
            # is_final: upstream counts a number that touches the end of the window as complete
            idx, typ, dep, open_string, first, res = _split(oracle, win, is_final=True)
            if idx.size == 0:
                continue
            keep, err = helpers.oracle_find_next_document_index(win, idx, open_string)
            if err:  # upstream: nothing left after dropping the unclosed string's quote
                assert res[2] == 0
                continue
            # upstream keeps n - 1 tokens when the window ends in an unclosed string that is the last
            # token and everything before it is complete; the definition reports the same prefix
            assert keep == res[2], (win, keep, res)
            # and the documents the definition finds are the ones the generator wrote
            assert [int(idx[i]) for i in first] == [s for s in starts if s < cut and s <= int(idx[-1])]
            checked += 1
    assert checked > 2000


# ---------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    from mojo_simdjson_amd.device import Stage1Device

    d = Stage1Device(0)
    yield d
    d.close()


def _gpu_window(dev, data, skip=0, capacity=None, is_final=False, after_tokens=False):
    """Stage 1 (non-final shard, zero carries) + token pre-pass + document split of one window."""
    import torch

    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(data) + 3 + 4, dtype=torch.int32, device=dev.device)
    cin, cout = dev.new_carry(), dev.new_carry()
    dev.shard(d_buf, len(data), d_idx, cin, cout, is_final=False, flags=(skip & 15) << 24)
    c = dev.fetch(cout)
    n = int(c.count)
    t, d, tok = dev.tokens(d_buf, len(data), d_idx, n)
    first = None
    if capacity is not None:
        first = torch.full((max(capacity, 1),), -1, dtype=torch.int32, device=dev.device)[:capacity]
    d_first, res = dev.documents(d_buf, len(data), d_idx, n, t, d, is_final=is_final, d_carry=cout, d_doc_first=first, after_tokens=after_tokens)
    got = (int(res.n_documents), int(res.n_complete), int(res.tokens_complete), int(res.resume_offset))
    k = min(got[0], d_first.numel())
    return (d_idx[:n].cpu().numpy().view(np.uint32), bool(c.in_string), t.cpu().numpy(), d.cpu().numpy(),
            d_first[:k].cpu().numpy().view(np.uint32), got)


def _check_window(dev, oracle, data, where, capacity=None):
    widx, wopen = helpers.oracle_window(oracle.msj_oracle_stage1, data)
    wtyp, wdep, _ = helpers.oracle_tokens(data, widx)
    for is_final, after_tokens in ((False, False), (True, True), (False, True)):  # with / without the counts of the pre-pass
        idx, open_string, typ, dep, first, got = _gpu_window(dev, data, capacity=capacity, is_final=is_final, after_tokens=after_tokens)
        assert np.array_equal(idx, widx) and open_string == wopen, where
        wfirst, want = helpers.oracle_documents(data, widx, wtyp, wdep, wopen, capacity=capacity, is_final=is_final)
        assert got == want, (where, is_final, got, want)
        assert np.array_equal(first, wfirst), where


@pytest.mark.gpu
def test_document_split_matches_the_definition(dev, oracle):
    rng = random.Random(11)
    for case in range(60):
        data, _ = _stream(rng, rng.randrange(1, 40), scalars=case % 2 == 0)
        for cut in sorted({len(data)} | {rng.randrange(1, len(data) + 1) for _ in range(4)}):
            _check_window(dev, oracle, data[:cut], f"stream {case} cut at {cut}")
    # many blocks of the compaction: 300 000 small documents; one document over many blocks
    lines = b"".join(json.dumps({"id": i, "tags": ["a", "b"], "u": {"n": "x" * (i % 7)}}).encode() + b"\n" for i in range(300000))
    _check_window(dev, oracle, lines, "ndjson, complete")
    _check_window(dev, oracle, lines[: len(lines) - 9], "ndjson, cut in the last line")
    _check_window(dev, oracle, lines[:5000011], "ndjson, cut somewhere")
    _check_window(dev, oracle, lines[:400000], "capacity clips the list", capacity=1000)
    _check_window(dev, oracle, lines[:1000], "no list at all", capacity=0)
    _check_window(dev, oracle, b"[" + b"1," * 500000 + b"1]", "one document over many blocks")
    _check_window(dev, oracle, b"[" + b"1," * 500000 + b"1", "... not closed")
    _check_window(dev, oracle, b"1 " * 100000, "scalars only")
    _check_window(dev, oracle, b"   \n ", "blank window")
    _check_window(dev, oracle, b'"abc', "nothing but an open string")
    # bracket soups: the kernels must follow the definition on nonsense as well
    nrng = np.random.default_rng(3)
    alphabet = np.frombuffer(b'{}[]{}[],: a1\n', dtype=np.uint8)
    for n in (1, 7, 8, 9, 2047, 2048, 2049, 50000):
        for _ in range(3):
            _check_window(dev, oracle, alphabet[nrng.integers(0, len(alphabet), n)].tobytes(), f"soup {n}")


@pytest.mark.gpu
def test_skip_flag_blanks_the_bytes_in_front_of_the_window(dev, oracle):
    rng = random.Random(2)
    body, _ = _stream(rng, 30)
    for skip in range(16):
        for junk in (b'"', b"\\", b'}]"\\{["x', b"\xe4\xb8", b"1234567890123456"):
            head = (junk * 16)[:skip]
            idx, open_string, *_ = _gpu_window(dev, head + body, skip=skip)
            widx, wopen = helpers.oracle_window(oracle.msj_oracle_stage1, b" " * skip + body)
            assert np.array_equal(idx, widx) and open_string == wopen, (skip, junk)
    # a window shorter than one 64-byte block, and one of exactly the skipped bytes
    idx, *_ = _gpu_window(dev, b'"}' + b"[1]", skip=2)
    assert list(idx) == [2, 3, 4]
    idx, *_ = _gpu_window(dev, b'"}', skip=2)
    assert idx.size == 0


@pytest.mark.gpu
def test_document_stream_windows(dev, oracle):
    import torch

    from mojo_simdjson_amd.document_stream import DocumentStream, DocumentStreamError

    rng = random.Random(7)
    data, starts = _stream(rng, 6000)
    widx, _ = helpers.oracle_window(oracle.msj_oracle_stage1, data)
    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    longest = max(b - a for a, b in zip(starts, starts[1:] + [len(data)]))
    for window in (max(64, (longest + 64) // 16 * 16), 4096, 65536, 1 << 20, 1 << 28):
        if window < longest + 32:
            continue
        stream = DocumentStream(dev, d_buf, len(data), window=window)
        offsets, tokens = [], []
        for w in stream:
            assert w.base % 16 == 0 and w.n_documents > 0
            offsets += w.document_offsets()
            tokens.append(w.d_idx.cpu().numpy().view(np.uint32).astype(np.int64) + w.base)
        assert offsets == starts, window
        assert np.array_equal(np.concatenate(tokens), widx.astype(np.int64)), window
        assert stream.windows >= min(2, len(data) // window)
    # a document that does not fit; a stream that ends inside a document / inside a string
    with pytest.raises(DocumentStreamError) as e:
        list(DocumentStream(dev, d_buf, len(data), window=64))
    assert e.value.code == 1
    for tail, code in ((b' {"a":[1,2', 3), (b' {"a":"xy', 15), (b' "xy', 15)):
        bad = data + tail
        d_bad = torch.from_numpy(np.frombuffer(bad, dtype=np.uint8).copy()).to(dev.device)
        with pytest.raises(DocumentStreamError) as e:
            list(DocumentStream(dev, d_bad, len(bad), window=65536))
        assert e.value.code == code, tail
    with pytest.raises(DocumentStreamError) as e:
        list(DocumentStream(dev, torch.from_numpy(np.frombuffer(b'{"a":1} ] {"b":2}', dtype=np.uint8).copy()).to(dev.device)))
    assert e.value.code == 3


@pytest.mark.gpu
def test_document_stream_one_gib(dev):
    """1 GiB of concatenated 64 MiB documents (the bench unit, whose length is not a multiple of 16, so
    every window but the first starts off the 16-byte grid) and 1 GiB of NDJSON lines."""
    import torch

    from mojo_simdjson_amd import synth
    from mojo_simdjson_amd.document_stream import DocumentStream

    unit = synth.unit(64 << 20, synth.SEED_MINIFIED, 0, 0, False)
    ulen = int(unit.size)
    reps = 16
    d_unit = torch.from_numpy(unit).to(dev.device)
    d_buf = d_unit.repeat(reps)
    unit_idx, _ = helpers.oracle_window(helpers.load_oracle_fast().msj_fast_stage1, unit.tobytes())
    seen = 0
    for w in DocumentStream(dev, d_buf, ulen * reps, window=200 << 20):
        offs = w.document_offsets()
        assert offs == [ulen * (seen + k) for k in range(len(offs))]
        assert w.n_tokens == unit_idx.size * len(offs)
        first = w.d_idx[: unit_idx.size].cpu().numpy().view(np.uint32).astype(np.int64) + w.base - ulen * seen
        assert np.array_equal(first, unit_idx.astype(np.int64))
        seen += len(offs)
    assert seen == reps
    del d_buf
    block = b"".join(json.dumps({"id": i, "text": "t" * (i % 50), "tags": [i, i + 1], "user": {"name": "n", "ok": True}},
                                separators=(",", ":")).encode() + b"\n" for i in range(12000))
    nrep = (1 << 30) // len(block)
    d_buf = torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev.device).repeat(nrep)
    docs = 0
    for w in DocumentStream(dev, d_buf, len(block) * nrep, window=256 << 20):
        assert bool((w.d_type[w.d_doc_first.long()] == ord("{")).all())
        assert bool((w.d_depth[w.d_doc_first.long()] == 0).all())
        docs += w.n_documents
    assert docs == 12000 * nrep


@pytest.mark.gpu
def test_cpp_document_stream():
    """include/document_stream.hpp (the C++ mirror above the C ABI) through tests/cpp/test_document_stream.cpp."""
    import os
    import subprocess

    out = os.path.join(helpers.ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "test_document_stream")
    libdir = os.path.join(helpers.ROOT, "mojo_simdjson_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(helpers.ROOT, "include"),
                           os.path.join(helpers.ROOT, "tests", "cpp", "test_document_stream.cpp"), "-o", exe,
                           "-L" + libdir, "-lmsj_stage1", "-Wl,-rpath," + libdir])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "test_document_stream ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
