"""GPU parity tests: the HIP path, called through the C ABI, against the oracle.

Bit-exact bar: index array, count, the three trailer words and the return code
must equal the oracle's (oracle/stage1_oracle.c, pinned to the reference's
golden fixtures by tests/test_oracle.py) on the same inputs.  The UTF-8 verdict
(not part of reference parity: the reference's checker is a stub) is checked
against CPython's strict decoder via the oracle's validator.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

TILE = 16384  # test sizes are built from this: four of the kernel's 4 KiB tiles, half a 32 KiB ticket range


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available(), "no GPU visible"
    return torch


@pytest.fixture(scope="module")
def dev(torch_mod):
    from mojo_simdjson_amd.device import Stage1Device

    d = Stage1Device(0)
    yield d
    d.close()


def host_stage1(data, flags=0):
    """Through the host-pointer C ABI (what the Mojo shim would call)."""
    from mojo_simdjson_amd import DomParserImplementation

    p = DomParserImplementation()
    rc = p.stage1(bytes(data), flags)
    return rc, p


FLAG_TWO_PASS = 0x100
FLAG_DEBUG_STALL = 0x200


def assert_matches_oracle(oracle, data, where="", flags=0):
    data = bytes(data)
    want = helpers.run_oracle(oracle.msj_oracle_stage1, data)
    rc, p = host_stage1(data, flags)
    assert rc == want[0], f"{where}: code {rc} != oracle {want[0]} (len {len(data)})"
    if want[1] is not None:
        n = want[1]
        assert p.n_structural_indexes == n, f"{where}: n {p.n_structural_indexes} != {n}"
        got = p.structural_indexes[: n + 3]
        if not np.array_equal(got, want[2]):
            bad = int(np.argmax(got != want[2]))
            raise AssertionError(f"{where}: first index mismatch at {bad}: {got[bad]} != {want[2][bad]}")
    assert p.utf8_verdict == oracle.msj_oracle_utf8(data, len(data)), f"{where}: utf8 verdict"


def device_indices(torch, dev, d_buf, length, cap, flags=0):
    d_idx = torch.full((cap,), -1, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res, flags=flags, length=length)
    res = dev.fetch(d_res)
    return d_idx, res


# ----------------------------------------------------------------------------
# the reference's own test, tests/test_stage_1.mojo, through the host mirror
def test_simple_json_reference_fixtures():
    files = helpers.golden_valid_files()
    assert len(files) > 5
    for f in files:
        js, mask = helpers.read_fixture(f)
        rc, parser = host_stage1(js)
        assert rc == 0, "unexpected error code"
        n = parser.n_structural_indexes
        idx = parser.structural_indexes
        assert all(idx[i - 1] < idx[i] for i in range(1, n))
        assert helpers.mask_from_indices(idx[:n], len(mask)) == mask, f
        assert idx[n] == len(js) and idx[n + 1] == len(js) and idx[n + 2] == 0


def test_extra_pins(oracle):
    with open(os.path.join(helpers.ROOT, "tests", "golden", "extra_pins.json")) as f:
        pins = json.load(f)
    for p in pins:
        d = bytes.fromhex(p["input_hex"])
        rc, parser = host_stage1(d)
        assert rc == p["code"], p["name"]
        if p["indices"] is not None:
            n = parser.n_structural_indexes
            assert list(parser.structural_indexes[:n]) == p["indices"], p["name"]
            assert list(parser.structural_indexes[n:n + 3]) == [len(d), len(d), 0], p["name"]
        assert parser.utf8_verdict == p["utf8"], p["name"]


def test_error_codes_and_empty():
    from mojo_simdjson_amd import errors

    assert host_stage1(b"")[0] == errors.EMPTY
    assert host_stage1(b"  \n ")[0] == errors.EMPTY
    assert host_stage1(b'"abc')[0] == errors.UNCLOSED_STRING
    assert host_stage1(b'["a\nb"]')[0] == errors.UNESCAPED_CHARS
    assert host_stage1(b'["a\x01b')[0] == errors.UNCLOSED_STRING
    assert host_stage1(b'["\xff"]')[0] == errors.SUCCESS            # reference parity ignores utf8
    assert host_stage1(b'["\xff"]', flags=1)[0] == errors.UTF8_ERROR  # strict flag
    rc, p = host_stage1(b"[[[[")
    assert rc == 0 and list(p.structural_indexes[:7]) == [0, 1, 2, 3, 4, 4, 0]


def test_fuzz_small(oracle):
    for i, d in enumerate(helpers.fuzz_inputs(4242, 1500)):
        assert_matches_oracle(oracle, d, f"fuzz#{i}")


def test_tile_boundaries(oracle):
    """Lengths around the 64-byte block, the 4 KiB wave and the 16 KiB tile."""
    import random

    rng = random.Random(77)
    alpha = b'\\\\"""[]{}:, \n\tab01-\x01\xc3\xa9tfn'
    lens = [4095, 4096, 4097, TILE - 1, TILE, TILE + 1, TILE + 63, TILE + 64, TILE + 65,
            2 * TILE - 1, 2 * TILE, 2 * TILE + 1, 5 * TILE + 777, 70 * TILE + 5]
    for n in lens:
        d = bytes(rng.choice(alpha) for _ in range(n))
        assert_matches_oracle(oracle, d, f"rand len {n}")
        body = (b'{"k":[1,2,"x\\"y",true,null,-3.5e2],"s":"a\\\\"} ' * (n // 40 + 2))[: n - 1] + b"7"
        assert_matches_oracle(oracle, body, f"json len {n}")


def test_backslash_runs_across_boundaries(oracle):
    """H6: long backslash runs ending exactly at lane / wave / tile boundaries,
    including >= 64 of them in front of a tile (descriptor fallback path)."""
    for boundary in (64, 4096, TILE, 2 * TILE, 3 * TILE):
        for run in (1, 2, 3, 61, 62, 63, 64, 65, 127, 128, 129, 200):
            for shift in (-1, 0, 1):
                pre = boundary + shift - run - 1
                if pre < 1:
                    continue
                d = b'"' + b"a" * (pre - 1) + b"\\" * run + b'" , "x" ] ' + b"1" * 40
                assert_matches_oracle(oracle, d, f"run {run} ending at {boundary + shift}")
    # whole tiles of backslashes (escape carry must propagate through a tile)
    for run in (TILE, TILE + 1, 2 * TILE, 2 * TILE + 1, 3 * TILE + 5):
        for lead in (1, 2, 100):
            d = b" " * (lead - 1) + b'"' + b"\\" * run + b'"  "' + b"b" * 10 + b'"'
            assert_matches_oracle(oracle, d, f"tile run {run} lead {lead}")


def test_strings_spanning_tiles(oracle):
    """in_string carried through the look-back chain over many tiles; control
    characters that are only an error on the in-string side."""
    big = b'["' + b"s" * (5 * TILE + 123) + b'",' + b"\n" * 70 + b'"' + b"t" * (3 * TILE) + b'"]'
    assert_matches_oracle(oracle, big, "long strings")
    bad = b'["' + b"s" * (2 * TILE + 17) + b"\n" + b"s" * (TILE) + b'"]'
    assert_matches_oracle(oracle, bad, "newline inside a long string")
    unclosed = b'["' + b"s" * (4 * TILE)
    assert_matches_oracle(oracle, unclosed, "unclosed long string")
    many = (b'"a"' + b" " * 61) * (3 * TILE // 64)  # a quote pair in every lane
    assert_matches_oracle(oracle, many, "quote pair per lane")
    odd = (b'"' + b" " * 63) * (4 * TILE // 64)      # parity flips in every lane
    assert_matches_oracle(oracle, odd, "one quote per lane")


def test_quiet_tiles(oracle):
    """Tiles without a quote or a backslash take a shorter path (no escape / string work): next to every kind of
    neighbour, inside and outside a string, with control characters, scalars and UTF-8 in them."""
    from mojo_simdjson_amd import errors

    q = TILE
    cases = {
        "blanks between values": b"[1," + b" " * (3 * q) + b"2]",
        "digits only": b"[" + b"7" * (2 * q + 5) + b"," + b"8" * (q + 9) + b"]",
        "inside a string": b'["' + b"x" * (q - 2) + b"y" * (2 * q) + b'"]',
        "string ends on the tile edge": b'["' + b"x" * (q - 3) + b'"' + b" " * (2 * q) + b',"z"]',
        "backslash as the byte before a quiet tile": b'["' + b"x" * (q - 3) + b"\\" + b"n" * (2 * q) + b'"]',
        "escaped quote as the byte before": b'["' + b"x" * (q - 4) + b'\\"' + b"n" * (2 * q) + b'"]',
        "escaped backslash then quiet": b'["' + b"x" * (q - 4) + b"\\\\" + b"n" * (2 * q) + b'"]',
        "long backslash run in front (unresolved window)": b'["' + b"x" * (q - 2 - 70) + b"\\" * 70 + b"n" * (2 * q) + b'"]',
        "odd backslash run in front": b'["' + b"x" * (q - 2 - 71) + b"\\" * 71 + b"n" * (2 * q) + b'"]',
        "newline in a quiet tile inside a string": b'["' + b"x" * (q + 100) + b"\n" + b"x" * (2 * q) + b'"]',
        "newline in a quiet tile outside": b"[1," + b" " * (q + 100) + b"\n" + b" " * (2 * q) + b"2]",
        "scalar over the edge into a quiet tile": b"[" + b" " * (q - 3) + b"12345" + b" " * (2 * q) + b",true]",
        "scalar run through quiet tiles": b"[" + b" " * (q - 3) + b"1" * (2 * q + 9) + b",null]",
        "brackets only": b"[" * (q + 7) + b" " * q + b"]" * (q + 7),
        "utf-8 in a quiet tile": b'["' + "é漢😀".encode() * (q // 3) + b'"]',
        "utf-8 cut by the edges of quiet tiles": b'["x' + "漢".encode() * (3 * q // 3 + 11) + b'"]',
        "bad utf-8 in a quiet tile": b'["' + b"x" * (q + 50) + b"\xff" + b"x" * q + b'"]',
        "unclosed": b'["' + b"x" * (3 * q),
        # round 5: tiles of ONE byte value (blank or plain scalar) skip the transpose and the classification altogether
        "uniform: blanks from the first byte": b" " * (2 * q) + b"[1]",
        "uniform: scalar tile behind a blank (its first byte is structural)": b"[" + b" " * (q - 1) + b"7" * q + b"   ,1]",
        "uniform: scalar tile behind a scalar": b"[" + b" " * (q - 2) + b"8" + b"7" * q + b"   ,1]",
        "uniform: scalar tile behind a closing quote": b'["' + b"a" * (q - 3) + b'"' + b"7" * q + b"]",
        "uniform: scalar tile behind an operator": b"[" * q + b"7" * q + b"]" * q,
        "uniform: scalar tiles, nothing else": b"7" * (3 * q),
        "uniform: blanks, nothing else": b" " * (3 * q),
        "uniform: a cut UTF-8 character in front (pending continuation)": b'["' + b"x" * (q - 3) + b"\xc3" + b"x" * q + b'"]',
        "uniform: a cut UTF-8 character in front of blanks": b'["' + b"x" * (q - 4) + b'"' + b"\xe4" + b" " * q + b"]",
        "uniform: tiles of quotes (not plain: the long way)": b'"' * (2 * q),
        "uniform: tiles of commas": b"," * (2 * q),
        "uniform: tiles of NUL": b"[" + b"\x00" * (2 * q - 1) + b"]",
        "uniform: tiles of NUL in a string": b'["' + b"\x00" * (2 * q - 2) + b'"]',
        "uniform: tiles of DEL": b"[" + b"\x7f" * (2 * q - 1) + b",1]",
        "uniform: tiles of 0x80": b'["' + b"x" * (q - 2) + b"\x80" * q + b'"]',
        "uniform: all but the tile's last byte": b"[" + b" " * (q - 1) + b"7" * (q - 1) + b"," + b"7" * q + b"]",
        "uniform: all but byte 28 (between the filter's dwords)": b"[" + b" " * (q - 1) + b"7" * 28 + b"," + b"7" * (q - 29) + b"]",
        "uniform: all but byte 4 of the last block": b"[" + b" " * (q - 1) + b"7" * (q - 60) + b"," + b"7" * 59 + b"]",
        "uniform: escaped first byte": b'["' + b"x" * (q - 3) + b"\\" + b"7" * q + b'"]',
    }
    for name, d in cases.items():
        assert_matches_oracle(oracle, d, name)
        try:
            d.decode("utf-8")
        except UnicodeDecodeError:
            # the reference ignores UTF-8 (its checker is a stub); with the strict flag a structurally good document is rejected
            if helpers.run_oracle(oracle.msj_oracle_stage1, d)[0] == 0:
                assert host_stage1(d, flags=1)[0] == errors.UTF8_ERROR, name
        else:
            assert_matches_oracle(oracle, d, name + " (strict utf-8)", flags=1)
    for cut in (q - 1, q, q + 1, 2 * q + 63, 2 * q + 64):  # a quiet tile cut short by the end of the input
        assert_matches_oracle(oracle, (b"[1," + b" " * (3 * q))[:cut], f"quiet, cut at {cut}")
        assert_matches_oracle(oracle, (b'["' + b"x" * (3 * q))[:cut], f"quiet in a string, cut at {cut}")


def test_density_extremes(oracle):
    from mojo_simdjson_amd import synth

    # kinds 0/1/4/5/6/7: more than 1 020 structurals per tile (4 096, 2 731, 2 048, 1 638, 1 820, 2 341 per tile; 6 and 7
    # with a number of indices per 64-byte block that is not whole: two-round staging near its limit / block-wise
    # emission with a different count in every block); 2/3: none
    for kind in range(8):
        for n in (3 * TILE + 1000, 40 * TILE + 123):
            d = synth.extreme(n, kind).tobytes()
            assert_matches_oracle(oracle, d, f"extreme kind {kind} len {n}")
    # density changing from tile to tile: every emission path next to every other one
    parts = [synth.extreme(TILE + 7 * k, k % 8).tobytes() for k in range(32)]
    assert_matches_oracle(oracle, b" ".join(parts), "mixed densities")


def test_utf8_negative_variants(oracle):
    from mojo_simdjson_amd import synth

    base = synth.workload("utf8", 6 * TILE).tobytes()
    assert_matches_oracle(oracle, base, "utf8 workload")
    n = len(base)
    first_hi = next(i for i, c in enumerate(base) if c >= 0xE0)
    for off in (first_hi, TILE - 1, TILE, 2 * TILE + 64, n - 2):
        for val in (0xFF, 0xC0, 0x80):
            d = bytearray(base)
            d[off] = val
            assert_matches_oracle(oracle, d, f"corrupt {val:#x} at {off}")
    # sequences straddling block / wave / tile boundaries, and truncated at EOF
    for boundary in (64, 4096, TILE, 2 * TILE):
        for seq in (b"\xc3\xa9", b"\xe4\xb8\xad", b"\xf0\x9f\x98\x80", b"\xed\xa0\x80", b"\xf4\x90\x80\x80",
                    b"\xe0\x80\x80", b"\xf0\x80\x80\x80", b"\xe4\xb8", b"\xf0\x9f\x98"):
            for k in range(len(seq) + 1):
                pre = boundary - k
                d = b'"' + b"a" * (pre - 1) + seq + b'"' + b" " * 50
                assert_matches_oracle(oracle, d, f"seq {seq.hex()} split {k} at {boundary}")
                d2 = b'"' + b"a" * (pre - 1) + seq  # truncated document
                assert_matches_oracle(oracle, d2, f"seq {seq.hex()} eof {k} at {boundary}")


@pytest.mark.parametrize("name", ["minified", "utf8", "pretty2", "pretty8", "pretty_tab_crlf"])
def test_workload_units_device_path(torch_mod, dev, oracle, name):
    """8 MiB units of every bench workload through the device-resident API."""
    torch = torch_mod
    from mojo_simdjson_amd import synth

    u = synth.workload(name, 8 << 20)
    b = u.tobytes()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0
    d_buf = torch.from_numpy(u).to(dev.device)
    d_idx, res = device_indices(torch, dev, d_buf, len(b), len(b) + 3)
    assert res.code == 0 and res.count == n and res.internal_error == 0
    got = d_idx[: n + 3].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, idx)
    assert res.utf8_error == 0 and res.in_string == 0
    assert int(d_idx[n + 3].item()) == -1  # nothing written past the trailer


def test_dense_tiles_around_the_emission_limits(oracle):
    """BitIndexer.write (json_structural_indexer.mojo:46-58) where the emission changes its form: up to 1 020 indices
    per 4 KiB tile are staged in one round, up to 2 051 in two rounds of the staging slice (the lane that straddles
    the border writes in both), more go block by block.  Periodic documents with 1 024 and 2 048 structurals per tile,
    shifted by 0..12 leading brackets so that the tile counts and the alignment of the output (count mod 4) take every
    value around the limits; tiles whose blocks all hold the same multiple of 32 indices (all lanes on one LDS bank:
    left to the block-wise form); random soups between the densities."""
    import random

    for unit in (b"1234567,", b"123,", b"1234,", b"12,", b"[1,", b'"a",12,'):
        body = unit * ((5 * TILE) // len(unit))
        for pre in range(0, 13):
            assert_matches_oracle(oracle, b"[" * pre + body + b"1" + b"]" * pre, f"{unit!r} pre {pre}")
    rng = random.Random(31)
    for k in range(40):
        dens = rng.choice((0.22, 0.26, 0.3, 0.4, 0.45, 0.5, 0.55))
        n = rng.choice((3 * TILE + 7, 5 * TILE - 1))
        d = bytearray(rng.choice(b"123456789") for _ in range(n))
        for i in range(n):
            if rng.random() < dens / 2:
                d[i] = 0x2C  # a comma and (usually) the scalar behind it: two structurals
        # stretches of other densities inside the same tile: the lanes' counts differ widely
        for _ in range(6):
            a = rng.randrange(0, n - 600)
            d[a:a + 512] = rng.choice((b" " * 512, b"[" * 512, b"1,[" * 170 + b"  "))
        assert_matches_oracle(oracle, b"[" + bytes(d) + b"]", f"soup {k} density {dens}")


def test_capacity_clip(torch_mod, dev, oracle):
    torch = torch_mod
    from mojo_simdjson_amd import synth

    u = synth.workload("minified", 1 << 20)
    b = u.tobytes()
    _, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    d_buf = torch.from_numpy(u).to(dev.device)
    # exactly enough
    d_idx, res = device_indices(torch, dev, d_buf, len(b), n + 3)
    assert res.code == 0 and np.array_equal(d_idx.cpu().numpy().view(np.uint32), idx)
    # too small: CAPACITY, and no write past the buffer (guard word intact)
    cap = n // 2
    d_idx = torch.full((cap + 64,), -1, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.lib.msj_stage1_device(dev.ctx, ctypes.c_void_p(d_buf.data_ptr()), len(b),
                              ctypes.c_void_p(d_idx.data_ptr()), cap, ctypes.c_void_p(d_res.data_ptr()),
                              dev._stream(), 0)
    res = dev.fetch(d_res)
    assert res.code == 1 and res.count == n
    host = d_idx.cpu().numpy()
    assert np.array_equal(host[:cap].view(np.uint32), idx[:cap])
    assert (host[cap:] == -1).all()


def test_shard_chain_equals_whole(torch_mod, dev, oracle):
    """Cutting the stream at arbitrary (tile-aligned and unaligned-length) points and
    chaining msj_stage1_shard_device through device-resident carries reproduces the
    single-call result: the property the multi-GPU stitch relies on."""
    torch = torch_mod
    from mojo_simdjson_amd import synth

    u = synth.workload("utf8", 3 << 20)
    b = u.tobytes()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0
    d_all = torch.from_numpy(u).to(dev.device)
    cuts = [0, 5 * TILE, 5 * TILE + 64 * 1000, 100 * TILE + 16 * 777, len(b)]
    d_idx = torch.full((n + 3,), -1, dtype=torch.int32, device=dev.device)
    cin = dev.new_carry()
    for k in range(len(cuts) - 1):
        lo, hi = cuts[k], cuts[k + 1]
        cout = dev.new_carry()
        rc, nseg = dev.shard(d_all[lo:], hi - lo, d_idx, cin, cout, has_prefix=(k > 0),
                             is_final=(k == len(cuts) - 2), trailer_len=len(b))
        assert rc == 0 and nseg == 1
        cin = cout
    res = dev.fetch(cin)
    assert res.code == 0 and res.count == n and res.bytes == len(b)
    assert res.utf8_error == 0  # cuts inside multi-byte characters are not errors
    got = d_idx.cpu().numpy().view(np.uint32)
    # shard-relative offsets: add each shard's byte base back before comparing
    bases = np.zeros(n + 3, dtype=np.uint32)
    counts = []
    for k in range(len(cuts) - 1):
        lo, hi = cuts[k], cuts[k + 1]
        cnt = int(np.searchsorted(idx[:n], hi) - np.searchsorted(idx[:n], lo))
        counts.append(cnt)
    pos = 0
    for k, cnt in enumerate(counts):
        bases[pos:pos + cnt] = cuts[k]
        pos += cnt
    assert np.array_equal(got[:n] + bases[:n], idx[:n])
    assert list(got[n:n + 3]) == [len(b), len(b), 0]


def test_summary_pass_parity(torch_mod, dev, oracle):
    """no_emit summary pass returns the shard's quote parity and writes nothing."""
    torch = torch_mod
    d = (b'{"a":"' + b"x" * (2 * TILE) + b'","b":[1,2,"q')  # ends inside a string
    u = np.frombuffer(d, dtype=np.uint8)
    d_buf = torch.from_numpy(u.copy()).to(dev.device)
    cin, cout = dev.new_carry(), dev.new_carry()
    dev.shard(d_buf, len(d), None, cin, cout, no_emit=True, flags=2)
    res = dev.fetch(cout)
    assert res.in_string == 1
    want = helpers.run_oracle(oracle.msj_oracle_stage1_serial, d + b'"')
    assert res.count == want[1]


@pytest.mark.parametrize("name", ["minified", "utf8", "pretty4"])
def test_full_size_1gib_replication_property(torch_mod, dev, oracle, name):
    """BASELINE.json configs 2/3/4 at full size.  The oracle indexes one ~64 MiB
    unit; the 1 GiB buffer is that unit repeated, and because every unit ends
    with all carries at zero the expected index array is unit_idx + k*unit_len
    (checked on the device, not by shipping 4 GB to the host)."""
    torch = torch_mod
    from mojo_simdjson_amd import synth

    u = synth.workload(name, 64 << 20)
    b = u.tobytes()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0
    reps = (1 << 30) // len(b)
    d_unit = torch.from_numpy(u).to(dev.device)
    d_buf = d_unit.repeat(reps)
    total = d_buf.numel()
    assert total % 128 != 0
    cap = n * reps + 3
    d_idx, res = device_indices(torch, dev, d_buf, total, cap)
    assert res.code == 0 and res.count == n * reps and res.internal_error == 0
    unit_idx = torch.from_numpy(idx[:n].astype(np.int64)).to(dev.device)
    for k in range(reps):
        got = d_idx[k * n:(k + 1) * n].to(torch.int64) & 0xFFFFFFFF
        assert torch.equal(got, unit_idx + k * len(b)), f"repetition {k}"
    tail = d_idx[n * reps:n * reps + 3].to(torch.int64) & 0xFFFFFFFF
    assert tail.tolist() == [total, total, 0]
    # size-independent properties: strictly increasing, every index on a plausible byte
    v = d_idx[: n * reps].to(torch.int64) & 0xFFFFFFFF
    assert bool((v[1:] > v[:-1]).all())
    assert res.utf8_error == 0
    if name == "utf8":
        # negative variants at full size: one corrupted byte inside a string body (first
        # tile, around a tile boundary deep in the buffer, last bytes).  The index array
        # must not change; the strict verdict must flip.
        first_hi = next(i for i, c in enumerate(b[:4096]) if c >= 0xE0)
        unit_hi = np.flatnonzero(u >= 0xE0)
        deep = (reps // 2) * len(b)
        near_tile = int(unit_hi[np.searchsorted(unit_hi, (4096 * 1000) - 1)])
        last_hi = int(unit_hi[-1]) + (reps - 1) * len(b)
        for off in (first_hi, deep + near_tile, last_hi):
            saved = int(d_buf[off].item())
            d_buf[off] = 0xFF
            d_idx2, res2 = device_indices(torch, dev, d_buf, total, cap)
            assert res2.code == 0 and res2.utf8_error == 1 and res2.count == res.count
            assert torch.equal(d_idx2[: n * reps + 3], d_idx[: n * reps + 3])
            _, res3 = device_indices(torch, dev, d_buf, total, cap, flags=1)
            assert res3.code == 11
            d_buf[off] = saved



@pytest.mark.parametrize("kind,density", [(5, 0.4), (6, 4 / 9), (4, 0.5), (7, 4 / 7), (1, 2 / 3), (0, 1.0)])
def test_full_size_density_extremes(torch_mod, dev, oracle, kind, density):
    """BASELINE.json config 4's synthetic extremes at full size (1 GiB): [1234,...] d = 0.4 (two rounds of the 32-bit
    staging slice per tile, kEmitStaged2), [123,1234,...] d = 0.44 (the same with 1 820 of its 2 051 slots used),
    [123,...] d = 0.5 (every block holds 32 indices: all lanes on one LDS bank, so the tile is left to the block-wise
    form), [12,123,...] d = 0.57, [10,...] d = 0.67 and [[[[...]]]] d = 1.0 (emit_dense: block by block, the block's
    mask as EXEC, straight to the output).
    Same replication property as the workloads above: every unit ends with all carries at zero, so
    the expected index array is unit_idx + k * unit_len, compared index by index on the device."""
    torch = torch_mod
    from mojo_simdjson_amd import synth

    u = synth.extreme((64 << 20) - 52, kind)
    b = u.tobytes()
    assert len(b) % 4096 != 0
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0 and abs(n / len(b) - density) < 0.01
    reps = (1 << 30) // len(b)
    d_buf = torch.from_numpy(u).to(dev.device).repeat(reps)
    total = d_buf.numel()
    cap = n * reps + 3
    d_idx, res = device_indices(torch, dev, d_buf, total, cap)
    assert res.code == 0 and res.count == n * reps and res.internal_error == 0 and res.utf8_error == 0
    unit_idx = torch.from_numpy(idx[:n].astype(np.int64)).to(dev.device)
    for k in range(reps):
        got = d_idx[k * n:(k + 1) * n].to(torch.int64) & 0xFFFFFFFF
        assert torch.equal(got, unit_idx + k * len(b)), f"repetition {k}"
        del got
    tail = d_idx[n * reps:n * reps + 3].to(torch.int64) & 0xFFFFFFFF
    assert tail.tolist() == [total, total, 0]


def test_two_pass_path_equals_oracle(oracle):
    """MSJ_FLAG_TWO_PASS: the kernels the library falls back to when a single-pass launch expires a wait
    (summary / scan / emission, no inter-workgroup waiting) give the oracle's result bit for bit -- fuzz,
    the reference's fixtures, tile boundaries, every density, and the tiles the scan pass has to compute
    itself (>= 63 backslashes in front of a tile, whole tiles of backslashes)."""
    import random
    from mojo_simdjson_amd import synth

    for f in helpers.golden_valid_files():
        js, _ = helpers.read_fixture(f)
        assert_matches_oracle(oracle, js, f, FLAG_TWO_PASS)
    for i, d in enumerate(helpers.fuzz_inputs(777, 500)):
        assert_matches_oracle(oracle, d, f"fuzz#{i}", FLAG_TWO_PASS)
    rng = random.Random(5)
    alpha = b'\\\\"""[]{}:, \n\tab01-\x01\xc3\xa9tfn'
    for n in (4095, 4096, 4097, TILE + 63, 2 * TILE + 1, 5 * TILE + 777, 70 * TILE + 5):
        d = bytes(rng.choice(alpha) for _ in range(n))
        assert_matches_oracle(oracle, d, f"rand len {n}", FLAG_TWO_PASS)
    for boundary in (4096, TILE, 3 * TILE):
        for run in (1, 62, 63, 64, 65, 129, 200):
            for shift in (-1, 0, 1):
                pre = boundary + shift - run - 1
                d = b'"' + b"a" * (pre - 1) + b"\\" * run + b'" , "x" ] ' + b"1" * 40
                assert_matches_oracle(oracle, d, f"run {run} ending at {boundary + shift}", FLAG_TWO_PASS)
    for run in (TILE, 2 * TILE + 1, 70 * 4096 + 5):
        for lead in (1, 100):
            d = b" " * (lead - 1) + b'"' + b"\\" * run + b'"  "' + b"b" * 10 + b'"'
            assert_matches_oracle(oracle, d, f"tile run {run} lead {lead}", FLAG_TWO_PASS)
    for kind in range(6):
        d = synth.extreme(40 * TILE + 123, kind).tobytes()
        assert_matches_oracle(oracle, d, f"extreme kind {kind}", FLAG_TWO_PASS)
    for name in ("minified", "utf8", "pretty4"):
        d = synth.workload(name, 8 << 20).tobytes()
        assert_matches_oracle(oracle, d, name, FLAG_TWO_PASS)
    # error codes and the strict UTF-8 verdict come out of the scan pass's finish()
    assert host_stage1(b'["abc', FLAG_TWO_PASS)[0] == 15
    assert host_stage1(b'["a\nb"]', FLAG_TWO_PASS)[0] == 14
    assert host_stage1(b"   ", FLAG_TWO_PASS)[0] == 13
    assert host_stage1(b'["\xff"]', FLAG_TWO_PASS | 1)[0] == 11


def test_expired_wait_falls_back_to_two_pass(torch_mod, dev, oracle):
    """A single-pass launch whose waits expire (test hooks: the resolver idles ~2 ms, the bound of every wait
    is lowered to 20 us) poisons its result; msj_carry_fetch then re-issues the call through the two-pass
    kernels, so the caller still gets the oracle's indices and code 0 -- never UNEXPECTED_ERROR (24)."""
    torch = torch_mod
    from mojo_simdjson_amd import synth

    u = synth.workload("minified", 32 << 20)
    b = u.tobytes()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0
    d_buf = torch.from_numpy(u).to(dev.device)
    before = dev.fallback_count()
    dev.set_wait_ticks(2000)
    try:
        d_idx, res = device_indices(torch, dev, d_buf, len(b), n + 3, flags=FLAG_DEBUG_STALL)
    finally:
        dev.set_wait_ticks(200000000)
    assert dev.fallback_count() == before + 1, "the stalled launch did not expire a wait"
    assert res.code == 0 and res.count == n and res.internal_error == 0
    assert np.array_equal(d_idx[: n + 3].cpu().numpy().view(np.uint32), idx[: n + 3])
    # and the context is healthy afterwards: the next single-pass launch is clean
    d_idx2, res2 = device_indices(torch, dev, d_buf, len(b), n + 3)
    assert dev.fallback_count() == before + 1 and res2.code == 0 and res2.count == n
    assert torch.equal(d_idx2[: n + 3], d_idx[: n + 3])

def _sharded_dataset(name):
    from mojo_simdjson_amd import synth

    if name == "guesswrong":
        # the cut falls inside a string of a letter that also occurs outside of strings, which goes on for more than
        # the 1 MiB a rank looks at before it guesses, and whose closing quote follows a ':' -> the in_string
        # speculation of rank 1 is wrong and it must run again with the exact carry
        return b'["' + b"a" * (3 << 20) + b':",1,2,"zz"]'
    u = synth.workload(name, 3 << 20)
    data = u.tobytes() * 2
    if name == "minified":  # extra tail with escapes
        data = data[:-1] + b',"k":"' + b"\\\\" * 40 + b'\\"x"}'
    return data


_SHARDED_SETS = ("utf8", "minified", "guesswrong")


def _sharded_worker(rank, world, port, q):
    """Two ranks sharing cuda:0 (gloo for the tiny collective): the same code path the
    8-GPU bench runs over RCCL, checked against the oracle on the whole stream."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mojo_simdjson_amd.device import Stage1Device
        from mojo_simdjson_amd.sharded import ShardedStage1

        dev = Stage1Device(0)
        results = []
        for name in _SHARDED_SETS:
            data = _sharded_dataset(name)
            total = len(data)
            shard_nominal = -(-total // world)
            shard_nominal = -(-shard_nominal // 16384) * 16384
            start = rank * shard_nominal
            end = min(total, start + shard_nominal)
            halo = 64 if rank > 0 else 0
            arr = np.frombuffer(data[start - halo:end], dtype=np.uint8).copy()
            d_alloc = torch.from_numpy(arr).to(dev.device)
            d_shard = d_alloc[halo:]
            d_idx = torch.full(((end - start) + 3,), -1, dtype=torch.int32, device=dev.device)
            sh = ShardedStage1(dev, rank, world)
            code, total_count, c = sh.run(d_shard, end - start, d_idx, total, has_prefix=(rank > 0))
            reruns_first = sh.reruns
            assert sh.last_placement[1:] == (start, int(c.count), end - start)
            index_begin = sh.last_placement[0]
            # two submissions in flight (what bench.py does for N > 1) give the same answer
            d_idx2 = torch.full_like(d_idx, -1)
            t1 = sh.submit(d_shard, end - start, d_idx2, total, has_prefix=(rank > 0))
            t2 = sh.submit(d_shard, end - start, d_idx, total, has_prefix=(rank > 0))
            r1, r2 = sh.result(t1), sh.result(t2)
            assert (r1[0], r1[1], int(r1[2].count)) == (code, total_count, int(c.count)) == (r2[0], r2[1], int(r2[2].count))
            assert torch.equal(d_idx2[: int(c.count)], d_idx[: int(c.count)])
            got = d_idx[: int(c.count) + (3 if rank == world - 1 else 0)].cpu().numpy().view(np.uint32)
            results.append((name, code, total_count, int(c.count), start, got.tobytes(), reruns_first, index_begin))
        q.put((rank, results))
        dev.close()
    finally:
        dist.destroy_process_group()


def test_sharded_two_ranks_one_gpu(oracle):
    import socket

    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for i, name in enumerate(_SHARDED_SETS):
        data = _sharded_dataset(name)
        code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, data)
        assert code == 0
        merged = []
        for rank in range(2):
            rname, rcode, total_count, cnt, start, raw, reruns, index_begin = out[rank][i]
            assert rname == name and rcode == code and total_count == n
            # the stitched offset: this shard's first index is index `index_begin` of the stream-wide array
            assert index_begin == int(np.searchsorted(idx[:n], start)) and (rank > 0 or index_begin == 0)
            if name == "guesswrong" and rank == 1:
                assert reruns == 1  # the refuted speculation was repaired by exactly one re-run
            vals = np.frombuffer(raw, dtype=np.uint32).astype(np.int64)
            if rank == 1:
                assert list(vals[-3:]) == [len(data), len(data), 0]
                vals = vals[:-3]
            merged.append(vals + start)  # shard-relative offsets -> stream offsets
        merged = np.concatenate(merged)
        assert np.array_equal(merged, idx[:n].astype(np.int64)), name



def test_sharded_world8_one_gpu(torch_mod, oracle):
    """The shape of BASELINE.json config 5 (one stream over 8 ranks) on ONE GPU and in ONE process: eight
    msj_ctx / msj_sharded pairs driven by eight threads through the library's C entry points
    (msj_stage1_sharded_submit / _result), with a loopback exchange in place of RCCL (the GPU box admits at most
    six processes on its card, and RCCL refuses two ranks on one device).  Cuts fall inside strings, right
    after backslash runs, inside 4-byte characters and inside a run of 70 backslashes; one rank's shard spans
    two uint32 segments (test hook: 64 KiB segments); several in_string guesses are refuted.  Checked against the
    oracle on the whole stream: code, total, every index of every shard, the trailer, the UTF-8 verdict."""
    import threading

    torch = torch_mod
    from mojo_simdjson_amd import sharded, synth
    from mojo_simdjson_amd._lib import MsjCarry, MsjSegment
    from mojo_simdjson_amd.device import Stage1Device

    world = 8
    u = synth.workload("utf8", 1 << 20).tobytes()
    # (a string longer than the 1 MiB of its head a rank looks at before it guesses, made of a letter that also occurs
    # outside of strings: nothing contradicts either hypothesis and the guess of the rank that starts inside it is wrong)
    parts = [u, b' ["' + b"a" * ((1 << 20) + 8000) + b':",1,2,"zz"] ', synth.workload("minified", 1 << 20).tobytes()]
    p1 = len(parts[0])
    p3 = sum(len(x) for x in parts)
    head3 = b' ["x' + b"\\" * 70 + b'","'
    pad = b"y" * ((2 - (p3 + len(head3))) % 4)  # the first emoji starts at an offset = 2 mod 4: an aligned cut splits one
    parts += [head3 + pad + b"\xf0\x9f\x98\x80" * 24 + b'"] ', synth.workload("pretty2", 512 << 10).tobytes()]
    data = b"".join(parts)
    total = len(data)
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, data)
    assert code == 0 and oracle.msj_oracle_utf8(data, total) == 0

    def align(x):
        return x // 16 * 16

    emoji = data.index(b"\xf0\x9f\x98\x80", p3)
    cut_emoji = align(emoji + 48)
    assert (cut_emoji - emoji) % 4 == 2 and data[cut_emoji] in (0x98, 0x9F, 0x80)  # inside a 4-byte character
    cuts = [0, align(p1 // 2), align(p1 + 1600),            # inside the ':"' string: a refuted guess
            align(p1 + len(parts[1]) + 300000), align(p3 + 48),   # inside the run of 70 backslashes
            cut_emoji, align(p3 + len(parts[3]) + 100000), align(total - 200000), total]
    assert len(set(cuts)) == world + 1 and cuts == sorted(cuts)
    L = sharded.lib()
    devs = [Stage1Device(0) for _ in range(world)]
    for d in devs:
        assert L.msj_debug_set_segment_bytes(d.ctx, 64 << 10) == 0
    d_data = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(devs[0].device)
    barrier = threading.Barrier(world)
    mine_ptrs = [None] * world
    lock = threading.Lock()
    L.msj_copy_to_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    L.msj_copy_to_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p]
    results = [None] * world
    errors_seen = []

    def rank_main(rank):
        try:
            dev = devs[rank]
            stream = torch.cuda.Stream(device=dev.device)
            sp = ctypes.c_void_p(stream.cuda_stream)

            def allgather(comm, d_send, d_recv, nbytes, st):
                # loopback: every rank reads every rank's report straight from device memory (one process)
                if L.msj_copy_to_host(dev.ctx, (ctypes.c_uint8 * 1)(), ctypes.c_void_p(d_send), 1, ctypes.c_void_p(st)) != 0:
                    return -3  # (synchronises this rank's stream: its kernel has finished)
                mine_ptrs[rank] = d_send
                barrier.wait()
                blob = b""
                for g in range(world):
                    buf = (ctypes.c_uint8 * nbytes)()
                    if L.msj_copy_to_host(dev.ctx, buf, ctypes.c_void_p(mine_ptrs[g]), nbytes, ctypes.c_void_p(st)) != 0:
                        return -3
                    blob += bytes(buf)
                rc = L.msj_copy_to_device(dev.ctx, ctypes.c_void_p(d_recv), blob, len(blob), ctypes.c_void_p(st))
                barrier.wait()
                return 0 if rc == 0 else -3

            cb = sharded.ALLGATHER_FN(allgather)
            x = sharded.MsjExchange(None, cb, rank, world, 0, 0)
            h = ctypes.c_void_p()
            assert L.msj_sharded_create(dev.ctx, ctypes.byref(x), None, ctypes.byref(h)) == 0
            lo, hi = cuts[rank], cuts[rank + 1]
            d_shard = d_data[lo:hi]
            d_idx = torch.full((hi - lo + 3,), -1, dtype=torch.int32, device=dev.device)
            nseg = -(-(hi - lo) // (64 << 10))
            d_seg = torch.zeros(nseg * 32, dtype=torch.uint8, device=dev.device)
            stream.wait_stream(torch.cuda.current_stream(dev.device))  # the fills above ran on this thread's current stream
            ticket = ctypes.c_uint32()
            rc = L.msj_stage1_sharded_submit(h, ctypes.c_void_p(d_shard.data_ptr()), hi - lo, ctypes.c_void_p(d_idx.data_ptr()),
                                             d_idx.numel(), total, int(rank > 0), None, ctypes.c_void_p(d_seg.data_ptr()), nseg,
                                             sp, 0, ctypes.byref(ticket))
            assert rc == 0, rc
            rcode, rtotal = ctypes.c_int32(), ctypes.c_uint64()
            local, used, place = MsjCarry(), MsjCarry(), sharded.MsjShardPlacement()
            rc = L.msj_stage1_sharded_result(h, ticket.value, ctypes.byref(rcode), ctypes.byref(rtotal), ctypes.byref(local),
                                             ctypes.byref(used), ctypes.byref(place))
            assert rc == 0, rc
            stream.synchronize()
            cnt = int(local.count)
            assert (int(place.byte_base), int(place.count), int(place.bytes)) == (lo, cnt, hi - lo)
            segs = np.frombuffer(d_seg.cpu().numpy().tobytes(), dtype=np.uint64).reshape(nseg, 4)
            vals = d_idx[: cnt + (3 if rank == world - 1 else 0)].cpu().numpy().view(np.uint32).astype(np.int64)
            # indices are relative to their segment's byte_base (msj_segment): back to stream offsets with the
            # placement the stitch returned
            out = vals[:cnt].copy()
            for base, blen, ibeg, c in segs:
                out[int(ibeg):int(ibeg) + int(c)] += int(base) + int(place.byte_base)
            results[rank] = (rcode.value, int(rtotal.value), out, vals[cnt:], int(L.msj_sharded_reruns(h)),
                             int(local.utf8_error), nseg, int(segs[:, 3].sum()) == cnt, int(place.index_begin))
            L.msj_sharded_destroy(h)
        except BaseException as exc:  # surface failures of worker threads
            with lock:
                errors_seen.append((rank, repr(exc)))
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors_seen, errors_seen
    assert all(r is not None for r in results)
    merged = np.concatenate([r[2] for r in results])
    assert all(r[0] == code and r[1] == n for r in results)
    assert np.array_equal(merged, idx[:n].astype(np.int64))
    # the stitched offsets address the oracle's global array: index_begin[g] + local k
    for g, r in enumerate(results):
        assert r[8] == int(np.searchsorted(idx[:n], cuts[g])), (g, r[8])
        assert np.array_equal(idx[r[8]:r[8] + len(r[2])].astype(np.int64), r[2])
    assert list(results[-1][3]) == [total & 0xFFFFFFFF, total & 0xFFFFFFFF, 0]
    assert all(r[7] for r in results) and max(r[6] for r in results) >= 2  # segment tables add up; some shard spans segments
    assert sum(r[4] for r in results) >= 1  # at least one refuted guess was repaired by a re-run
    assert all(r[5] == 0 for r in results)
    for d in devs:
        d.close()

def test_multi_segment_over_4gib(torch_mod, dev, oracle):
    """> 4 GiB in one shard call: chained segments with device-resident carries, offsets
    relative to each segment's byte base (the reference's UInt32
    offsets would silently wrap here, json_structural_indexer.mojo:138).  Checked with the
    replication property on the device."""
    torch = torch_mod
    from mojo_simdjson_amd import _lib, synth

    SEG = 0xFFFF0000
    u = synth.workload("minified", 64 << 20)
    b = u.tobytes()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0
    reps = SEG // len(b) + 4
    d_unit = torch.from_numpy(u).to(dev.device)
    d_buf = d_unit.repeat(reps)
    total = d_buf.numel()
    assert total > SEG
    cap = n * reps + 3
    d_idx = torch.empty(cap, dtype=torch.int32, device=dev.device)
    segs = torch.zeros(4 * 32, dtype=torch.uint8, device=dev.device)
    cin, cout = dev.new_carry(), dev.new_carry()
    rc, nseg = dev.shard(d_buf, total, d_idx, cin, cout, segments=segs, is_final=True, trailer_len=total)
    assert rc == 0 and nseg == 2
    res = dev.fetch(cout)
    assert res.code == 0 and res.count == n * reps and res.bytes == total and res.internal_error == 0
    table = np.frombuffer(segs.cpu().numpy().tobytes(), dtype=np.uint64).reshape(4, 4)
    assert list(table[0][:2]) == [0, SEG] and list(table[1][:2]) == [SEG, total - SEG]
    c0, c1 = int(table[0][3]), int(table[1][3])
    assert int(table[0][2]) == 0 and int(table[1][2]) == c0 and c0 + c1 == n * reps
    unit_idx = torch.from_numpy(idx[:n].astype(np.int64)).to(dev.device)
    for k in range(reps):
        want = unit_idx + k * len(b)
        want = torch.where(want >= SEG, want - SEG, want)  # segment-relative
        got = d_idx[k * n:(k + 1) * n].to(torch.int64) & 0xFFFFFFFF
        assert torch.equal(got, want), f"repetition {k}"
    assert c0 == int((unit_idx[None, :] + (torch.arange(reps, device=dev.device) * len(b))[:, None] < SEG).sum())
    tail = (d_idx[n * reps:n * reps + 3].to(torch.int64) & 0xFFFFFFFF).tolist()
    assert tail == [total & 0xFFFFFFFF, total & 0xFFFFFFFF, 0]


@pytest.mark.parametrize("shape", ["all_brackets", "d_two_thirds"])
def test_one_segment_at_the_uint32_edge(torch_mod, dev, shape):
    """the largest segment one launch indexes (0xFFFF0000 bytes) at the densities where the
    32-bit launch-relative count of the range prefixes (stage1_kernel.hip: 31:0 of pre[]) comes closest to its limit:
    `[[[[...` -- every byte structural, 4 294 901 760 indices = 17 GB, index k = k -- and `[10,10,...` at 3.9 GiB
    (2.8 G indices).  Checked on the device against the closed form, in slices."""
    torch = torch_mod
    SEG = 0xFFFF0000
    if shape == "all_brackets":
        n_bytes = SEG
        d_buf = torch.full((n_bytes,), ord("["), dtype=torch.uint8, device=dev.device)
        want_n = n_bytes

        def expected(a, b):  # indices a .. b-1
            return torch.arange(a, b, dtype=torch.int64, device=dev.device)
    else:
        m = (int(3.9 * (1 << 30)) - 1) // 3
        n_bytes = 1 + 3 * m
        d_buf = torch.empty(n_bytes + 15, dtype=torch.uint8, device=dev.device)[:n_bytes]
        d_buf[0] = ord("[")
        d_buf[1:].view(m, 3).copy_(torch.tensor(list(b"10,"), dtype=torch.uint8, device=dev.device).expand(m, 3))
        want_n = 1 + 2 * m  # '[', then '1' and ',' of every "10,"

        def expected(a, b):
            k = torch.arange(a, b, dtype=torch.int64, device=dev.device)
            j = (k - 1) // 2
            e = torch.where((k - 1) % 2 == 0, 1 + 3 * j, 3 + 3 * j)
            return torch.where(k == 0, torch.zeros_like(k), e)
    assert d_buf.data_ptr() % 16 == 0
    d_idx = torch.empty(want_n + 3 + 1, dtype=torch.int32, device=dev.device)
    cin, cout = dev.new_carry(), dev.new_carry()
    rc, nseg = dev.shard(d_buf, n_bytes, d_idx, cin, cout, is_final=True, trailer_len=n_bytes)
    assert rc == 0 and nseg == 1
    res = dev.fetch(cout)
    assert (res.code, res.count, res.bytes, res.internal_error, res.capacity_error) == (0, want_n, n_bytes, 0, 0)
    step = 1 << 28
    for a in range(0, want_n, step):
        b = min(want_n, a + step)
        got = d_idx[a:b].to(torch.int64) & 0xFFFFFFFF
        want = expected(a, b)
        if not torch.equal(got, want):
            bad = int((got != want).nonzero()[0]) + a
            raise AssertionError(f"{shape}: index {bad}: {int(d_idx[bad]) & 0xFFFFFFFF} != {int(expected(bad, bad + 1)[0])}")
        del got, want
    tail = (d_idx[want_n:want_n + 3].to(torch.int64) & 0xFFFFFFFF).tolist()
    assert tail == [n_bytes & 0xFFFFFFFF, n_bytes & 0xFFFFFFFF, 0]


def test_config5_share_of_a_non_first_rank_8gib(torch_mod, dev, oracle):
    """BASELINE.json config 5 as one of its ranks sees it, at full size, index by index: the stream is 1 024 units
    (64 GiB) cut into 8 byte ranges of ~8 GiB; this is the exact share of a rank > 0 whose cut falls INSIDE A STRING
    of a unit -- 64-byte halo in front (has_prefix), carry-in from msj_shard_speculate on the halo and the shard's
    first 4 KiB (checked against the oracle's state at the cut), two uint32 segments, is_final = 0.  Every index is
    compared on the device with the oracle's unit indices + k * unit_len (tests/replication.py), placed with the
    segment table the launch wrote; the carry out is the oracle's state at the shard's last byte.  (The reference's
    own harness compares every index and the trailer, tests/test_stage_1.mojo:43-82; its UInt32 offsets would wrap
    where the segments start, json_structural_indexer.mojo:138.)"""
    torch = torch_mod
    from mojo_simdjson_amd import sharded, synth
    from tests import replication

    world, SEG = 8, 0xFFFF0000
    u = synth.workload("minified", 64 << 20)
    b = u.tobytes()
    L = len(b)
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0
    u_idx = idx[:n].astype(np.int64)
    total = (world * (8 << 30) // L) * L
    nominal = -(-(-(-total // world)) // 16384) * 16384
    fast = helpers.load_oracle_fast()

    def state_at(off):
        """(in_string, next_is_escaped, prev_scalar) after unit[:off]: in_string from the oracle (a prefix that ends
        inside a string is UNCLOSED_STRING), the other two from the bytes (tests/test_sharded_cpu.py does the same)."""
        scratch = np.zeros(off + 3, dtype=np.uint32)
        nn = ctypes.c_uint64(0)
        rc = fast.msj_fast_stage1(b[:off], off, scratch.ctypes.data, scratch.size, ctypes.byref(nn))
        assert rc in (0, 15), rc
        run = 0
        while b[off - 1 - run] == 0x5C:
            run += 1
        e = run & 1
        c = b[off - 1]
        nonscalar = c in (0x20, 0x09, 0x0A, 0x0D, 0x0C, 0x1A, 0x2C, 0x3A, 0x5B, 0x5D, 0x7B, 0x7D)
        if run:
            ps = 1
        elif c == 0x22:
            r2 = 0
            while b[off - 2 - r2] == 0x5C:
                r2 += 1
            ps = r2 & 1
        else:
            ps = int(not nonscalar)
        return int(rc == 15), e, ps

    rank = next(r for r in range(1, world) if state_at((r * nominal) % L)[0] == 1)  # a cut inside a string
    start = rank * nominal
    shard_len = min(total, start + nominal) - start
    assert shard_len > SEG and start % L != 0 and (start + shard_len) % L != 0
    d_unit = torch.from_numpy(u).to(dev.device)
    d_alloc = synth.stream_shard(d_unit, L, start - 64, shard_len + 64)
    d_shard = d_alloc[64:]
    assert d_shard.data_ptr() % 16 == 0
    halo = b[(start - 64) % L:(start - 64) % L + 64] if (start % L) >= 64 else None
    assert halo is not None
    head = d_shard[:4096].cpu().numpy().tobytes()
    assert d_alloc[:64].cpu().numpy().tobytes() == halo
    spec = sharded.speculate_bytes(halo, head)
    assert spec == state_at(start % L), "the speculation from the shard's own bytes is the oracle's state at the cut"
    ib = replication.expected_index_begin(u_idx, L, start)
    ie = replication.expected_index_begin(u_idx, L, start + shard_len)
    cap = ie - ib + 16
    d_idx = torch.empty(cap, dtype=torch.int32, device=dev.device)
    segs = torch.zeros(4 * 32, dtype=torch.uint8, device=dev.device)
    cin, cout = dev.make_carry(*spec), dev.new_carry()
    rc, nseg = dev.shard(d_shard, shard_len, d_idx, cin, cout, segments=segs, has_prefix=True, is_final=False,
                         trailer_len=total)
    assert rc == 0 and nseg == 2
    res = dev.fetch(cout)
    assert res.internal_error == 0 and res.capacity_error == 0 and res.utf8_error == 0 and res.unescaped_error == 0
    assert (res.count, res.bytes) == (ie - ib, shard_len)
    assert (res.in_string, res.next_is_escaped, res.prev_scalar) == state_at((start + shard_len) % L)
    table = np.frombuffer(segs.cpu().numpy().tobytes(), dtype=np.uint64).reshape(4, 4)[:2]
    assert [int(x) for x in table[0][:3]] == [0, SEG, 0] and [int(x) for x in table[1][:2]] == [SEG, shard_len - SEG]
    c0 = replication.expected_index_begin(u_idx, L, start + SEG) - ib
    assert [int(table[0][3]), int(table[1][2]), int(table[1][3])] == [c0, c0, ie - ib - c0]
    d_uidx = torch.from_numpy(u_idx).to(dev.device)
    bad, h = replication.check_shard(torch, d_idx, ie - ib, d_uidx, L, start, ib,
                                     [(int(r[0]), int(r[2]), int(r[3])) for r in table])
    assert bad == 0
    # and the hash is what the expected array itself gives (the piece bench.py's ranks add up)
    want = 0
    for a in range(ib, ie, 1 << 26):
        j = torch.arange(a, min(ie, a + (1 << 26)), dtype=torch.int64, device=dev.device)
        q = torch.div(j, n, rounding_mode="floor")
        want = (want + int(((d_uidx[j - q * n] + q * L + 1) * (2 * j + 1)).sum().item())) & replication.MASK64
    assert h == want


def test_capacity_clip_is_reported_by_non_final_shards(torch_mod, dev, oracle):
    """a non-final shard whose index buffer is too small used to clip its writes and say nothing
    (only a FINAL segment compared n + 3 with the capacity), so a sharded stream came back as SUCCESS.  The carry
    out now has a sticky capacity_error; a final shard still returns CAPACITY (1)."""
    torch = torch_mod
    from mojo_simdjson_amd import synth

    u = synth.workload("minified", 2 << 20)
    b = u.tobytes()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    d_buf = torch.from_numpy(u).to(dev.device)
    for cap, final, want_flag in ((n - 1000, False, 1), (n, False, 0), (n + 2, True, 1), (n + 3, True, 0)):
        d_idx = torch.full((n + 64,), -1, dtype=torch.int32, device=dev.device)
        cin, cout = dev.new_carry(), dev.new_carry()
        dev.shard(d_buf, len(b), d_idx[:cap], cin, cout, is_final=final, trailer_len=len(b))
        res = dev.fetch(cout)
        assert res.capacity_error == want_flag, (cap, final)
        assert res.code == (1 if (final and want_flag) else 0)
        got = d_idx.cpu().numpy().view(np.uint32)
        m = min(cap, n)
        assert np.array_equal(got[:m], idx[:m]) and (got[cap:] == 0xFFFFFFFF).all()  # clipped, nothing past the end
        # sticky along a chain: the next shard of the stream inherits it
        cout2 = dev.new_carry()
        dev.shard(d_buf, 4096, d_idx[:8], cout, cout2, is_final=False, no_emit=True)
        assert dev.fetch(cout2).capacity_error == want_flag


def test_single_document_over_a_segment_boundary(torch_mod, oracle):
    """msj_stage1_device takes documents up to 2^32 - 1 bytes but one launch indexes at most
    0xFFFF0000; a document in the last 64 KiB runs as two segments, and without a segment table the second one's
    offsets must stay relative to the document.  With the test hook's small segments: equal to the oracle."""
    torch = torch_mod
    from mojo_simdjson_amd import sharded, synth
    from mojo_simdjson_amd.device import Stage1Device

    d2 = Stage1Device(0)
    try:
        assert sharded.lib().msj_debug_set_segment_bytes(d2.ctx, 256 << 10) == 0
        u = synth.workload("minified", 1 << 20)
        b = u.tobytes()
        code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
        d_buf = torch.from_numpy(u).to(d2.device)
        d_idx, res = device_indices(torch, d2, d_buf, len(b), n + 3)
        assert res.code == code == 0 and res.count == n
        assert np.array_equal(d_idx[: n + 3].cpu().numpy().view(np.uint32), idx[: n + 3])
    finally:
        d2.close()


def test_host_pipeline_unavailable_falls_back_to_plain_staging(oracle):
    """when the chunked pinned pipeline cannot be set up (no pinned memory to be had, e.g. a memlock
    limit), large host-pointer inputs used to fail outright.  With the set-up forced to fail the call goes through
    the plain staging path -- same result -- and the context stops retrying."""
    from mojo_simdjson_amd import _lib, synth

    lib = _lib.load()
    u = synth.workload("pretty4", 40 << 20)
    b = u.tobytes()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    out = np.zeros(len(b) + 3, dtype=np.uint32)
    nn = ctypes.c_uint64(0)
    lib.msj_debug_set_pipeline_min_bytes(None, 24 << 20)
    try:
        assert lib.msj_debug_fail_pipeline_setup(None, 1) == 0
        for _ in range(2):
            out[:] = 0
            rc = lib.msj_stage1(b, len(b), out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(nn), None, 0)
            assert rc == code == 0 and nn.value == n and np.array_equal(out[: n + 3], idx[: n + 3])
        assert lib.msj_debug_fail_pipeline_setup(None, 1) == 1  # given up: the plain path from now on
        assert lib.msj_debug_fail_pipeline_setup(None, 0) == 0
        out[:] = 0
        rc = lib.msj_stage1(b, len(b), out.ctypes.data_as(ctypes.c_void_p), out.size, ctypes.byref(nn), None, 0)  # the pipeline again
        assert rc == 0 and nn.value == n and np.array_equal(out[: n + 3], idx[: n + 3])
    finally:
        lib.msj_debug_fail_pipeline_setup(None, 0)
        lib.msj_debug_set_pipeline_min_bytes(None, 0)


def _rccl_single_rank_worker(port, q, exchange="rccl"):
    """world_size 1 over the real "nccl" (= RCCL) backend: the asynchronous all-gather, the side
    stream and the pinned read-back of ShardedStage1 on the one GPU this box has."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from mojo_simdjson_amd.device import Stage1Device
        from mojo_simdjson_amd.sharded import ShardedStage1

        dev = Stage1Device(0)
        data = _sharded_dataset(_SHARDED_SETS[0])
        d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
        sh = ShardedStage1(dev, 0, 1, always_gather=True, exchange=exchange)
        bufs = [torch.full((len(data) + 3,), -1, dtype=torch.int32, device=dev.device) for _ in range(3)]
        tickets = []
        out = []
        for k in range(6):  # DEPTH submissions in flight at any time
            tickets.append(sh.submit(d_buf, len(data), bufs[k % 3], len(data), has_prefix=False))
            if len(tickets) == ShardedStage1.DEPTH:
                out.append(sh.result(tickets.pop(0)))
        while tickets:
            out.append(sh.result(tickets.pop(0)))
        code, total, c = out[-1]
        got = bufs[2][: total + 3].cpu().numpy().view(np.uint32).tobytes()
        st = sh.stats()
        assert sh.exchange_used == exchange and sh.rccl_ranks == (1 if exchange == "rccl" else 0)
        assert st["results"] == 6 and st["rounds"] == 6 and st["reruns"] == 0 and st["stitch_device_ns"] > 0
        assert st["kernel_device_ns"] > 0 and st["last_kernel_ns"] > 0 and st["reruns_behind_queue"] == 0
        assert sh.last_placement == (0, 0, total, len(data))
        q.put(([(o[0], o[1]) for o in out], got))
        dev.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["rccl", "torch"])
def test_sharded_rccl_plumbing_single_rank(oracle, exchange):
    """exchange "rccl": the library's own ncclAllGather on a communicator of the module's own; "torch": the
    fallback the module takes when it cannot get one -- the same 128 bytes through torch.distributed's RCCL group."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(29500 + (os.getpid() % 400) + 411 + (exchange == "torch"), q, exchange))
    p.start()
    res, got = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    data = _sharded_dataset(_SHARDED_SETS[0])
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, data)
    assert all(r == (code, n) for r in res), res
    want = np.concatenate([idx[:n], np.array([len(data), len(data), 0], dtype=np.uint32)]).astype(np.uint32)
    assert got == want.tobytes()


def _rccl_overlap_worker(port, q):
    """world 1 over the real RCCL backend, the library's own ncclAllGather, three
    submissions of a 3 GiB shard in flight: when result(0) returns, the kernel of submission 2 has not finished
    (msj_stage1_sharded_result used to hipStreamSynchronize the stream all three sit on)."""
    import time

    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from mojo_simdjson_amd import sharded, synth
        from mojo_simdjson_amd.device import Stage1Device

        dev = Stage1Device(0)
        unit = synth.workload("minified", 32 << 20)
        reps = (3 << 30) // unit.size
        d_buf = torch.from_numpy(unit).to(dev.device).repeat(reps)
        n_bytes = int(d_buf.numel())
        sh = sharded.ShardedStage1(dev, 0, 1, always_gather=True, exchange="rccl")
        d_idx = torch.empty(n_bytes // 2, dtype=torch.int32, device=dev.device)

        def three():
            tickets = [sh.submit(d_buf, n_bytes, d_idx, n_bytes, has_prefix=False) for _ in range(3)]
            t0 = time.perf_counter()
            r0 = sh.result(tickets[0])
            states = [sh.ticket_state(t) for t in tickets[1:]]
            t1 = time.perf_counter()
            rest = [sh.result(t) for t in tickets[1:]]
            t2 = time.perf_counter()
            return r0, states, rest, (t1 - t0) * 1e3, (t2 - t1) * 1e3

        three()  # warm-up: communicator, streams, clocks
        torch.cuda.synchronize()
        r0, states, rest, ms_first, ms_rest = three()
        st = sh.stats()
        q.put(dict(exchange=sh.exchange_used, states=states, codes=[r0[0]] + [r[0] for r in rest],
                   counts=[r0[1]] + [r[1] for r in rest], unit=unit.tobytes(), reps=reps, ms_first=ms_first, ms_rest=ms_rest,
                   last_kernel_ms=st["last_kernel_ns"] / 1e6, last_stitch_us=st["last_stitch_ns"] / 1e3, stats=st))
        sh.close()
        dev.close()
    finally:
        dist.destroy_process_group()


def test_sharded_result_does_not_drain_later_submissions(oracle):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_overlap_worker, args=(29500 + (os.getpid() % 400) + 433, q))
    p.start()
    r = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert r["exchange"] == "rccl"
    code, n, _ = helpers.run_oracle(oracle.msj_oracle_stage1, r["unit"])
    assert r["codes"] == [0, 0, 0] and code == 0 and r["counts"] == [n * r["reps"]] * 3
    from mojo_simdjson_amd import sharded

    # submission 2's kernel is still running (or queued) when result(0) has returned; the time result(0) blocked is
    # about one kernel, and the two later results cost the host about two kernels more -- not zero, as they did when
    # result(0) had drained the stream
    assert not (r["states"][1] & sharded.KERNEL_DONE), r
    assert r["ms_rest"] > 0.8 * r["last_kernel_ms"], r
    print(f"result(0) blocked {r['ms_first']:.3f} ms, results 1+2 {r['ms_rest']:.3f} ms more; kernel {r['last_kernel_ms']:.3f} ms, "
          f"kernel end -> reports in {r['last_stitch_us']:.1f} us")


def test_bench_line_of_the_sharded_path_rehearsed_on_one_rank():
    """The line the driver gets at N > 1 -- process group over RCCL, msj_stage1_sharded_submit / _result with the
    library's own ncclAllGather on the side stream, three submissions in flight, per-rank kernel-only time, stitch
    latency, skew, standalone rate and efficiency, device-side verification with the stitched offsets -- produced by
    `bench.py --rehearse-sharded` with ONE rank (all this box has): every field present and sane."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rehearse-sharded", "--steps", "8", "--warmup", "2",
                        "--settle-ms", "50", "--gib-per-gpu", "1"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    cfg, st = line["config"], line["config"]["stitch"]
    assert cfg["verified"] == "indices" and cfg["verify"]["mismatches"] == 0, cfg
    assert st["exchange"] == "rccl" and st["rccl_ranks"] == 1 and st["reruns"] == 0 and st["in_flight"] == 3, st
    assert st["allgather_rounds"] == 8 and len(st["kernel_only_ms"]) == 1 and 0.2 < st["kernel_only_ms"][0] < 1.0, st
    assert 0.2 < st["standalone_ms"][0] < 1.0 and st["reports_in_us"]["min_rank_median"] > 0, st
    eff = line["scaling_efficiency"]["value"]
    assert 0.7 < eff < 1.1, line["scaling_efficiency"]
    assert line["n_gpus"] == 1 and line["roofline"]["frac"] > 0.3 and "rehearsal" in cfg


def _shared_gpu_worker(k, barrier, q):
    """One of several processes hammering the same GPU at once: every kernel then has only part
    of its persistent workgroups resident, which is what the deadlock-freedom argument is about."""
    import torch

    from mojo_simdjson_amd import synth
    from mojo_simdjson_amd.device import Stage1Device

    dev = Stage1Device(0)
    u = synth.workload(("minified", "utf8", "pretty4")[k % 3], 32 << 20)
    d_buf = torch.from_numpy(u).to(dev.device).repeat(8)  # 256 MiB
    d_idx = torch.empty(d_buf.numel() // 2, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    barrier.wait()
    out = []
    for _ in range(8):
        dev.index(d_buf, d_idx, d_res)
        r = dev.fetch(d_res)
        out.append((int(r.code), int(r.count), int(r.internal_error)))
    q.put((k, u.tobytes(), out))
    dev.close()


def test_processes_sharing_one_gpu(oracle):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    n = 4
    q, barrier = ctx.Queue(), ctx.Barrier(n)
    procs = [ctx.Process(target=_shared_gpu_worker, args=(k, barrier, q)) for k in range(n)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(n)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, unit, out in got:
        code, cnt, _ = helpers.run_oracle(oracle.msj_oracle_stage1, unit)
        assert code == 0
        assert all(o == (0, 8 * cnt, 0) for o in out), (k, out, cnt)


def test_random_byte_soups():
    """A short run of tests/stress.py (randomised byte soups against the oracle, indices compared
    even when the reference returns an error code); the script runs for minutes by hand."""
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tests", "stress.py"), "12", "7"],
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "stress ok" in out.stdout


def test_host_pointer_large_inputs(oracle):
    """The host-pointer entry point on inputs of 50-64 MiB: same bits as the oracle, also for the
    error codes and the reference's capacity quirk."""
    from mojo_simdjson_amd import _lib, synth

    # from 24 MiB on through the chunked pipeline (the default threshold is 64 MiB: the largest input here is just over)
    assert _lib.load().msj_debug_set_pipeline_min_bytes(None, 24 << 20) == 0
    base = synth.workload("minified", 52 << 20).tobytes()
    assert len(base) > 48 << 20
    assert_matches_oracle(oracle, base, "52 MiB minified")
    u8 = synth.workload("utf8", 49 << 20).tobytes()
    assert_matches_oracle(oracle, u8 + b" " * ((64 << 20) + 5 - len(u8)), "utf8 padded to 64 MiB + 5 bytes")
    # error codes decided far apart
    bad = bytearray(base)
    bad[20 << 20] = 0x01 if base[(20 << 20) - 1:(20 << 20)] else 0x01
    assert_matches_oracle(oracle, bytes(bad), "control character 20 MiB in")
    assert_matches_oracle(oracle, base + b'"abc', "string left open at the very end")
    assert_matches_oracle(oracle, b'"' + base, "everything flipped by a quote in front")
    # the reference's capacity quirk: all structural, n + 3 > len
    dense = b"[" * (50 << 20)
    assert_matches_oracle(oracle, dense, "50 MiB of brackets")
    assert _lib.load().msj_debug_set_pipeline_min_bytes(None, 0) == 0
    assert_matches_oracle(oracle, base, "52 MiB minified, plain staging")


def test_host_pointer_registered_buffers(oracle):
    """msj_host_register: the caller's index array (and input) pinned once, the pipeline then moves them by DMA
    without its staging copies -- same bits as the oracle, also from the middle of a registered range, also after
    unregistering; and what the two calls refuse."""
    import ctypes

    from mojo_simdjson_amd import _lib, synth

    lib = _lib.load()
    assert lib.msj_debug_set_pipeline_min_bytes(None, 24 << 20) == 0  # 40 MiB: through the chunked pipeline
    data = synth.workload("minified", 40 << 20).tobytes() + b" [1,2]"
    code, n, want = helpers.run_oracle(oracle.msj_oracle_stage1, data)
    assert code == 0
    want = np.concatenate([want[:n], np.array([len(data), len(data), 0], dtype=np.uint32)])
    arena = np.zeros(len(data) + 3 + 1024, dtype=np.uint32)
    inbuf = np.frombuffer(data, dtype=np.uint8).copy()
    vp = lambda a, off=0: ctypes.c_void_p(a.ctypes.data + off)  # noqa: E731

    def run(idx_view, src):
        idx_view[:] = 0xDEADBEEF
        got_n, verdict = ctypes.c_uint64(0), ctypes.c_int32(-1)
        rc = lib.msj_stage1(vp(src), len(data), vp(idx_view), idx_view.size,
                            ctypes.byref(got_n), ctypes.byref(verdict), 0)
        assert (rc, got_n.value, verdict.value) == (0, n, 0)
        assert np.array_equal(idx_view[: n + 3], want)

    assert lib.msj_host_register(None, None, 10) == -1
    assert lib.msj_host_register(None, vp(arena), 0) == -1
    assert lib.msj_host_unregister(None, vp(arena)) == -1  # not registered
    assert lib.msj_host_register(None, vp(arena), arena.nbytes) == 0
    run(arena[: len(data) + 3], inbuf)                    # indices by DMA into the caller's array, input staged
    run(arena[512: 512 + len(data) + 3], inbuf)          # a view inside the registered range
    assert lib.msj_host_register(None, vp(inbuf), inbuf.nbytes) == 0
    run(arena[: len(data) + 3], inbuf)                    # both sides direct
    assert lib.msj_host_unregister(None, vp(arena, 64)) == -1  # not the start of a range
    assert lib.msj_host_unregister(None, vp(arena)) == 0
    run(arena[: len(data) + 3], inbuf)                    # input direct, indices staged again
    assert lib.msj_host_unregister(None, vp(inbuf)) == 0
    run(arena[: len(data) + 3], inbuf)
    assert lib.msj_debug_set_pipeline_min_bytes(None, 0) == 0


def test_shard_carry_out_at_any_length(torch_mod, dev, oracle):
    """A non-final shard may end anywhere, not only on a 4 KiB tile: its carry-out
    (next_is_escaped, prev_scalar, in_string, count) is the state after its LAST BYTE, and a
    multi-byte character cut by the end is the next shard's business, not an error."""
    torch = torch_mod
    from mojo_simdjson_amd.sharded import speculate_bytes

    doc = ('{"k\\\\":"v\\"x\\\\\\"y", "e":"\u00e9\u20ac\U0001F600 z", "n":[12345,true,null,"\\\\"],"s":"abc def"}' * 400).encode()
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, doc)
    assert code == 0
    d_all = torch.from_numpy(np.frombuffer(doc, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(doc) + 8, dtype=torch.int32, device=dev.device)
    unit = len(doc) // 400
    zero = dev.new_carry()
    for cut in list(range(3 * unit, 3 * unit + unit)) + [4096, 4097, 8191, 8192 + 63, 8192 + 64, 20000]:
        cout = dev.new_carry()
        dev.shard(d_all, cut, d_idx, zero, cout, is_final=False)
        c = dev.fetch(cout)
        e, ps = speculate_bytes(doc[cut - 64:cut], b" ")[1:]  # exact: no run of backslashes fills the 64 bytes
        want_n = int(np.searchsorted(idx[:n], cut))
        # in_string after `cut` bytes: parity of the unescaped quotes (the oracle counts them as structurals
        # only when they open a string, so recompute from the bytes)
        esc = False
        ins = 0
        for ch in doc[:cut]:
            if esc:
                esc = False
            elif ch == 0x5C:
                esc = True
            elif ch == 0x22:
                ins ^= 1
        assert (int(c.next_is_escaped), int(c.prev_scalar), int(c.in_string), int(c.count), int(c.utf8_error)) == \
            (e, ps, ins, want_n, 0), (cut, doc[max(0, cut - 12):cut])
        # ... and continuing from that carry gives the whole document
        if cut % 97 == 0:
            # the second shard in its own 16-byte aligned buffer, behind a 64-byte halo
            d_tail = torch.empty(64 + len(doc) - cut, dtype=torch.uint8, device=dev.device)
            d_tail.copy_(d_all[cut - 64:])
            cfin = dev.new_carry()
            dev.shard(d_tail[64:], len(doc) - cut, d_idx, cout, cfin, has_prefix=True, is_final=True,
                      trailer_len=len(doc))
            r = dev.fetch(cfin)
            assert (int(r.code), int(r.count), int(r.utf8_error)) == (0, n, 0), cut
            got = d_idx[:n].cpu().numpy().view(np.uint32).astype(np.int64)
            got[want_n:] += cut  # shard-relative offsets
            assert np.array_equal(got, idx[:n].astype(np.int64)), cut


def _build_cpp_twin():
    import subprocess

    out = os.path.join(helpers.ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "test_stage_1")
    libdir = os.path.join(helpers.ROOT, "mojo_simdjson_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(helpers.ROOT, "include"),
                           os.path.join(helpers.ROOT, "tests", "cpp", "test_stage_1.cpp"), "-o", exe,
                           "-L" + libdir, "-lmsj_stage1", "-Wl,-rpath," + libdir])
    return exe


def test_cpp_mirror_runs_the_reference_test():
    """tests/cpp/test_stage_1.cpp = the reference's tests/test_stage_1.mojo in C++, on the C++ mirror of
    DomParserImplementation (include/dom_parser_implementation.hpp) and the C ABI: same fixtures, same checks."""
    import subprocess

    exe = _build_cpp_twin()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, os.path.join(helpers.ROOT, "tests", "golden", "jsons_for_test")], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0 and "test_stage_1 ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_abi_argument_errors(torch_mod, dev):
    """What the C entry points refuse (include/msj_stage1.h): null / misaligned pointers, lengths beyond one
    uint32 segment, token pre-pass limits.  Negative codes never collide with the reference's."""
    import ctypes

    torch = torch_mod
    lib = dev.lib
    d_buf = torch.zeros(4096 + 16, dtype=torch.uint8, device=dev.device)
    d_idx = torch.zeros(4096 + 16, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)
    ok = lib.msj_stage1_device(dev.ctx, p(d_buf), 64, p(d_idx), 100, p(d_res), None, 0)
    assert ok == 0
    assert lib.msj_stage1_device(dev.ctx, p(d_buf, 1), 64, p(d_idx), 100, p(d_res), None, 0) == -1      # input not 16-byte aligned
    assert lib.msj_stage1_device(dev.ctx, p(d_buf), 64, p(d_idx, 4), 100, p(d_res), None, 0) == -1      # index buffer not 16-byte aligned
    assert lib.msj_stage1_device(dev.ctx, p(d_buf), 64, None, 100, p(d_res), None, 0) == -1             # no index buffer
    assert lib.msj_stage1_device(dev.ctx, p(d_buf), 64, p(d_idx), 100, None, None, 0) == -1             # no result
    assert lib.msj_stage1_device(None, p(d_buf), 64, p(d_idx), 100, p(d_res), None, 0) == -1            # no context
    assert lib.msj_stage1_device(dev.ctx, p(d_buf), 0, p(d_idx), 100, p(d_res), None, 0) == 13          # EMPTY, :91-92
    assert lib.msj_stage1_device(dev.ctx, p(d_buf), 1 << 32, p(d_idx), 100, p(d_res), None, 0) == 1    # CAPACITY, base.mojo:2
    n = ctypes.c_uint64(0)
    assert lib.msj_stage1(None, 5, p(d_idx), 8, ctypes.byref(n), None, 0) == -1
    assert lib.msj_stage1(b"[1]", 3, None, 8, ctypes.byref(n), None, 0) == -1
    host_idx = (ctypes.c_uint32 * 8)()
    assert lib.msj_stage1(b"[1]", 3, host_idx, 8, None, None, 0) == -1
    assert lib.msj_stage1(b"", 0, host_idx, 8, ctypes.byref(n), None, 0) == 13
    d_t = torch.zeros(64, dtype=torch.uint8, device=dev.device)
    d_d = torch.zeros(64, dtype=torch.int32, device=dev.device)
    d_r = torch.zeros(24, dtype=torch.uint8, device=dev.device)
    assert lib.msj_tokens_device(dev.ctx, p(d_buf), 64, p(d_idx), 4, p(d_t), p(d_d, 4), None, p(d_r), None) == -1   # depth not 16-byte aligned
    assert lib.msj_tokens_device(dev.ctx, p(d_buf), 64, p(d_idx), 1 << 31, p(d_t), p(d_d), None, p(d_r), None) == 1  # too many tokens
    assert lib.msj_tokens_device(dev.ctx, p(d_buf), 64, p(d_idx), 0, None, None, None, p(d_r), None) == 0            # nothing to do is fine
    assert lib.msj_token_spans_device(dev.ctx, p(d_buf), 64, p(d_idx), 4, None, None, None) == -1
    d_docs = torch.zeros(32, dtype=torch.uint8, device=dev.device)
    assert lib.msj_documents_device(dev.ctx, p(d_buf), 64, 0, p(d_idx), 4, p(d_t), p(d_d), None, None, 0, None, None) == -1        # no result
    assert lib.msj_documents_device(dev.ctx, p(d_buf), 64, 0, p(d_idx), 4, p(d_t), p(d_d), None, None, 8, p(d_docs), None) == -1  # capacity without a list
    assert lib.msj_documents_device(dev.ctx, p(d_buf), 64, 0, p(d_idx), 4, p(d_t), p(d_d, 4), None, None, 0, p(d_docs), None) == -1
    assert lib.msj_documents_device(dev.ctx, None, 64, 0, p(d_idx), 4, p(d_t), p(d_d), None, None, 0, p(d_docs), None) == -1
    assert lib.msj_documents_device(dev.ctx, None, 0, 1, None, 0, None, None, None, None, 0, p(d_docs), None) == 0                # an empty window is fine
    torch.cuda.synchronize()
