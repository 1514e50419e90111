"""CPU tests that PIN the oracle (oracle/stage1_oracle.c) to the reference.

The reference (pure Mojo) cannot run here, so its own golden fixtures -- the 14
files of tests/jsons_for_test/valid/ plus the two harness meta-fixtures, copied
as data into tests/golden/jsons_for_test/ -- are the anchor, exactly as
tests/test_stage_1.mojo uses them.
"""
import json
import os

import numpy as np
import pytest

from tests import helpers

STRUCTURAL_CHARS = b'{}[]:,tfn-0123456789"'  # tests/test_stage_1.mojo:35


def check_stage1(fn, path):
    """tests/test_stage_1.mojo:85-96 check_stage1 + :43-82 verify_expected_..."""
    js, mask = helpers.read_fixture(path)
    code, n, idx = helpers.run_oracle(fn, js)
    assert code == 0, "unexpected error code"                       # :93
    for i in range(min(len(js), len(mask))):                        # :28-40
        if mask[i:i + 1] == b"1" and js[i] not in STRUCTURAL_CHARS:
            raise AssertionError(f"Wrong tagging of characters, {chr(js[i])} is not a structural character")
    assert all(idx[i - 1] < idx[i] for i in range(1, n))            # :23-25
    detected = helpers.mask_from_indices(idx[:n], len(mask))        # :52-58
    if detected != mask:
        raise AssertionError("Detected and expected structural characters do not match")
    assert idx[n] == len(js) and idx[n + 1] == len(js) and idx[n + 2] == 0  # :70-82


@pytest.mark.parametrize("which", ["msj_oracle_stage1", "msj_oracle_stage1_serial"])
def test_simple_json(oracle, which):
    """tests/test_stage_1.mojo:113-122 test_simple_json: every file of valid/."""
    files = helpers.golden_valid_files()
    assert len(files) > 5 and len(files) == 14
    for f in files:
        check_stage1(getattr(oracle, which), f)


def test_wrong_tagging(oracle):
    """tests/test_stage_1.mojo:99-102."""
    with pytest.raises(AssertionError, match="l is not a structural character"):
        check_stage1(oracle.msj_oracle_stage1, os.path.join(helpers.GOLDEN, "wrong_tagging.json"))


def test_detect_incorrect_result(oracle):
    """tests/test_stage_1.mojo:105-110."""
    with pytest.raises(AssertionError, match="Detected and expected structural characters do not match"):
        check_stage1(oracle.msj_oracle_stage1, os.path.join(helpers.GOLDEN, "detect_incorrect_result.json"))


def test_survey_decoded_vectors(oracle):
    """The index lists SURVEY.md section 4 decodes from the fixtures."""
    exp = {
        "simple_json.json": [0, 1, 2, 4, 5],
        "simple_floats.json": [0, 2, 6, 8, 16],
        "simple_strings.json": [0, 2, 12, 14, 24],
        "escaping.json": [0, 2, 13, 15, 17, 20, 21, 28, 30, 34, 36, 42, 44, 50, 51, 53, 56, 57, 63],
        "escaping_very_long.json": [0, 9, 20, 48, 50, 53, 54, 61, 63, 67, 76, 82, 94, 100, 101,
                                    119, 123, 125, 131],
    }
    for name, want in exp.items():
        js, _ = helpers.read_fixture(os.path.join(helpers.GOLDEN, "valid", name))
        code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, js)
        assert code == 0 and list(idx[:n]) == want


def test_classify_both_shuffle_readings(oracle):
    """haswell.mojo:22-74: index-mod-32 and pshufb readings of _dynamic_shuffle
    agree with the plain sets on all 256 byte values."""
    ws = {0x09, 0x0A, 0x0D, 0x20}
    op = {0x0C, 0x1A, 0x2C, 0x3A, 0x5B, 0x5D, 0x7B, 0x7D}
    for b in range(256):
        want = (1 if b in ws else 0) | (2 if b in op else 0)
        assert oracle.msj_oracle_classify_byte(b, 0) == want
        assert oracle.msj_oracle_classify_byte(b, 1) == want


def test_block_equals_serial_fuzz(oracle):
    for d in helpers.fuzz_inputs(1234, 6000):
        a = helpers.run_oracle(oracle.msj_oracle_stage1, d)
        b = helpers.run_oracle(oracle.msj_oracle_stage1_serial, d)
        assert a[0] == b[0], d
        assert a[1] == b[1]
        if a[1] is not None:
            assert np.array_equal(a[2], b[2]), d


def test_error_codes_and_quirks(oracle):
    run = lambda d: helpers.run_oracle(oracle.msj_oracle_stage1, d)
    assert run(b"")[0] == 13                       # json_structural_indexer.mojo:91-92
    assert run(b"   ")[0] == 13                    # :176-177
    assert run(b'"abc')[0] == 15                   # :151-155
    assert run(b'"a\nb"')[0] == 14                 # :157-158
    assert run(b'"a\x00b')[0] == 15                # 15 wins over 14 (order of finish())
    assert run(b'"abc')[1] is None                 # n / trailer untouched on early return
    assert run(b"\xff\xfe")[0] == 0                # utf8 checker is a stub (:16-30)
    c, n, idx = run(b"[[[[")                       # H5: n == len, trailer at len..len+2
    assert (c, n, list(idx)) == (0, 4, [0, 1, 2, 3, 4, 4, 0])
    assert list(run(b'[1"a"]')[2][:3]) == [0, 1, 5]    # quote after scalar
    assert list(run(b'"a"b')[2][:2]) == [0, 3]         # scalar after closing quote
    assert list(run(b"\x0c\x1a")[2][:2]) == [0, 1]     # 0x0C / 0x1A classify as operators
    assert run(b"[1,2]" + b" " * 123)[1] == 5          # len exactly 128
    # capacity: the trailer needs len + 3 slots
    import ctypes
    buf = np.zeros(6, dtype=np.uint32)
    n = ctypes.c_uint64(0)
    assert oracle.msj_oracle_stage1(b"[[[[", 4, buf.ctypes.data, 6, ctypes.byref(n)) == 1


def test_utf8_oracle_matches_cpython(oracle):
    import random

    rng = random.Random(5)
    vals = [0x41, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC1, 0xC2, 0xDF, 0xE0, 0xE1,
            0xEC, 0xED, 0xEE, 0xEF, 0xF0, 0xF1, 0xF3, 0xF4, 0xF5, 0xFF]
    for _ in range(20000):
        d = bytes(rng.choice(vals) for _ in range(rng.randint(0, 10)))
        try:
            d.decode("utf-8")
            want = 0
        except UnicodeDecodeError:
            want = 11
        assert oracle.msj_oracle_utf8(d, len(d)) == want, d


def test_extra_pins(oracle):
    """tests/golden/extra_pins.json: vectors the reference has none of (error
    codes, block-boundary lengths, backslash at bit 62/63/126/127, odd control
    characters, all-'[').  Generated by tests/golden/make_extra_pins.py from the
    oracle pair; re-checked here so a later oracle edit cannot drift silently."""
    with open(os.path.join(helpers.ROOT, "tests", "golden", "extra_pins.json")) as f:
        pins = json.load(f)
    assert len(pins) > 40
    for p in pins:
        d = bytes.fromhex(p["input_hex"])
        for fn in (oracle.msj_oracle_stage1, oracle.msj_oracle_stage1_serial):
            code, n, idx = helpers.run_oracle(fn, d)
            assert code == p["code"], p["name"]
            if p["indices"] is not None:
                assert list(idx[:n]) == p["indices"], p["name"]
        assert oracle.msj_oracle_utf8(d, len(d)) == p["utf8"], p["name"]
