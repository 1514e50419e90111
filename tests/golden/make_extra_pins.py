"""Generates tests/golden/extra_pins.json from the oracle pair.

The reference's own fixtures cover only 14 tiny valid documents (SURVEY.md
section 4); these extra vectors pin the cases it never tests.  An input is only
written when the block restatement and the byte-serial spec agree on it.
Run:  python tests/golden/make_extra_pins.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import helpers  # noqa: E402


def main():
    o = helpers.load_oracle()
    cases = []

    def add(name, data):
        cases.append((name, bytes(data)))

    add("empty_ws", b"  \n\t ")
    add("unclosed", b'{"a":"bc')
    add("unescaped_nl", b'["a\nb"]')
    add("unescaped_nul", b'["a\x00b"]')
    add("unclosed_beats_unescaped", b'["a\x01b')
    add("all_open", b"[" * 200)
    add("formfeed_sub", b"\x0c\x1a 1")
    add("ctrl_outside", b"\x00\x01[1]")
    add("bad_utf8_ff", b'["\xff\xfe"]')
    add("overlong_c0", b'["\xc0\x80"]')
    add("surrogate", b'["\xed\xa0\x80"]')
    add("too_big", b'["\xf4\x90\x80\x80"]')
    add("trunc_utf8_eof", b'"ab\xe4\xb8')
    add("good_utf8", '["é中😀"]'.encode())
    for n in (63, 64, 65, 127, 128, 129, 255, 256, 257):
        body = (b'{"k":[1,2,"x y",true,null,-3.5e2],"s":"a\\"b\\\\"} ' * 8)[: n - 1]
        add(f"len_{n}", body + b"7")
        add(f"len_{n}_spaces", b" " * (n - 1) + b"1")
    for off in (61, 62, 63, 125, 126, 127):
        # backslash right at a block / step boundary, followed by a quote
        s = bytearray(b'"' + b"a" * 200 + b'"')
        s[off] = 0x5C
        s[off + 1] = 0x22
        add(f"bs_at_{off}", s)
        s2 = bytearray(s)
        s2[off - 1] = 0x5C  # two backslashes: the quote is real again
        add(f"bs2_at_{off}", s2)
    add("bs_run_64", b'"' + b"\\" * 64 + b'"')
    add("bs_run_65", b'"' + b"\\" * 65 + b'"')
    add("bs_run_128", b'"' + b"\\" * 128 + b'" ')
    add("bs_run_129", b'"' + b"\\" * 129 + b'"x"')
    add("quote_after_scalar", b'[1"a"]')
    add("scalar_after_quote", b'"a"b')

    pins = []
    for name, d in cases:
        a = helpers.run_oracle(o.msj_oracle_stage1, d)
        b = helpers.run_oracle(o.msj_oracle_stage1_serial, d)
        assert a[0] == b[0] and a[1] == b[1], name
        if a[1] is not None:
            assert list(a[2]) == list(b[2]), name
        pins.append({
            "name": name,
            "input_hex": d.hex(),
            "code": a[0],
            "indices": None if a[1] is None else [int(v) for v in a[2][: a[1]]],
            "utf8": o.msj_oracle_utf8(d, len(d)),
        })
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "extra_pins.json")
    with open(out, "w") as f:
        json.dump(pins, f, indent=0)
    print(f"wrote {len(pins)} pins to {out}")


if __name__ == "__main__":
    main()
