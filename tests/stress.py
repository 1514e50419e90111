#!/usr/bin/env python3
"""Randomised parity stress on the GPU box (not part of pytest: runs for minutes).

Byte soups over an alphabet chosen to hit every carry (quotes, long backslash runs, scalars,
operators, control characters, multi-byte UTF-8 incl. invalid sequences) at sizes from a few
bytes to tens of MiB, with density changing along the stream; every index, the count, the
trailer, the return code and the strict UTF-8 verdict are compared with the oracle.
usage: tests/stress.py [seconds] [seed]      (MSJ_STRESS_FLAGS=0x100: through the two-pass fallback kernels)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from mojo_simdjson_amd.dom_parser_implementation import DomParserImplementation  # noqa: E402

ALPHABETS = [
    b'{}[],: \n"\\ab1',
    b'""\\\\\\\\ a,',
    b'[[[[]]]],,,,1234',
    b'"abcdefghijklmnop\\"',
    b'{"key":"v\xc3\xa9\xe2\x82\xac\xf0\x9f\x98\x80",\x01\t}',
    b'\x80\xc0\xe0\xed\xa0\xf4\x90\xf8" a',
    b'          \n\t\r1',
    b'\x0c\x1a:,{}x"',
]


def soup(rng, n):
    out = []
    left = n
    while left > 0:
        a = np.frombuffer(ALPHABETS[rng.integers(len(ALPHABETS))], dtype=np.uint8)
        seg = int(min(left, rng.choice([1, 7, 63, 64, 65, 4095, 4096, 4097, 32768, 100000, 1 << 20])))
        w = rng.random(len(a)) ** rng.choice([1, 3, 8])  # skewed weights: sparse and dense stretches
        out.append(a[rng.choice(len(a), size=seg, p=w / w.sum())])
        left -= seg
    return np.concatenate(out)[:n].tobytes()


def main():
    import ctypes

    import torch

    from mojo_simdjson_amd.device import Stage1Device

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    flags = int(os.environ.get("MSJ_STRESS_FLAGS", "0"), 0)  # e.g. 0x100 = MSJ_FLAG_TWO_PASS: the fallback kernels
    rng = np.random.default_rng(seed)
    oracle = helpers.load_oracle()
    dev = Stage1Device(0)
    sizes = [1, 2, 63, 64, 65, 127, 4095, 4096, 4097, 8 * 4096 - 1, 8 * 4096, 8 * 4096 + 1, 32 * 4096 + 5,
             1 << 20, (1 << 22) + 77, (1 << 24) + 4099, (1 << 25) + 1]
    t0 = time.time()
    cases = nbytes = errs = 0
    while time.time() - t0 < budget:
        n = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(1, 1 << 22))
        data = soup(rng, n)
        # oracle: code, and every index it wrote (on codes 14 / 15 the reference has written all
        # indices but neither n nor the trailer)
        idx = np.full(n + 3, helpers.SENTINEL, dtype=np.uint32)
        nn = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
        code = oracle.msj_oracle_stage1(data, n, idx.ctypes.data, idx.size, ctypes.byref(nn))
        k = int(nn.value) if nn.value != 0xFFFFFFFFFFFFFFFF else int((idx != helpers.SENTINEL).sum())
        d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
        d_idx = torch.full((n + 3 + 4,), -1, dtype=torch.int32, device=dev.device)
        d_res = dev.new_carry()
        dev.index(d_buf, d_idx, d_res, flags=flags)
        r = dev.fetch(d_res)
        tag = f"case {cases} (seed {seed}, len {n})"
        assert r.internal_error == 0, tag
        assert int(r.code) == code, f"{tag}: code {int(r.code)} != {code}"
        assert int(r.count) == k, f"{tag}: count {int(r.count)} != {k}"
        got = d_idx[:k].cpu().numpy().view(np.uint32)
        if not np.array_equal(got, idx[:k]):
            bad = int(np.argmax(got != idx[:k]))
            raise AssertionError(f"{tag}: index {bad}: {got[bad]} != {idx[bad]}")
        if code in (0, 13):
            assert d_idx[k:k + 3].cpu().numpy().view(np.uint32).tolist() == [n, n, 0], f"{tag}: trailer"
        assert (11 if r.utf8_error else 0) == oracle.msj_oracle_utf8(data, n), f"{tag}: utf8"
        cases += 1
        nbytes += n
        errs += code != 0
        if cases % 50 == 0:
            print(f"{cases} cases ({errs} with an error code), {nbytes / 1e6:.0f} MB, {time.time() - t0:.0f} s", flush=True)
    print(f"stress ok: {cases} cases ({errs} with an error code), {nbytes / 1e6:.0f} MB compared bit for bit (seed {seed})")
    dev.close()


if __name__ == "__main__":
    main()
