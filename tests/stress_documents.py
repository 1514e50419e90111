#!/usr/bin/env python3
"""Randomised stress of the rows marked "next" on the GPU box (not part of pytest: runs for minutes).

Random streams of concatenated documents (NDJSON separators, blanks or none; scalars, nested containers,
escapes, multi-byte UTF-8) through DocumentStream at random window sizes: the documents found and the
tokens of all windows must equal those of the whole stream; the token pre-pass, the bracket partners and
the token spans of the whole stream are compared with their CPU definitions (LDS path and global path).
usage: tests/stress_documents.py [seconds] [seed]
"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from tests.test_documents import _stream  # noqa: E402


def main():
    import torch

    from mojo_simdjson_amd.device import Stage1Device
    from mojo_simdjson_amd.document_stream import DocumentStream

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    oracle = helpers.load_oracle()
    dev = Stage1Device(0)
    t0 = time.time()
    cases = windows = nbytes = 0
    while time.time() - t0 < budget:
        ndocs = rng.choice([1, 3, 40, 500, 5000, 30000])
        data, starts = _stream(rng, ndocs, scalars=rng.random() < 0.5)
        widx, _ = helpers.oracle_window(oracle.msj_oracle_stage1, data)
        d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
        longest = max(b - a for a, b in zip(starts, starts[1:] + [len(data)]))
        for _ in range(3):
            window = max(64, (longest + 48) // 16 * 16, rng.choice([64, 256, 4096, 65536, 1 << 20, 1 << 26]) // 16 * 16)
            offsets, tokens = [], []
            stream = DocumentStream(dev, d_buf, len(data), window=window)
            for w in stream:
                offsets += w.document_offsets()
                tokens.append(w.d_idx.cpu().numpy().view(np.uint32).astype(np.int64) + w.base)
            assert offsets == starts, (seed, cases, window)
            assert np.array_equal(np.concatenate(tokens), widx.astype(np.int64)), (seed, cases, window)
            windows += stream.windows
        # the whole stream as one window: token pre-pass, partners, spans
        d_idx = torch.empty(len(data) + 8, dtype=torch.int32, device=dev.device)
        cin, cout = dev.new_carry(), dev.new_carry()
        dev.shard(d_buf, len(data), d_idx, cin, cout, is_final=False)
        n = int(dev.fetch(cout).count)
        assert n == widx.size
        t, d, res, m = dev.tokens(d_buf, len(data), d_idx, n, match=True)
        wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, widx)
        assert np.array_equal(t.cpu().numpy(), wt) and np.array_equal(d.cpu().numpy(), wd), (seed, cases)
        assert (res.final_depth, res.min_depth, res.max_depth) == (final, mn, mx)
        assert np.array_equal(m.cpu().numpy().view(np.uint32), helpers.oracle_match(wt)), (seed, cases)
        we, wf = helpers.oracle_token_spans(data, widx)
        for limit in ("", "0", "2048"):
            dev.lib.msj_debug_set_span_limits(dev.ctx, int(limit) if limit else 0xFFFFFFFF, 0xFFFFFFFF)
            e, f = dev.token_spans(d_buf, len(data), d_idx, n)
            assert np.array_equal(f.cpu().numpy(), wf), (seed, cases, limit)
            assert np.array_equal(e.cpu().numpy().view(np.uint32), we), (seed, cases, limit)
        dev.lib.msj_debug_set_span_limits(dev.ctx, 0xFFFFFFFF, 0xFFFFFFFF)
        cases += 1
        nbytes += len(data)
        if cases % 20 == 0:
            print(f"{cases} streams, {windows} windows, {nbytes / 1e6:.1f} MB, {time.time() - t0:.0f} s", flush=True)
    print(f"stress_documents ok: seed {seed}, {cases} streams, {windows} windows, {nbytes / 1e6:.1f} MB in {time.time() - t0:.0f} s")
    dev.close()


if __name__ == "__main__":
    main()
