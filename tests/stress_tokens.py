#!/usr/bin/env python3
"""Randomised parity stress of the f rows on the GPU box (not part of pytest: runs for minutes).

Byte soups (tests/stress.py's alphabets plus number- and string-heavy ones) through stage 1, then
msj_stage2_prep_device with bracket matching, the pairs form, msj_tokens_device and msj_token_spans_device, each compared with
the definitions in oracle/tokens_oracle.c token by token: type, depth, final / min / max depth, partner, span end,
span flags; every fourth case also in two or three chained pieces (the depth carried from call to call).  Every third case runs with a lowered MSJ_SPANS_LDS_LIMIT-like stretch length (long strings) so that the
per-token path from global memory is taken too.
usage: tests/stress_tokens.py [seconds] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from tests.stress import ALPHABETS, soup  # noqa: E402

EXTRA = [
    b'0123456789-+.eE,: []',
    b'"\\\\\\\\\\" ,:x',
    b'{"a":1.5e3,"b":"c\\n"} \n',
    b'"' + b"y" * 40 + b'\\',
    b'truefalsn,[]{}:" 1',
]


def main():
    import torch

    from mojo_simdjson_amd import _lib
    from mojo_simdjson_amd.device import Stage1Device

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ALPHABETS.extend(EXTRA)
    dev = Stage1Device(0)
    sizes = [1, 2, 63, 64, 65, 511, 512, 513, 4095, 4096, 4097, 12288, 12289, 40000, 1 << 18, (1 << 20) + 77, (1 << 22) + 5]
    t0 = time.time()
    cases = tokens = 0
    while time.time() - t0 < budget:
        n = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(1, 1 << 20))
        data = soup(rng, n)
        if cases % 3 == 2:  # a long string in the middle: its workgroup's stretch does not fit LDS
            cut = int(rng.integers(0, n))
            data = data[:cut] + b' "' + b"s" * int(rng.integers(12000, 60000)) + b'" ' + data[cut:]
            n = len(data)
        d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
        d_idx = torch.empty(n + 3 + 4, dtype=torch.int32, device=dev.device)
        d_res = dev.new_carry()
        dev.index(d_buf, d_idx, d_res)
        k = int(dev.fetch(d_res).count)
        idx = d_idx[:k].cpu().numpy().view(np.uint32)
        tag = f"case {cases} (seed {seed}, len {n}, {k} tokens)"
        wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx)
        we, wf = helpers.oracle_token_spans(data, idx)
        t, d, res, m, e, f = dev.stage2_prep(d_buf, n, d_idx, k, match=True)
        assert np.array_equal(t.cpu().numpy(), wt), tag + ": type"
        assert np.array_equal(d.cpu().numpy(), wd), tag + ": depth"
        assert (res.n, res.final_depth, res.min_depth, res.max_depth) == (k, final, mn, mx), tag + ": result"
        assert np.array_equal(m.cpu().numpy().view(np.uint32), helpers.oracle_match(wt)), tag + ": match"
        got_f, got_e = f.cpu().numpy(), e.cpu().numpy().view(np.uint32)
        if not (np.array_equal(got_f, wf) and np.array_equal(got_e, we)):
            bad = int(np.argmax((got_f != wf) | (got_e != we)))
            raise AssertionError(f"{tag}: token {bad} at {idx[bad]}: end {got_e[bad]} / flags {got_f[bad]} != {we[bad]} / {wf[bad]}: "
                                 f"{data[idx[bad]:idx[bad] + 40]!r}")
        # the pairs form (round 5): one {open, close} record per container = the definition's match[] read at the opening
        # brackets, through the fused call and through msj_tokens_pairs_device in turn
        opens = np.nonzero((wt == ord("{")) | (wt == ord("[")))[0]
        wm = helpers.oracle_match(wt)
        tp_, dp_, pr, _, _, d_tr = dev.stage2_prep_pairs(d_buf, n, d_idx, k, spans=cases % 2 == 0)
        rr = _lib.MsjTokensResult.from_buffer_copy(d_tr.cpu().numpy().tobytes())
        assert rr.reserved == len(opens) and (rr.final_depth, rr.min_depth, rr.max_depth) == ((final, mn, mx) if k else (rr.final_depth, rr.min_depth, rr.max_depth)), tag + ": pairs result"
        got_p = pr[:len(opens)].cpu().numpy().view(np.uint32)
        want_p = np.stack([opens.astype(np.uint32), wm[opens]], axis=1) if len(opens) else np.zeros((0, 2), dtype=np.uint32)
        assert np.array_equal(got_p, want_p), tag + ": pairs"
        assert np.array_equal(tp_.cpu().numpy(), wt) and np.array_equal(dp_.cpu().numpy(), wd), tag + ": pairs call type / depth"
        t2, d2, res2 = dev.tokens(d_buf, n, d_idx, k)[:3]
        assert np.array_equal(t2.cpu().numpy(), wt) and np.array_equal(d2.cpu().numpy(), wd), tag + ": msj_tokens_device"
        e2, f2 = dev.token_spans(d_buf, n, d_idx, k)
        assert np.array_equal(f2.cpu().numpy(), wf) and np.array_equal(e2.cpu().numpy().view(np.uint32), we), tag + ": msj_token_spans_device"
        if cases % 4 == 1 and k >= 8:
            # the same tokens handed over in two or three pieces (msj_stage2_prep_chain_device / msj_tokens_chain_device):
            # depths and final / min / max of the whole, partners of each piece alone
            cuts = sorted(set([0, k] + [int(c) // 4 * 4 for c in rng.integers(1, k, int(rng.integers(1, 3)))]))
            prev = None
            for a, b in zip(cuts[:-1], cuts[1:]):
                buf = torch.zeros(24, dtype=torch.uint8, device=dev.device)
                if cases % 8 == 1:
                    tp, dp, _, mp, _, _ = dev.stage2_prep(d_buf, n, d_idx[a:], b - a, match=True, d_prev=prev, d_result=buf)
                else:
                    tp, dp, _, mp = dev.tokens(d_buf, n, d_idx[a:], b - a, match=True, d_result=buf, sync=False, d_prev=prev)
                assert np.array_equal(tp.cpu().numpy(), wt[a:b]) and np.array_equal(dp.cpu().numpy(), wd[a:b]), f"{tag}: chained [{a}, {b})"
                assert np.array_equal(mp.cpu().numpy().view(np.uint32), helpers.oracle_match(wt[a:b])), f"{tag}: chained partners [{a}, {b})"
                # ... and as the pairs form: the piece's own containers, the depth carried
                _, dq, pq, _, _, bq = dev.stage2_prep_pairs(d_buf, n, d_idx[a:], b - a, spans=False, d_prev=prev)
                op = np.nonzero((wt[a:b] == ord("{")) | (wt[a:b] == ord("[")))[0]
                wmp = helpers.oracle_match(wt[a:b])
                wantq = np.stack([op.astype(np.uint32), wmp[op]], axis=1) if len(op) else np.zeros((0, 2), dtype=np.uint32)
                assert np.array_equal(pq[:len(op)].cpu().numpy().view(np.uint32), wantq), f"{tag}: chained pairs [{a}, {b})"
                assert np.array_equal(dq.cpu().numpy(), wd[a:b]), f"{tag}: chained pairs depth [{a}, {b})"
                prev = buf
            r = _lib.MsjTokensResult.from_buffer_copy(prev.cpu().numpy().tobytes())
            assert (r.final_depth, r.min_depth, r.max_depth) == (final, mn, mx), f"{tag}: chained result {cuts}"
        cases += 1
        tokens += k
        if cases % 50 == 0:
            print(f"{cases} cases, {tokens / 1e6:.1f} M tokens, {time.time() - t0:.0f} s", flush=True)
    print(f"stress_tokens ok: seed {seed}, {cases} cases, {tokens / 1e6:.1f} M tokens compared token by token")
    dev.close()


if __name__ == "__main__":
    main()
