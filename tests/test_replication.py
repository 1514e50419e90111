"""CPU tests of the replication checker (tests/replication.py) that bench.py's verification step and the
full-size GPU tests rely on: a checker that cannot fail proves nothing."""
import numpy as np
import torch

from tests import replication as rep


def _stream(unit_idx, unit_len, reps):
    return np.concatenate([unit_idx + k * unit_len for k in range(reps)]).astype(np.int64)


def test_stream_hash_closed_form_equals_brute_force():
    rng = np.random.default_rng(3)
    for unit_len, n, reps in ((1000, 37, 1), (70_000, 9_000, 5), (67_105_101, 4_000, 1024), (5_000_000_000, 2000, 3)):
        u = np.sort(rng.choice(unit_len, n, replace=False)).astype(np.int64)
        g = _stream(u, unit_len, reps)
        want = 0
        for j0 in range(0, g.size, 1 << 20):  # python ints: no wrap-around to trust
            part = g[j0:j0 + (1 << 20)].astype(object)
            jj = np.arange(j0, j0 + part.size).astype(object)
            want += int(((part + 1) * (2 * jj + 1)).sum())
        assert rep.stream_hash(u, unit_len, reps) == want & rep.MASK64


def test_check_shard_finds_every_kind_of_error():
    rng = np.random.default_rng(4)
    unit_len, n, reps = 50_000, 6_000, 7
    u = np.sort(rng.choice(unit_len, n, replace=False)).astype(np.int64)
    g = _stream(u, unit_len, reps)
    d_u = torch.from_numpy(u)
    total = unit_len * reps
    seg = 65536  # small "uint32 segments"
    cuts = [0, 12_345 // 16 * 16, 170_000 // 16 * 16, 170_016, total]
    hashes = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        ib = rep.expected_index_begin(u, unit_len, lo)
        ie = rep.expected_index_begin(u, unit_len, hi)
        assert (ib, ie) == (int(np.searchsorted(g, lo)), int(np.searchsorted(g, hi)))
        local = g[ib:ie] - lo
        segs = []
        for s in range(0, hi - lo, seg):
            a, b = int(np.searchsorted(local, s)), int(np.searchsorted(local, min(hi - lo, s + seg)))
            segs.append((s, a, b - a))
        rel = local - (local // seg) * seg
        d_idx = torch.from_numpy(rel.astype(np.int64).astype(np.uint32).view(np.int32))
        bad, h = rep.check_shard(torch, d_idx, ie - ib, d_u, unit_len, lo, ib, segs, chunk=1000)
        assert bad == 0
        hashes.append(h)
        if ie - ib > 10:
            # one wrong value, a wrong index_begin, a wrong byte base, a wrong segment table: all are seen
            d_bad = d_idx.clone()
            d_bad[5] += 1
            assert rep.check_shard(torch, d_bad, ie - ib, d_u, unit_len, lo, ib, segs)[0] == 1
            assert rep.check_shard(torch, d_idx, ie - ib, d_u, unit_len, lo, ib + 1, segs)[0] > 0
            assert rep.check_shard(torch, d_idx, ie - ib, d_u, unit_len, lo + 16, ib, segs)[0] == ie - ib
            if len(segs) > 1:
                wrong = [segs[0], (segs[1][0], segs[1][1] + 1, segs[1][2])] + segs[2:]
                assert rep.check_shard(torch, d_idx, ie - ib, d_u, unit_len, lo, ib, wrong)[0] > 0
    assert sum(hashes) & rep.MASK64 == rep.stream_hash(u, unit_len, reps)
