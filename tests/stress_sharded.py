#!/usr/bin/env python3
"""Randomised parity stress of the N-GPU protocol on ONE GPU (not part of pytest: runs for minutes).

Every case: a byte soup (tests/stress.py: quotes, backslash runs, control characters, multi-byte UTF-8, ...) cut at
random 16-byte aligned offsets into `world` (2..8) shards, each shard driven by its own thread, msj_ctx and
msj_sharded through the library's C entry points (msj_stage1_sharded_submit / _result) with a loopback exchange in
place of RCCL (tests/test_stage1_gpu.py::test_sharded_world8_one_gpu explains why), small uint32 segments so that shards
span several.  Code, total, every index of every shard and the trailer against the oracle on the whole stream --
whatever the speculation guessed.
usage: tests/stress_sharded.py [seconds] [seed]
"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from tests.stress import soup  # noqa: E402


def run_case(torch, L, sharded, MsjCarry, devs, data, cuts, seg_bytes):
    world = len(cuts) - 1
    total = len(data)
    d_data = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(devs[0].device)
    barrier = threading.Barrier(world)
    mine_ptrs = [None] * world
    results = [None] * world
    errors_seen = []
    lock = threading.Lock()

    def rank_main(rank):
        try:
            dev = devs[rank]
            stream = torch.cuda.Stream(device=dev.device)
            sp = ctypes.c_void_p(stream.cuda_stream)

            def allgather(comm, d_send, d_recv, nbytes, st):
                if L.msj_copy_to_host(dev.ctx, (ctypes.c_uint8 * 1)(), ctypes.c_void_p(d_send), 1, ctypes.c_void_p(st)) != 0:
                    return -3
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
            d_idx = torch.full((hi - lo + 3,), -1, dtype=torch.int32, device=dev.device)
            nseg = -(-(hi - lo) // seg_bytes)
            d_seg = torch.zeros(nseg * 32, dtype=torch.uint8, device=dev.device)
            stream.wait_stream(torch.cuda.current_stream(dev.device))  # the fills above ran on this thread's current stream
            ticket = ctypes.c_uint32()
            rc = L.msj_stage1_sharded_submit(h, ctypes.c_void_p(d_data.data_ptr() + lo), hi - lo, ctypes.c_void_p(d_idx.data_ptr()),
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
            segs = np.frombuffer(d_seg.cpu().numpy().tobytes(), dtype=np.uint64).reshape(nseg, 4)
            vals = d_idx[: cnt + 3].cpu().numpy().view(np.uint32).astype(np.int64)
            out = vals[:cnt].copy()
            assert (int(place.byte_base), int(place.count), int(place.bytes)) == (lo, cnt, hi - lo), (rank, lo, cnt)
            for base, blen, ibeg, c in segs:
                out[int(ibeg):int(ibeg) + int(c)] += int(base) + int(place.byte_base)
            results[rank] = (rcode.value, int(rtotal.value), out, vals[cnt:], int(L.msj_sharded_reruns(h)), int(place.index_begin))
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
        t.join(timeout=120)
    assert not errors_seen, errors_seen
    assert all(r is not None for r in results)
    return results


def main():
    import torch

    from mojo_simdjson_amd import sharded
    from mojo_simdjson_amd._lib import MsjCarry
    from mojo_simdjson_amd.device import Stage1Device

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    oracle = helpers.load_oracle()
    L = sharded.lib()
    L.msj_copy_to_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    L.msj_copy_to_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p]
    seg_bytes = 64 << 10
    devs = [Stage1Device(0) for _ in range(8)]
    for d in devs:
        assert L.msj_debug_set_segment_bytes(d.ctx, seg_bytes) == 0
    t0 = time.time()
    cases = reruns = nbytes = 0
    while time.time() - t0 < budget:
        world = int(rng.integers(2, 9))
        n = int(rng.integers(world * 64, 3 << 20))
        data = soup(rng, n)
        if rng.random() < 0.5:  # valid-looking stretches: speculation has something to go on, and to get wrong
            unit = b'{"key":"some text, with: punctuation","e":"true false","n":[1,2.5e3,-7,null],"s":"\\\\"q\\\\" \\\\\\\\"} '
            data = (unit * (n // len(unit) + 1))[:n]
            data = data[: n // 2] + soup(rng, n - n // 2) if rng.random() < 0.3 else data
        inner = sorted(set(int(c) // 16 * 16 for c in rng.integers(64, n - 16, world - 1)))
        cuts = [0] + [c for c in inner if 0 < c < n] + [n]
        if len(set(cuts)) != len(cuts) or len(cuts) < 3:
            continue
        idx = np.full(n + 3, helpers.SENTINEL, dtype=np.uint32)
        nn = ctypes.c_uint64(0xFFFFFFFFFFFFFFFF)
        code = oracle.msj_oracle_stage1(data, n, idx.ctypes.data, idx.size, ctypes.byref(nn))
        k = int(nn.value) if nn.value != 0xFFFFFFFFFFFFFFFF else int((idx != helpers.SENTINEL).sum())
        res = run_case(torch, L, sharded, MsjCarry, devs, data, cuts, seg_bytes)
        tag = f"case {cases} (seed {seed}, len {n}, world {len(cuts) - 1}, cuts {cuts})"
        assert all(r[0] == code for r in res), f"{tag}: codes {[r[0] for r in res]} != {code}"
        assert all(r[1] == k for r in res), f"{tag}: totals {[r[1] for r in res]} != {k}"
        merged = np.concatenate([r[2] for r in res])
        if not (merged.size == k and np.array_equal(merged, idx[:k].astype(np.int64))):
            want = idx[:k].astype(np.int64)
            m = min(merged.size, k)
            bad = int(np.argmax(merged[:m] != want[:m])) if (merged[:m] != want[:m]).any() else m
            counts = [int(r[2].size) for r in res]
            owner = int(np.searchsorted(np.cumsum(counts), bad, side="right"))
            raise AssertionError(f"{tag}: code {code}, index {bad} of {k} (rank {owner}, counts {counts}, re-runs {[r[4] for r in res]}): "
                                 f"got {merged[bad - 2:bad + 3].tolist()} want {want[bad - 2:bad + 3].tolist()}; bytes there {data[max(0, int(want[min(bad, k - 1)]) - 20):int(want[min(bad, k - 1)]) + 20]!r}")
        # the stitched offsets: every shard's first index sits at index_begin of the stream-wide array
        begins = np.concatenate([[0], np.cumsum([r[2].size for r in res])[:-1]])
        assert [r[5] for r in res] == begins.tolist(), f"{tag}: index_begin {[r[5] for r in res]} != {begins.tolist()}"
        if code in (0, 13):
            assert list(res[-1][3]) == [n, n, 0], f"{tag}: trailer {list(res[-1][3])}"
        cases += 1
        nbytes += n
        reruns += sum(r[4] for r in res)
        if cases % 25 == 0:
            print(f"{cases} cases, {nbytes / 1e6:.0f} MB, {reruns} shard re-runs, {time.time() - t0:.0f} s", flush=True)
    print(f"stress_sharded ok: seed {seed}, {cases} cases, {nbytes / 1e6:.0f} MB, {reruns} refuted speculations repaired")
    for d in devs:
        d.close()


if __name__ == "__main__":
    main()
