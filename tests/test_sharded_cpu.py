"""CPU tests of the multi-GPU stitch logic (mojo_simdjson_amd/sharded.py):
pure-host carry resolution, and the torch.distributed exchange on `gloo` with
world_size 2 (the GPU box uses the same code over RCCL)."""
import os
import random
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mojo_simdjson_amd import sharded
from tests import helpers


def serial_state(data):
    """(next_is_escaped, in_string, prev_scalar) after `data` per the serial spec."""
    esc = instr = pnq = 0
    for c in data:
        escaped = esc
        if escaped:
            esc = 0
        elif c == 0x5C:
            esc = 1
        quote = (c == 0x22) and not escaped
        instr ^= int(quote)
        scalar = c not in sharded._NONSCALAR
        pnq = int(scalar and not quote)
    return esc, instr, pnq


def test_boundary_carry_matches_serial():
    rng = random.Random(2)
    alpha = b'\\\\\\\\""a1 ,:[]{}\n'
    for _ in range(3000):
        nsh = rng.randint(1, 4)
        shards = [bytes(rng.choice(alpha) for _ in range(rng.randint(0, 9))) for _ in range(nsh)]
        cap = rng.choice([1, 2, 3, 100])
        tails = [s[max(0, len(s) - cap):] for s in shards]
        complete = [len(t) == len(s) for t, s in zip(tails, shards)]
        got = sharded.boundary_carry(tails, complete)
        e, _, ps = serial_state(b"".join(shards))
        if got is not None:
            assert got == (e, ps), (shards, cap)
        else:
            assert cap < max(len(s) for s in shards)  # only truncated tails may be undecided
    # full tails always decide
    for _ in range(500):
        shards = [bytes(rng.choice(alpha) for _ in range(rng.randint(0, 9))) for _ in range(3)]
        e, _, ps = serial_state(b"".join(shards))
        assert sharded.boundary_carry(shards, [True] * 3) == (e, ps)


def test_parity_prefix_and_code():
    s, last = sharded.parity_prefix([1, 0, 1, 1])
    assert s == [0, 1, 1, 0] and last == 1
    assert sharded.global_code(1, True, 5, True, False, True) == 15
    assert sharded.global_code(0, True, 5, True, False, True) == 14
    assert sharded.global_code(0, False, 0, True, False, True) == 13
    assert sharded.global_code(0, False, 3, True, False, True) == 11
    assert sharded.global_code(0, False, 3, True, False, False) == 0
    assert sharded.global_code(0, False, 3, False, True, False) == 24


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, stream_hex, cuts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = bytes.fromhex(stream_hex)
        lo, hi = cuts[rank], cuts[rank + 1]
        shard = data[lo:hi]
        dev = torch.device("cpu")
        cap = 4
        while True:  # same retry loop as ShardedStage1.boundary_exchange
            tails, lens = sharded.exchange_tails(shard[max(0, len(shard) - cap):], len(shard), dev,
                                                 tail_cap=cap)
            res = sharded.resolve_boundaries(tails, lens)
            if all(r is not None for r in res):
                break
            cap *= 16
        e_in, ps_in = res[rank]
        # parity of my shard given my exact escape carry (stand-in for the summary pass)
        esc, par = e_in, 0
        for c in shard:
            escaped = esc
            if escaped:
                esc = 0
            elif c == 0x5C:
                esc = 1
            par ^= int(c == 0x22 and not escaped)
        got = sharded.exchange_words([par], dev)
        s_list, _ = sharded.parity_prefix([int(w[0]) for w in got])
        q.put((rank, e_in, ps_in, s_list[rank]))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_exchange():
    rng = random.Random(8)
    alpha = b'\\\\\\""a1 ,:[]{}'
    for trial in range(3):
        data = bytes(rng.choice(alpha) for _ in range(200))
        if trial == 2:  # long backslash run across the cut: forces the tail-growth retry
            data = b'["' + b"\\" * 150 + b'\\"x"' + b",1]" * 10
        cut = rng.randint(20, len(data) - 20) if trial != 2 else 120
        cuts = [0, cut, len(data)]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, data.hex(), cuts, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert res[0][1:] == (0, 0, 0)
        e, s, ps = serial_state(data[:cut])
        assert res[1][1:] == (e, ps, s), (data, cut)


def test_guess_and_verify_chain():
    """The speculation is only a guess; verify_chain must accept exactly the right ones."""
    rng = random.Random(4)
    doc = (b'{"a":"x y","b":[1,2,"q:\\"z\\"",true],"c":"k:"}' * 40)
    right = wrong = 0
    for _ in range(400):
        cut = rng.randint(70, len(doc) - 70)
        halo, head = doc[cut - 64:cut], doc[cut:cut + 4096]
        e, s, ps = serial_state(doc[:cut])
        hc = sharded.halo_carry(halo)
        assert hc == (e, ps)
        g = sharded.guess_in_string(halo, head, e)
        right += g == s
        wrong += g != s
        # rank 1 used guess g: the chain check must flag it iff g != s
        e1, s1, ps1 = serial_state(doc)
        e0, s0, ps0 = serial_state(doc[:cut])
        rep = [dict(s_used=0, e_used=0, ps_used=0, s_out=s0, e_out=e0, ps_out=ps0),
               dict(s_used=g, e_used=e, ps_used=ps, s_out=s1 ^ s ^ g, e_out=e1, ps_out=ps1)]
        first_wrong, true_in = sharded.verify_chain(rep)
        assert (first_wrong == 2) == (g == s)
        assert true_in[1] == (s, e, ps)
    assert right > 300 and wrong > 0  # good but not perfect: the re-run path matters
