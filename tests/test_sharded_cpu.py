"""CPU tests of the N-GPU path's host protocol (mojo_simdjson_amd/csrc/sharded.cpp through
mojo_simdjson_amd/sharded.py): the pure functions against the serial spec, and the LIVE protocol --
the library's own msj_stage1_sharded_submit / _result, re-run loop included -- with world_size 2 on
`gloo`, a shard runner that follows the serial spec standing in for the kernel (msj_sharded_ops) and
host memory standing in for the device.  The GPU box runs the same C code over the HIP kernels and
RCCL."""
import ctypes
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mojo_simdjson_amd import sharded
from mojo_simdjson_amd._lib import MsjCarry

_NONSCALAR = frozenset([0x20, 0x09, 0x0A, 0x0D, 0x0C, 0x1A, 0x2C, 0x3A, 0x5B, 0x5D, 0x7B, 0x7D])


def serial_run(data, esc=0, instr=0, pnq=0, base=0):
    """The serial spec from a given state: (indices, bad, esc, instr, pnq)."""
    idx, bad = [], 0
    for i, c in enumerate(data):
        escaped = esc
        if escaped:
            esc = 0
        elif c == 0x5C:
            esc = 1
        quote = (c == 0x22) and not escaped
        before = instr
        instr ^= int(quote)
        scalar = c not in _NONSCALAR
        op = c in (0x0C, 0x1A, 0x2C, 0x3A, 0x5B, 0x5D, 0x7B, 0x7D)
        if (op or (scalar and not pnq)) and not before:
            idx.append(base + i)
        bad |= int(c <= 0x1F and instr)
        pnq = int(scalar and not quote)
    return idx, bad, esc, instr, pnq


def serial_state(data):
    _, _, esc, instr, pnq = serial_run(data)
    return esc, instr, pnq


def test_speculate_matches_serial():
    """next_is_escaped / prev_scalar from the 64-byte halo are exact (unless a backslash run fills it); the
    in_string guess is right most of the time and never trusted."""
    rng = random.Random(4)
    doc = (b'{"a":"x y","b":[1,2,"q:\\"z\\"",true],"c":"k:","d":"\\\\\\\\"}' * 40)
    right = wrong = 0
    for _ in range(600):
        cut = rng.randint(70, len(doc) - 70)
        halo, head = doc[cut - 64:cut], doc[cut:cut + 4096]
        e, s, ps = serial_state(doc[:cut])
        g, ge, gps = sharded.speculate_bytes(halo, head)
        assert (ge, gps) == (e, ps), (cut, halo)
        right += g == s
        wrong += g != s
    assert wrong <= 3, (right, wrong)  # text and keys contradict the wrong hypothesis within a few bytes
    # ... unless every string is made of characters a document also holds outside of strings: then the first quote's
    # neighbours decide, which is good but not perfect -- the re-run path matters
    doc = (b'{"a":"e l","s":[1,2,"e:1,",true],"u":"e:","f":", 1"}' * 40)
    right = wrong = 0
    for _ in range(600):
        cut = rng.randint(70, len(doc) - 70)
        halo, head = doc[cut - 64:cut], doc[cut:cut + 4096]
        e, s, ps = serial_state(doc[:cut])
        g, ge, gps = sharded.speculate_bytes(halo, head)
        assert (ge, gps) == (e, ps), (cut, halo)
        right += g == s
        wrong += g != s
    assert right > 450 and wrong > 0, (right, wrong)
    # short halos (shard near the start of the stream), no halo at all, undecidable halo
    assert sharded.speculate_bytes(b"", b'"abc') == (0, 0, 0)
    assert sharded.speculate_bytes(b'["a', b'b",1]')[1:] == (0, 1)
    assert sharded.speculate_bytes(b"\\" * 64, b'"x')[1:] == (0, 1)  # any guess: the chain check settles it
    # msj_shard_speculate_ex says whether a hypothesis was contradicted (else the caller may look at more bytes)
    L = sharded.lib()
    L.msj_shard_speculate_ex.restype = ctypes.c_int32
    L.msj_shard_speculate_ex.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(MsjCarry),
                                         ctypes.POINTER(ctypes.c_int32)]
    for head, want in ((b"1 2 3 4 5", 0), (b"1 2 3 x", 1), (b'abc",1]', 1), (b"\n 12", 1), (b"", 0)):
        c, d = MsjCarry(), ctypes.c_int32(-1)
        assert L.msj_shard_speculate_ex(b'{"k":"' + b"1" * 58, 64, head, len(head), ctypes.byref(c), ctypes.byref(d)) == 0
        assert d.value == want, (head, d.value)
    c, d = MsjCarry(), ctypes.c_int32(-1)
    assert L.msj_shard_speculate_ex(b"", 0, b"xyz", 3, ctypes.byref(c), ctypes.byref(d)) == 0 and d.value == 1  # start of the stream
    rng = random.Random(9)
    alpha = b'\\\\\\""a1 ,:[]{}\n'
    for _ in range(2000):
        n = rng.randint(1, 64)
        pre = bytes(rng.choice(alpha) for _ in range(200))
        e, _, ps = serial_state(pre)
        halo = pre[-n:]
        if all(c == 0x5C for c in halo) or (halo[-1] == 0x22 and all(c == 0x5C for c in halo[:-1])):
            continue  # the run reaches the halo's first byte: undecidable from these bytes
        assert sharded.speculate_bytes(halo, b"x")[1:] == (e, ps), halo


def test_verify_chain():
    """msj_shard_verify: replays the chain through wrong in_string guesses, stops at wrong escape carries or a
    poisoned launch, and names every rank that has to index again."""
    rng = random.Random(6)
    doc = (b'{"a":"x y","b":[1,2,"q:\\"z\\"",true],"c":"k:","e":"\\\\"}' * 30)
    for _ in range(300):
        world = rng.randint(2, 8)
        cuts = [0] + sorted(rng.sample(range(10, len(doc) - 10), world - 1)) + [len(doc)]
        true_in = [serial_state(doc[:c]) for c in cuts[:-1]]  # (e, s, ps)
        reports, want_mask = [], 0
        broken_at = None
        for g in range(world):
            e, s, ps = true_in[g]
            mode = rng.choice(["ok", "ok", "ok", "wrong_s", "wrong_e", "poison"]) if g else "ok"
            us, ue, ups = s, e, ps
            poison = 0
            if mode == "wrong_s":
                us ^= 1
            elif mode == "wrong_e":
                ue ^= 1
            elif mode == "poison":
                poison = 1
            _, _, oe, os_, ops = serial_run(doc[cuts[g]:cuts[g + 1]], ue, us, ups)
            reports.append(((us, ue, ups), (os_, oe, ops, poison)))
            if broken_at is None:
                if mode != "ok":
                    want_mask |= 1 << g
                if mode in ("wrong_e", "poison"):
                    broken_at = g
        known, mask, exact = sharded.verify_reports(reports)
        assert mask == want_mask, (reports, mask, want_mask)
        assert known == (world if broken_at is None else broken_at + 1)
        for g in range(known):
            e, s, ps = true_in[g]
            assert exact[g] == (s, e, ps)
        # the stitched offsets: exclusive sums of the counts / bytes the launches reported
        counts = [(rng.randint(0, 1000), cuts[g + 1] - cuts[g]) for g in range(world)]
        known2, mask2, exact2 = sharded.verify_reports(reports, counts=counts, offsets=True)
        assert (known2, mask2) == (known, mask)
        for g in range(known):
            assert exact2[g][3:] == (sum(c for c, _ in counts[:g]), cuts[g])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _HostOps:
    """msj_sharded_ops over host memory: the 'device' is this process's heap, the 'kernel' is the serial spec."""

    def __init__(self):
        self.bufs = {}
        self.launches = 0
        self.fail_at = None  # launch number (1-based) whose run_shard fails
        self.poison_next = False  # the next launch WITHOUT MSJ_FLAG_TWO_PASS reports an expired wait (internal_error) and garbage
        self.two_pass_launches = 0
        O = sharded.MsjShardedOps
        f = dict(O._fields_)
        self.alloc = f["alloc"](self._alloc)
        self.free = f["free"](self._free)
        self.copy = f["copy"](self._copy)
        self.sync = f["sync"](lambda user, stream: 0)
        self.run_shard = f["run_shard"](self._run)
        self.ops = O(None, self.alloc, self.free, self.copy, self.sync, self.run_shard)  # no events: synchronous

    def _alloc(self, user, nbytes, pinned, out):
        b = ctypes.create_string_buffer(int(nbytes))
        self.bufs[ctypes.addressof(b)] = b
        out[0] = ctypes.addressof(b)
        return 0

    def _free(self, user, p, pinned):
        self.bufs.pop(p, None)

    def _copy(self, user, dst, src, nbytes, to_host, stream):
        ctypes.memmove(dst, src, nbytes)
        return 0

    def _run(self, user, d_shard, length, d_idx, cap, d_in, d_out, d_seg, max_seg, has_prefix, is_final, trailer_len,
             stream, flags):
        self.launches += 1
        if self.fail_at == self.launches:
            return -3
        data = ctypes.string_at(d_shard, length)
        cin = MsjCarry.from_address(d_in)
        if flags & 0x100:  # MSJ_FLAG_TWO_PASS: what the library re-issues a poisoned launch through
            self.two_pass_launches += 1
        elif self.poison_next:
            # a single-pass launch whose inter-workgroup wait expired: internal_error set, everything else untrustworthy
            self.poison_next = False
            out = MsjCarry.from_address(d_out)
            ctypes.memset(d_out, 0, ctypes.sizeof(MsjCarry))
            out.internal_error, out.count, out.bytes, out.in_string = 1, 12345, cin.bytes + length, 1
            return 0
        idx, bad, esc, instr, pnq = serial_run(data, cin.next_is_escaped, cin.in_string, cin.prev_scalar)
        out = MsjCarry.from_address(d_out)
        ctypes.memset(d_out, 0, ctypes.sizeof(MsjCarry))
        out.count, out.bytes = cin.count + len(idx), cin.bytes + length
        out.in_string, out.next_is_escaped, out.prev_scalar, out.unescaped_error = instr, esc, pnq, bad
        # like the kernel (finish_launch): clipped writes and a sticky flag when the index buffer is too small
        need = len(idx) + (3 if is_final else 0)
        out.capacity_error = int(need > cap)
        arr = (ctypes.c_uint32 * max(1, min(need, cap))).from_address(d_idx)
        for k, v in enumerate(idx[:cap]):
            arr[k] = v
        if is_final and need <= cap:
            arr[len(idx)], arr[len(idx) + 1], arr[len(idx) + 2] = trailer_len & 0xFFFFFFFF, trailer_len & 0xFFFFFFFF, 0
        return 0


class _DeferredOps(_HostOps):
    """The same 'device', but ASYNCHRONOUS like the real one: a stream is a FIFO of closures that run only when
    somebody waits (sync drains a stream; waiting for an event runs its stream up to the record and no further), with
    events and a second stream for the exchange -- the optional part of msj_sharded_ops.  What ran, and when, is
    what the tests of the per-slot waits look at."""

    MAIN, SIDE = 0x1000, 0x2000

    def __init__(self):
        super().__init__()
        self.queues = {self.MAIN: [], self.SIDE: [], None: []}
        self.events = {}
        self.log = []  # ("kernel", n) / ("exchange", n) in execution order
        O = sharded.MsjShardedOps
        f = dict(O._fields_)
        self.sync = f["sync"](self._sync)
        self.cbs = [f["event_create"](self._ev_create), f["event_destroy"](self._ev_destroy),
                    f["event_record"](self._ev_record), f["event_wait"](self._ev_wait), f["stream_wait"](self._stream_wait),
                    f["event_query"](self._ev_query)]
        self.ops = O(None, self.alloc, self.free, self.copy, self.sync, self.run_shard, *self.cbs,
                     f["event_elapsed_ns"](), self.SIDE)

    # every operation with a stream argument is only QUEUED
    def _copy(self, user, dst, src, nbytes, to_host, stream):
        self.queues[stream].append(lambda: ctypes.memmove(dst, src, nbytes))
        return 0

    def _run(self, user, *a):
        stream = a[11]
        if self.fail_at == self.launches + 1:  # a launch failure is reported at enqueue time
            self.launches += 1
            return -3

        def go():
            rc = _HostOps._run(self, user, *a)
            assert rc == 0
            self.log.append(("kernel", self.launches))
        self.queues[stream].append(go)
        return 0

    def enqueue_exchange(self, stream, fn):
        def go():
            fn()
            self.log.append(("exchange", sum(1 for k, _ in self.log if k == "exchange") + 1))
        self.queues[stream].append(go)

    def _drive(self, stream, until=None):
        q = self.queues[stream]
        while q and not (until is not None and self.events[until]["done"]):
            q.pop(0)()

    def _sync(self, user, stream):
        self._drive(stream)
        return 0

    def _ev_create(self, user, out):
        h = 0x100 + len(self.events)
        self.events[h] = {"done": False, "stream": None, "gen": 0}
        out[0] = h
        return 0

    def _ev_destroy(self, user, ev):
        self.events.pop(ev, None)

    def _ev_record(self, user, ev, stream):
        e = self.events[ev]
        e["done"], e["stream"] = False, stream
        e["gen"] += 1
        gen = e["gen"]

        def go():
            if e["gen"] == gen:
                e["done"] = True
        self.queues[stream].append(go)
        return 0

    def _ev_wait(self, user, ev):
        e = self.events[ev]
        if e["stream"] is not None or e["gen"]:
            self._drive(e["stream"], until=ev)
        assert e["done"], "waited for an event nothing records"
        return 0

    def _stream_wait(self, user, stream, ev):
        self.queues[stream].append(lambda: self._ev_wait(None, ev))
        return 0

    def _ev_query(self, user, ev):
        return int(self.events[ev]["done"])


def test_result_waits_for_its_own_submission_only():
    """msj_stage1_sharded_result used to drain the whole stream, so that with three
    submissions in flight result(0) returned when submission 2 had finished.  With event operations it waits for
    the arrival of ITS reports: on an asynchronous fake device (nothing runs until somebody waits) exactly submission
    0's kernel and exchange have run when result(0) returns; the exchange is on the side stream, behind the kernel's
    event; a refuted guess indexes again behind the kernels submitted since, and is counted."""
    L = sharded.lib()
    host = _DeferredOps()

    def allgather(comm, d_send, d_recv, nbytes, stream):
        assert stream == host.SIDE  # the exchange's own stream, not the kernels'
        host.enqueue_exchange(stream, lambda: ctypes.memmove(d_recv, d_send, nbytes))
        return 0

    cb = sharded.ALLGATHER_FN(allgather)
    x = sharded.MsjExchange(None, cb, 0, 1, 0, 0)
    h = ctypes.c_void_p()
    assert L.msj_sharded_create(None, ctypes.byref(x), ctypes.byref(host.ops), ctypes.byref(h)) == 0
    docs = [b'["abc",1]', b'{"k":[1,2,3],"s":"x y"}', b"[true,false,null,12.5]"]
    bufs = [ctypes.create_string_buffer(d, len(d)) for d in docs]
    idxs = [(ctypes.c_uint32 * (len(d) + 3))() for d in docs]

    def submit(k, spec=None):
        t = ctypes.c_uint32()
        assert L.msj_stage1_sharded_submit(h, ctypes.addressof(bufs[k]), len(docs[k]), ctypes.addressof(idxs[k]), len(docs[k]) + 3,
                                           len(docs[k]), 0, ctypes.byref(spec) if spec else None, None, 0, host.MAIN, 0,
                                           ctypes.byref(t)) == 0
        return t.value

    def result(t):
        code, total = ctypes.c_int32(), ctypes.c_uint64()
        assert L.msj_stage1_sharded_result(h, t, ctypes.byref(code), ctypes.byref(total), None, None, None) == 0
        return code.value, total.value

    tickets = [submit(k) for k in range(3)]
    assert host.log == [] and host.launches == 0  # nothing has run: submit only enqueues
    assert [L.msj_sharded_ticket_state(h, t) for t in tickets] == [0, 0, 0]
    for k in range(3):
        want = serial_run(docs[k])[0]
        assert result(tickets[k]) == (0, len(want)) and list(idxs[k][:len(want)]) == want
        # exactly the submissions up to k have run; the later ones are still queued
        assert host.log == [(kind, j + 1) for j in range(k + 1) for kind in ("kernel", "exchange")], (k, host.log)
        assert [L.msj_sharded_ticket_state(h, t) for t in tickets[k + 1:]] == [0] * (2 - k)
        assert L.msj_sharded_ticket_state(h, tickets[k]) == -1  # the ticket is gone
    # a refuted guess with two later submissions in flight: the second launch queues behind their kernels
    wrong = MsjCarry()
    wrong.in_string = 1
    host.log.clear()
    tickets = [submit(0, wrong), submit(1), submit(2)]
    want = serial_run(docs[0])[0]
    assert result(tickets[0]) == (0, len(want)) and list(idxs[0][:len(want)]) == want
    kinds = [k for k, _ in host.log]
    # its own round first; the second launch is the LAST kernel: the two submissions made since ran in front of it
    assert kinds == ["kernel", "exchange"] * 4, kinds
    for k in (1, 2):
        want_k = serial_run(docs[k])[0]
        assert list(idxs[k][:len(want_k)]) == want_k
    st = sharded.MsjShardedStats()
    assert L.msj_sharded_get_stats(h, ctypes.byref(st)) == 0
    assert (st.reruns, st.reruns_behind_queue) == (1, 1)
    for k in (1, 2):
        want = serial_run(docs[k])[0]
        assert result(tickets[k]) == (0, len(want)) and list(idxs[k][:len(want)]) == want
    # destroying with a submission nobody asked the result of drains it first (nothing is freed under pending work)
    t = submit(1)
    assert L.msj_sharded_ticket_state(h, t) == 0
    before = len(host.log)
    L.msj_sharded_destroy(h)
    assert len(host.log) == before + 2 and not host.queues[host.MAIN] and not host.queues[host.SIDE]


def _live_worker(rank, world, port, cases, q, deferred=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = sharded.lib()
        host = _DeferredOps() if deferred else _HostOps()

        def gather_now(d_send, d_recv, nbytes):
            mine = torch.frombuffer(bytearray(ctypes.string_at(d_send, nbytes)), dtype=torch.uint8)
            gathered = torch.empty(world * nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(gathered, mine)
            blob = gathered.numpy().tobytes()
            ctypes.memmove(d_recv, blob, len(blob))

        def allgather(comm, d_send, d_recv, nbytes, stream):
            if deferred:  # the collective runs when the fake device gets there, like an enqueued ncclAllGather
                host.enqueue_exchange(stream, lambda: gather_now(d_send, d_recv, nbytes))
            else:
                gather_now(d_send, d_recv, nbytes)
            return 0

        cb = sharded.ALLGATHER_FN(allgather)
        x = sharded.MsjExchange(None, cb, rank, world, 0, 0)
        h = ctypes.c_void_p()
        assert L.msj_sharded_create(None, ctypes.byref(x), ctypes.byref(host.ops), ctypes.byref(h)) == 0
        results = []
        pending = []  # (ticket, lo, hi, cap, idx, alloc, launches before): deferred = up to three submissions in flight

        def collect(ticket, lo, hi, cap, idx, alloc, before):
            code, total = ctypes.c_int32(), ctypes.c_uint64()
            local, used, place = MsjCarry(), MsjCarry(), sharded.MsjShardPlacement()
            assert L.msj_stage1_sharded_result(h, ticket, ctypes.byref(code), ctypes.byref(total), ctypes.byref(local),
                                               ctypes.byref(used), ctypes.byref(place)) == 0
            n = int(local.count)
            assert (int(place.byte_base), int(place.count), int(place.bytes)) == (lo, n, hi - lo)
            results.append((code.value, int(total.value), [int(idx[k]) + lo for k in range(min(n, cap))],
                            (used.next_is_escaped, used.in_string, used.prev_scalar),
                            None if deferred else host.launches - before,  # in flight together: not attributable
                            [int(idx[n + k]) for k in range(3)] if rank == world - 1 and n + 3 <= cap else None,
                            int(place.index_begin)))

        for case in cases:
            data_hex, cuts = case[0], case[1]
            short = case[2] if len(case) > 2 else None  # (rank, capacity): that rank's index buffer is too small
            if len(case) > 3 and case[3] == rank:       # this rank's first launch of the case is poisoned (an expired wait)
                host.poison_next = True
            data = bytes.fromhex(data_hex)
            lo, hi = cuts[rank], cuts[rank + 1]
            # the "device" holds the 64-byte halo in front of the shard, like a placed shard does
            halo = 64 if rank > 0 else 0
            pad = bytes(64 - min(64, lo)) if rank > 0 else b""
            alloc = ctypes.create_string_buffer(pad + data[max(0, lo - 64):hi], len(pad) + hi - max(0, lo - 64))
            d_shard = ctypes.addressof(alloc) + (len(pad) + min(64, lo) if rank > 0 else 0)
            idx = (ctypes.c_uint32 * (hi - lo + 3))()
            cap = short[1] if short and short[0] == rank else hi - lo + 3
            ticket = ctypes.c_uint32()
            before = host.launches
            rc = L.msj_stage1_sharded_submit(h, d_shard, hi - lo, ctypes.addressof(idx), cap, len(data), int(rank > 0),
                                             None, None, 0, None, 0, ctypes.byref(ticket))
            assert rc == 0 and halo in (0, 64)
            pending.append((ticket.value, lo, hi, cap, idx, alloc, before))
            if len(pending) == (3 if deferred else 1):
                collect(*pending.pop(0))
        while pending:
            collect(*pending.pop(0))
        if deferred:
            assert host.launches == len(cases) + L.msj_sharded_reruns(h)
        st = sharded.MsjShardedStats()
        assert L.msj_sharded_get_stats(h, ctypes.byref(st)) == 0
        assert (st.results, st.rounds, st.reruns) == (len(cases), L.msj_sharded_rounds(h), L.msj_sharded_reruns(h))
        assert st.stitch_device_ns == 0  # host operations: nothing times the device side
        # a launch that fails at submission leaves no slot busy behind it (both ranks fail their own launch, so
        # nobody is left alone in the all-gather)
        data = b'["' + b"a" * 200 + b'"]'
        alloc = ctypes.create_string_buffer(bytes(64) + data, 64 + len(data))
        idx = (ctypes.c_uint32 * (len(data) + 3))()
        for _ in range(4):  # more often than there are slots
            host.fail_at = host.launches + 1
            ticket = ctypes.c_uint32()
            rc = L.msj_stage1_sharded_submit(h, ctypes.addressof(alloc) + 64, len(data), ctypes.addressof(idx), len(data) + 3,
                                             2 * len(data), int(rank > 0), None, None, 0, None, 0, ctypes.byref(ticket))
            assert rc == -3, rc
        host.fail_at = None
        q.put((rank, results, int(L.msj_sharded_reruns(h)), int(L.msj_sharded_rounds(h)), host.two_pass_launches))
        L.msj_sharded_destroy(h)
    finally:
        dist.destroy_process_group()


def test_failed_rerun_frees_the_ticket():
    """a launch that fails while a rank indexes again used to return with the slot still busy, so
    three such failures exhausted the object.  World 1, host operations, a speculation that is refuted by the
    chain (the stream cannot start inside a string), a shard runner that fails the re-run."""
    L = sharded.lib()
    host = _HostOps()

    def allgather(comm, d_send, d_recv, nbytes, stream):
        ctypes.memmove(d_recv, d_send, nbytes)
        return 0

    cb = sharded.ALLGATHER_FN(allgather)
    x = sharded.MsjExchange(None, cb, 0, 1, 0, 0)
    h = ctypes.c_void_p()
    assert L.msj_sharded_create(None, ctypes.byref(x), ctypes.byref(host.ops), ctypes.byref(h)) == 0
    data = b'["abc",1]'
    buf = ctypes.create_string_buffer(data, len(data))
    idx = (ctypes.c_uint32 * (len(data) + 3))()
    wrong = MsjCarry()
    wrong.in_string = 1
    for attempt in range(5):  # more often than there are slots
        host.fail_at = host.launches + 2  # the first launch runs, the re-run fails
        ticket = ctypes.c_uint32()
        assert L.msj_stage1_sharded_submit(h, ctypes.addressof(buf), len(data), ctypes.addressof(idx), len(data) + 3, len(data),
                                           0, ctypes.byref(wrong), None, 0, None, 0, ctypes.byref(ticket)) == 0, attempt
        code, total = ctypes.c_int32(), ctypes.c_uint64()
        rc = L.msj_stage1_sharded_result(h, ticket.value, ctypes.byref(code), ctypes.byref(total), None, None, None)
        assert rc == -3, (attempt, rc)
        # the ticket is gone: asking again is a bad argument, not a hang
        assert L.msj_stage1_sharded_result(h, ticket.value, None, None, None, None, None) == -1
    host.fail_at = None
    ticket = ctypes.c_uint32()
    assert L.msj_stage1_sharded_submit(h, ctypes.addressof(buf), len(data), ctypes.addressof(idx), len(data) + 3, len(data),
                                       0, ctypes.byref(wrong), None, 0, None, 0, ctypes.byref(ticket)) == 0
    code, total = ctypes.c_int32(), ctypes.c_uint64()
    place = sharded.MsjShardPlacement()
    assert L.msj_stage1_sharded_result(h, ticket.value, ctypes.byref(code), ctypes.byref(total), None, None, ctypes.byref(place)) == 0
    want = serial_run(data)[0]
    assert (code.value, total.value, list(idx[:len(want)])) == (0, len(want), want)
    assert (place.index_begin, place.byte_base, place.count, place.bytes) == (0, 0, len(want), len(data))
    assert L.msj_sharded_reruns(h) == 6
    L.msj_sharded_destroy(h)


@pytest.mark.parametrize("deferred", [False, True], ids=["synchronous_ops", "events_three_in_flight"])
def test_gloo_world2_live_protocol(deferred):
    """The library's submit / result loop under gloo, world 2, on CPU: right guesses take one launch and one
    all-gather; a refuted guess makes exactly that rank launch again (the other one only re-contributes its
    report); results equal the serial spec of the whole stream, error codes included.  Once with operations that
    complete before they return, once on the asynchronous fake device (_DeferredOps: events, the exchange on its own
    stream, per-slot waits) with three submissions in flight the way bench.py runs the GPUs."""
    rng = random.Random(8)
    alpha = b'\\\\\\""a1 ,:[]{}'
    cases = []
    alphabets = [alpha, b'{}[]:,"\\ abtrue1.5e\n', b'"xyz\\" \t:,', b'{"k":"v w","n":[1,2.5e3,true,null]} ']
    for trial in range(48):
        a = alphabets[trial % len(alphabets)]
        data = bytes(rng.choice(a) for _ in range(rng.randint(150, 400)))
        cases.append((data, [0, rng.randint(70, len(data) - 70), len(data)]))
    # the cut falls inside a string whose closing quote follows a ':' -> rank 1's guess is refuted
    wrong = b'["' + b"a" * 90 + b':",1,2,"ee"]'
    cases.append((wrong, [0, 40, len(wrong)]))
    # a long backslash run across the cut (the halo cannot decide the escape carry)
    bs = b'["' + b"\\" * 150 + b'\\"x"' + b",1]" * 10
    cases.append((bs, [0, 120, len(bs)]))
    cases.append((bs, [0, 121, len(bs)]))
    # unclosed string / control character inside a string on rank 1
    cases.append((b'[1,2,"abc' + b" " * 80 + b'"x', [0, 50, 92]))
    cases.append((b'["' + b"s" * 70 + b"\n" + b's"]', [0, 30, 76]))
    # a cut inside a 12 KB string of digits and blanks: the shard's first 4 KiB contradict neither hypothesis (and hold no
    # quote for the neighbour rule), so the library looks at its first 64 KiB before it guesses -- no second launch
    longstr = b'["' + b"1 2 3 " * 2000 + b'x",7]'
    long_case = len(cases)
    cases.append((longstr, [0, 80, len(longstr)]))
    # an index buffer that is too small on ONE rank (a non-last rank used to clip silently and the
    # stream still came back as SUCCESS): the whole stream reports CAPACITY, whichever rank it is
    dense = b"[" * 100 + b"1" + b"]" * 100
    n_cases_plain = len(cases)
    cases.append((dense, [0, 96, len(dense)], (0, 50)))
    cases.append((dense, [0, 96, len(dense)], (1, 50)))
    cases.append((dense, [0, 96, len(dense)], (1, len(dense) - 96 + 2)))  # room for the indices, not for the trailer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    payload = [(c[0].hex(),) + tuple(c[1:]) for c in cases]
    procs = [ctx.Process(target=_live_worker, args=(r, 2, port, payload, q, deferred)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=180) for _ in range(2)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total_reruns = 0
    for k, case in enumerate(cases):
        data, cuts = case[0], case[1]
        idx, bad, esc, instr, pnq = serial_run(data)
        want_code = 15 if instr else 14 if bad else 13 if not idx else 0
        r0, r1 = res[0][0][k], res[1][0][k]
        # the stitched offsets (msj_shard_placement): rank 1's first index is index_begin of the stream's array
        assert r0[6] == 0 and r1[6] == sum(1 for v in idx if v < cuts[1]), (k, r0[6], r1[6])
        assert r0[1] == r1[1] == len(idx)
        if k >= n_cases_plain:
            assert r0[0] == r1[0] == 1, (k, r0[0], r1[0])  # CAPACITY on every rank
            continue
        assert r0[0] == r1[0] == want_code, (k, r0[0], want_code)
        assert r0[2] + r1[2] == idx, f"case {k}"
        e, s, ps = serial_state(data[:cuts[1]])
        assert r1[3] == (e, s, ps) and r0[3] == (0, 0, 0)
        if deferred:
            if k == long_case:
                assert r1[3] == (0, 1, 0), r1[3]
        else:
            assert r0[4] == 1 and r1[4] in (1, 2)  # rank 0 never launches twice
            if k == long_case:
                assert r1[4] == 1 and r1[3] == (0, 1, 0), r1[3:5]  # the longer look got it right: one launch
            total_reruns += r1[4] - 1
        if want_code in (0, 13):
            assert r1[5] == [len(data), len(data), 0]
    if deferred:
        total_reruns = res[1][1]
    assert res[0][1] == 0 and res[1][1] == total_reruns and total_reruns >= 2  # the refuted-guess cases did re-run
    assert res[0][2] == res[1][2] == len(cases) + total_reruns  # one all-gather per launch round, on every rank


def test_gloo_world8_live_protocol():
    """the live C protocol at WORLD 8 -- eight separate processes over gloo, each on the
    asynchronous fake device (_DeferredOps: streams that run only when somebody waits, events, the exchange on its own
    stream) with three submissions in flight, the way bench.py drives eight GPUs.  Streams of random text cut into eight
    shards at arbitrary bytes; a rank in the MIDDLE of the chain whose in-string guess is refuted (it indexes again, the
    ranks behind it only re-contribute); a rank whose launch is POISONED (an expired wait: internal_error) -- it goes
    through MSJ_FLAG_TWO_PASS, and the ranks behind it, whose carries the replay could not judge, are verified in the
    next round; both in one stream.  Every rank's indices, placement, carry-in and the stream's code against the serial
    spec of the whole stream."""
    world = 8
    rng = random.Random(88)
    alphabets = [b'\\\\\\""a1 ,:[]{}', b'{}[]:,"\\ abtrue1.5e\n', b'"xyz\\" \t:,', b'{"k":"v w","n":[1,2.5e3,true,null]} ']

    def cuts_of(n, forced=()):
        while True:
            inner = sorted(set(list(forced) + [rng.randint(70, n - 70) for _ in range(world - 1 - len(forced))]))
            c = [0] + inner + [n]
            if len(c) == world + 1 and all(b - a >= 66 for a, b in zip(c[:-1], c[1:])):
                return c

    cases = []
    for trial in range(20):
        a = alphabets[trial % len(alphabets)]
        data = bytes(rng.choice(a) for _ in range(rng.randint(900, 1600)))
        cases.append((data, cuts_of(len(data)), None, None))
    # rank 4's shard starts inside a string whose closing quote is followed by ':' (the neighbour rule guesses "outside")
    filler = b'{"k":[1,2,3],"m":"v"},'
    pre = b"[" + filler * 25
    refuted = pre + b'"' + b"a" * 90 + b':",1,2,"ee",' + filler * 30 + b"0]"
    cut4 = len(pre) + 40
    c = cuts_of(len(refuted))
    c = sorted(set([0, len(refuted), cut4] + [v for v in c[1:-1] if abs(v - cut4) >= 70][:world - 2]))
    while len(c) < world + 1:  # (fill up deterministically if a random cut fell too close)
        v = rng.randint(70, len(refuted) - 70)
        if all(abs(v - u) >= 70 for u in c):
            c = sorted(c + [v])
    k4 = c.index(cut4)
    refuted_case = len(cases)
    cases.append((refuted, c, None, None))
    # a poisoned launch on rank 5 of an ordinary stream ...
    data = bytes(rng.choice(alphabets[3]) for _ in range(1400))
    poisoned_case = len(cases)
    cases.append((data, cuts_of(len(data)), None, 5))
    # ... and both at once: the refuted rank in front of the poisoned one
    both_case = len(cases)
    cases.append((refuted, c, None, min(world - 1, k4 + 2)))
    # CAPACITY on one rank of eight
    dense = b"[" * 400 + b"1" + b"]" * 400
    cap_case = len(cases)
    cases.append((dense, [0, 100, 200, 300, 400, 500, 600, 700, len(dense)], (3, 50), None))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    payload = [(cs[0].hex(), cs[1], cs[2], cs[3]) for cs in cases]
    procs = [ctx.Process(target=_live_worker, args=(r, world, port, payload, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, (data, cuts, short, poison) in enumerate(cases):
        idx, bad, esc, instr, pnq = serial_run(data)
        want_code = 15 if instr else 14 if bad else 13 if not idx else 0
        rows = [res[r][0][k] for r in range(world)]
        assert all(row[1] == len(idx) for row in rows), k  # the stream's total on every rank
        if k == cap_case:
            assert all(row[0] == 1 for row in rows), [row[0] for row in rows]  # CAPACITY, whichever rank clipped
            continue
        assert all(row[0] == want_code for row in rows), (k, [row[0] for row in rows], want_code)
        assert sum((row[2] for row in rows), []) == idx, f"case {k}"
        for r, row in enumerate(rows):
            assert row[6] == sum(1 for v in idx if v < cuts[r]), (k, r)          # index_begin of the stitched array
            assert row[3] == (serial_state(data[:cuts[r]]) if r else (0, 0, 0)), (k, r)  # the carry the shard was indexed with
        if want_code in (0, 13):
            assert rows[-1][5] == [len(data), len(data), 0]
    reruns = [res[r][1] for r in range(world)]
    two_pass = [res[r][3] for r in range(world)]
    rounds = [res[r][2] for r in range(world)]
    p2 = min(world - 1, k4 + 2)
    assert sum(two_pass) == 2 and two_pass[5] >= 1 and two_pass[p2] >= 1, (two_pass, p2)  # exactly the two poisoned launches went through the two-pass kernels
    assert reruns[k4] >= 2, (k4, reruns)            # the refuted rank indexed again, in both streams that hold the case
    assert reruns[0] == 0                            # rank 0 is never refuted
    assert len(set(rounds)) == 1 and rounds[0] >= len(cases) + 3  # every rank took part in every all-gather round
