"""Byte-range sharding of one JSON stream over the GPUs of a node.

The reference has no parallelism of any kind; the only
cross-block state of its stage 1 is three bits and a count
(json_escape_scanner.mojo:13, json_string_scanner.mojo:49, json_scanner.mojo:57,
json_structural_indexer.mojo:34).  Sharding therefore needs one tiny exchange:

  1. every rank derives (next_is_escaped, prev_scalar) from the 64 bytes in front of
     its shard (pure byte inspection) and SPECULATES in_string from the context of the
     first unescaped quote;
  2. one single-pass kernel launch per GPU with that carry-in;
  3. ONE all-gather of (carry used, carry out, count, error bits) per rank: every rank
     replays the chain, which proves or refutes each speculation; a refuted rank runs
     its shard again with the now exact carry (rare), so the result is always exact.

Collectives go through torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests); payloads are a few bytes to a few KiB,
so they are latency-bound.  No bulk data ever crosses xGMI: input shards are
placed on their GPU up front and the index arrays stay shard-local.

The functions in the first half are pure host logic (tested with gloo on CPU);
``ShardedStage1`` at the bottom drives the HIP kernels.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import errors
from ._lib import MsjCarry

TAIL_BYTES = 4096
_NONSCALAR = frozenset([0x20, 0x09, 0x0A, 0x0D, 0x0C, 0x1A, 0x2C, 0x3A, 0x5B, 0x5D, 0x7B, 0x7D])


def boundary_carry(tails, complete):
    """Exact (next_is_escaped, prev_scalar) after the last byte of tails[-1].

    tails: list of bytes objects, the trailing bytes of shards 0..g in stream
    order; complete[j] is True when tails[j] is the *whole* shard j.  Returns
    None when the available bytes cannot decide (a tail consisting only of
    backslashes that does not reach its shard's start) -- the caller then
    gathers longer tails.

    Semantics (json_escape_scanner.mojo:18-45, json_scanner.mojo:64-79): a byte
    is escaped iff it is preceded by an odd-length run of backslashes;
    prev_scalar is 1 iff the last byte is a scalar character (not whitespace,
    not an operator) that is not an unescaped quote.
    """

    def run_ending_before(shard, pos):
        """Length of the backslash run ending just before (shard, pos); None if unknown."""
        n = 0
        j, p = shard, pos
        while True:
            t = tails[j]
            while p > 0 and t[p - 1] == 0x5C:
                p -= 1
                n += 1
            if p > 0:
                return n
            if not complete[j]:
                return None  # ran off the front of a truncated tail
            j -= 1
            if j < 0:
                return n  # start of the stream
            p = len(tails[j])

    g = len(tails) - 1
    # skip empty shards at the end (cannot happen with the partitioner, but be exact)
    while g >= 0 and len(tails[g]) == 0:
        if not complete[g]:
            return None
        g -= 1
    if g < 0:
        return (0, 0)
    r = run_ending_before(g, len(tails[g]))
    if r is None:
        return None
    if r >= 1:
        return (r & 1, 1)
    c = tails[g][-1]
    if c in _NONSCALAR:
        return (0, 0)
    if c != 0x22:
        return (0, 1)
    r2 = run_ending_before(g, len(tails[g]) - 1)
    if r2 is None:
        return None
    return (0, r2 & 1)  # escaped quote = non-quote scalar


def resolve_boundaries(all_tails, all_lens):
    """Per-rank (next_is_escaped, prev_scalar) carry-in from the gathered tails."""
    world = len(all_tails)
    complete = [len(all_tails[j]) == all_lens[j] for j in range(world)]
    out = [(0, 0)]
    for g in range(1, world):
        res = boundary_carry(all_tails[:g], complete[:g])
        out.append(res)
    return out


def parity_prefix(parities):
    """in_string at the start of each shard given the per-shard quote parities."""
    s, out = 0, []
    for p in parities:
        out.append(s)
        s ^= int(p) & 1
    return out, s


def global_code(final_in_string, any_unescaped, total_count, any_utf8, any_internal, strict_utf8):
    """Reference precedence, json_structural_indexer.mojo:147-186."""
    if any_internal:
        return errors.UNEXPECTED_ERROR
    if final_in_string:
        return errors.UNCLOSED_STRING
    if any_unescaped:
        return errors.UNESCAPED_CHARS
    if total_count == 0:
        return errors.EMPTY
    if strict_utf8 and any_utf8:
        return errors.UTF8_ERROR
    return errors.SUCCESS


def _all_gather_bytes(payload, device, group=None):
    """all_gather of one fixed-size uint8 tensor per rank -> list of bytes."""
    world = dist.get_world_size(group)
    # NCCL (= RCCL) moves device tensors; gloo (CPU tests, one-GPU rehearsal) host tensors
    if dist.get_backend(group) != "nccl":
        device = "cpu"
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return [bytes(o.cpu().numpy().tobytes()) for o in outs]


def exchange_tails(tail_bytes, shard_len, device, group=None, tail_cap=TAIL_BYTES):
    """Collective 1: gather every shard's last <= tail_cap bytes and its length."""
    hdr = np.array([shard_len, len(tail_bytes)], dtype=np.uint64).tobytes()
    body = bytes(tail_bytes).rjust(tail_cap, b"\x00")
    got = _all_gather_bytes(hdr + body, device, group)
    lens, tails = [], []
    for blob in got:
        h = np.frombuffer(blob[:16], dtype=np.uint64)
        lens.append(int(h[0]))
        tails.append(blob[16 + tail_cap - int(h[1]):])
    return tails, lens


def exchange_words(words, device, group=None):
    """Collectives 2/3: gather a few uint64 words per rank."""
    got = _all_gather_bytes(np.asarray(words, dtype=np.uint64).tobytes(), device, group)
    return [np.frombuffer(b, dtype=np.uint64).copy() for b in got]


_OPENERS = frozenset(b":,[{")
_CLOSERS = frozenset(b":,]}")
_WS = frozenset(b" \t\n\r")


def halo_carry(halo):
    """(next_is_escaped, prev_scalar) from the bytes right before a shard, or None when the
    halo cannot decide (it ends in a backslash run that reaches its first byte)."""
    return boundary_carry([bytes(halo)], [False])


def guess_in_string(halo, head, next_is_escaped):
    """Speculative in_string at a shard start, from the context of the first unescaped quote.

    A quote preceded by one of ``: , [ {`` opens a string (so the shard starts outside
    one); a quote followed by ``: , ] }`` closes one.  This is only a GUESS: after the
    kernels have run, the all-gathered carries prove or refute it and a refuted rank runs
    again with the exact carry, so correctness never depends on it.
    """
    esc = next_is_escaped
    q = -1
    for i, c in enumerate(head):
        escaped = esc
        if escaped:
            esc = 0
        elif c == 0x5C:
            esc = 1
        if c == 0x22 and not escaped:
            q = i
            break
    if q < 0:
        return 0
    ctx = bytes(halo) + bytes(head)
    k = len(halo) + q - 1
    while k >= 0 and ctx[k] in _WS:
        k -= 1
    if k >= 0 and ctx[k] in _OPENERS:
        return 0
    k = len(halo) + q + 1
    while k < len(ctx) and ctx[k] in _WS:
        k += 1
    if k < len(ctx) and ctx[k] in _CLOSERS:
        return 1
    return 0


def verify_chain(reports):
    """reports[g] = dict(s_used, e_used, ps_used, s_out, e_out, ps_out) from every rank.

    Returns (first_wrong, true_in) where true_in[g] = exact (s, e, ps) at the start of
    shard g for every g <= first_wrong (first_wrong == world when every guess was right).
    A shard's quote parity is s_out ^ s_used whatever s_used was; its e_out / ps_out do
    not depend on the string state at all (only on its own bytes, unless the whole shard
    is backslashes, which the re-run loop also covers because it re-verifies).
    """
    world = len(reports)
    true_in = [(0, 0, 0)]
    for g in range(world):
        r = reports[g]
        used = (int(r["s_used"]), int(r["e_used"]), int(r["ps_used"]))
        if used[1:] != true_in[g][1:]:
            return g, true_in  # its outputs were computed from wrong escape carries
        parity = int(r["s_out"]) ^ used[0]
        nxt = (true_in[g][0] ^ parity, int(r["e_out"]), int(r["ps_out"]))
        if used[0] != true_in[g][0]:
            return g, true_in  # indices emitted under the wrong string state
        true_in.append(nxt)
    return world, true_in


class ShardedStage1:
    """Drives the HIP kernels for this rank's shard (one process per GPU).

    One single-pass kernel launch per shard and ONE all-gather (128 bytes per rank) in
    the common case: the carries into the shard are derived from its own 64-byte halo and
    a speculative in_string, the all-gathered end states verify the whole chain, and only
    a rank whose speculation was refuted runs again.

    ``submit`` enqueues the kernel and the all-gather and returns at once; ``result``
    waits for the gathered carries (on a side stream, so that the next ``submit`` --
    another document, or the next benchmark step -- can already be running on the GPU)
    and verifies them.  ``run`` = ``result(submit(...))``.
    """

    DEPTH = 3  # submissions that may be in flight (slots for their carries)

    def __init__(self, dev, rank, world, group=None, always_gather=False):
        """always_gather: take the collective path even for world == 1 (lets a one-GPU box
        exercise the RCCL / side-stream plumbing)."""
        self.dev, self.rank, self.world, self.group = dev, rank, world, group
        self.always_gather = always_gather
        self.reruns = 0
        self.last_spec = None  # the carry-in that the last verified run of this rank actually used
        self._slots = None
        self._next = 0

    # ---- speculation from local bytes only (host logic; compute once per placed shard)
    def speculate(self, has_prefix, d_shard=None, shard_len=0, d_halo=None, host_halo=None, host_head=None):
        """(in_string, next_is_escaped, prev_scalar) assumed at the shard's first byte.

        host_halo / host_head: host copies of the 64 stream bytes before the shard and of
        its first <= 4096 bytes -- whoever placed the shard on the GPU had them in host
        memory; without them they are fetched from the device."""
        if not has_prefix:
            return (0, 0, 0)
        if host_halo is None:
            if d_halo is None:
                d_halo = torch.as_strided(d_shard, (64,), (1,), d_shard.storage_offset() - 64)
            host_halo = d_halo.cpu().numpy().tobytes()
        if host_head is None:
            host_head = d_shard[: min(4096, shard_len)].cpu().numpy().tobytes()
        hc = halo_carry(bytes(host_halo))
        e_used, ps_used = hc if hc is not None else (0, 1)
        s_used = guess_in_string(bytes(host_halo), bytes(host_head), e_used)
        return (s_used, e_used, ps_used)

    def _make_slots(self):
        dev = self.dev
        nccl = dist.get_backend(self.group) == "nccl"
        slots = []
        for _ in range(self.DEPTH):
            sl = dict(mine=torch.zeros(128, dtype=torch.uint8, device=dev.device), spec=None)
            if nccl:
                sl["gathered"] = torch.empty(self.world * 128, dtype=torch.uint8, device=dev.device)
                sl["host"] = torch.empty(self.world * 128, dtype=torch.uint8).pin_memory()
                sl["event"] = torch.cuda.Event()
            slots.append(sl)
        self._slots = slots
        self._nccl = nccl
        self._side = torch.cuda.Stream(device=dev.device) if nccl else None

    def submit(self, d_shard, shard_len, d_idx, total_len, has_prefix, flags=0, segments=None,
               d_halo=None, host_halo=None, host_head=None, speculation=None):
        """Enqueue this rank's shard (kernel + all-gather of the carries); returns a ticket."""
        if self.world == 1 and not self.always_gather:
            dev = self.dev
            cin = dev.make_carry(0, 0, 0)
            cout = dev.new_carry()
            dev.shard(d_shard, shard_len, d_idx, cin, cout, segments=segments, has_prefix=has_prefix,
                      is_final=True, trailer_len=total_len, flags=flags)
            return dict(single=cout)
        if speculation is None:
            speculation = self.speculate(has_prefix, d_shard, shard_len, d_halo, host_halo, host_head)
        if self._slots is None:
            self._make_slots()
        sl = self._slots[self._next]
        self._next = (self._next + 1) % self.DEPTH
        args = dict(d_shard=d_shard, shard_len=shard_len, d_idx=d_idx, total_len=total_len,
                    has_prefix=has_prefix, flags=flags, segments=segments)
        self._launch(sl, speculation, args)
        return dict(slot=sl, args=args)

    def _launch(self, sl, speculation, a):
        """Kernel + all-gather of (carry used | carry out) = 128 bytes per rank, both
        stream-ordered behind each other on the device; nothing waits on the host here."""
        dev = self.dev
        last = self.rank == self.world - 1
        mine = sl["mine"]
        if sl["spec"] != speculation:  # the carry used lives in the first half of the 128-byte report
            mine[:64].copy_(dev.make_carry(*speculation))
            sl["spec"] = speculation
        dev.shard(a["d_shard"], a["shard_len"], a["d_idx"], mine[:64], mine[64:], segments=a["segments"],
                  has_prefix=a["has_prefix"], is_final=last, trailer_len=a["total_len"], flags=a["flags"])
        if self._nccl:
            work = dist.all_gather_into_tensor(sl["gathered"], mine, group=self.group, async_op=True)
            with torch.cuda.stream(self._side):
                work.wait()  # the side stream (not the host, not the compute stream) waits for the collective
                sl["host"].copy_(sl["gathered"], non_blocking=True)
                sl["event"].record(self._side)
            sl["blob"] = None
        else:  # gloo (CPU tests, rehearsal): host tensors, synchronous
            m = mine.cpu()
            gathered = torch.empty(self.world * 128, dtype=torch.uint8)
            dist.all_gather_into_tensor(gathered, m, group=self.group)
            sl["blob"] = gathered.numpy().tobytes()

    def _collect(self, sl):
        if sl["blob"] is None:
            sl["event"].synchronize()
            sl["blob"] = sl["host"].numpy().tobytes()
        blob = sl["blob"]
        reports, carries = [], []
        for g in range(self.world):
            used = MsjCarry.from_buffer_copy(blob[128 * g:128 * g + 64])
            out = MsjCarry.from_buffer_copy(blob[128 * g + 64:128 * g + 128])
            carries.append(out)
            reports.append(dict(s_used=used.in_string, e_used=used.next_is_escaped,
                                ps_used=used.prev_scalar, s_out=out.in_string,
                                e_out=out.next_is_escaped, ps_out=out.prev_scalar))
        return reports, carries

    def result(self, ticket, flags=None):
        """Wait for a submission; returns (code, total_count, local msj_carry)."""
        if "single" in ticket:
            c = self.dev.fetch(ticket["single"])
            return int(c.code), int(c.count), c
        sl, a = ticket["slot"], ticket["args"]
        while True:
            reports, carries = self._collect(sl)
            first_wrong, true_in = verify_chain(reports)
            if first_wrong == self.world:
                break
            # every rank sees the same reports, so all agree on who runs again; ranks after
            # first_wrong keep their speculation and are re-verified in the next round
            spec = sl["spec"]
            if self.rank == first_wrong:
                spec = tuple(true_in[first_wrong])
                self.reruns += 1
            self._launch(sl, spec, a)
        self.last_spec = tuple(sl["spec"])  # exact by now: a caller re-submitting the same shard can pass it
        c = carries[self.rank]
        total = sum(int(x.count) for x in carries)
        code = global_code(int(carries[-1].in_string), any(int(x.unescaped_error) for x in carries), total,
                           any(int(x.utf8_error) for x in carries), any(int(x.internal_error) for x in carries),
                           bool(a["flags"] & 1))
        return code, total, c

    def run(self, d_shard, shard_len, d_idx, total_len, has_prefix, flags=0, segments=None,
            d_halo=None, host_halo=None, host_head=None, speculation=None):
        """Index this rank's shard.  Returns (code, total_count, local msj_carry)."""
        return self.result(self.submit(d_shard, shard_len, d_idx, total_len, has_prefix, flags=flags,
                                       segments=segments, d_halo=d_halo, host_halo=host_halo,
                                       host_head=host_head, speculation=speculation))
