"""Byte-range sharding of one JSON stream over the GPUs of a node: thin binding of the library's
native N-GPU entry points (include/msj_stage1.h ``msj_stage1_sharded_submit`` / ``_result``,
csrc/sharded.cpp).

The reference has no parallelism of any kind; the only cross-block state of
its stage 1 is three bits and a count (json_escape_scanner.mojo:13, json_string_scanner.mojo:49,
json_scanner.mojo:57, json_structural_indexer.mojo:34).  The protocol -- speculate the carries from
the shard's own bytes, one single-pass launch, ONE all-gather of a 128-byte report per rank, replay
the chain, only refuted ranks index again -- lives in C so that any host (the Mojo shim of
INTEGRATION.md included) reaches it; this module only supplies the exchange:

  * backend "nccl": an RCCL communicator of its own (ncclCommInitRank, the id broadcast through
    torch.distributed) handed to ``msj_exchange_rccl``: the library calls ncclAllGather itself, on the
    kernel's stream, over xGMI;
  * backend "gloo" (CPU tests, several ranks sharing one GPU): a callback that moves the 128 bytes
    through torch.distributed -- also what backend "nccl" falls back to (all ranks together, with a line on
    stderr) when the communicator cannot be had, and what ``ShardedStage1(exchange="torch")`` selects.

No bulk data ever crosses xGMI: input shards are placed on their GPU up front and the index arrays
stay shard-local.
"""
import ctypes
import os
import sys

import torch
import torch.distributed as dist

from . import _lib
from ._lib import MsjCarry


class MsjShardReport(ctypes.Structure):
    """``msj_shard_report``: the carry a rank's launch used, and the state after its last byte."""

    _fields_ = [("used", MsjCarry), ("out", MsjCarry)]


ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                ctypes.c_void_p)


class MsjExchange(ctypes.Structure):
    _fields_ = [("comm", ctypes.c_void_p), ("allgather", ALLGATHER_FN), ("rank", ctypes.c_uint32),
                ("world", ctypes.c_uint32), ("owns_comm", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]


class MsjShardPlacement(ctypes.Structure):
    """``msj_shard_placement``: where a shard's indices and bytes sit in the stream (the stitched offsets)."""

    _fields_ = [("index_begin", ctypes.c_uint64), ("byte_base", ctypes.c_uint64), ("count", ctypes.c_uint64),
                ("bytes", ctypes.c_uint64)]


class MsjShardedStats(ctypes.Structure):
    _fields_ = [("results", ctypes.c_uint64), ("rounds", ctypes.c_uint64), ("reruns", ctypes.c_uint64),
                ("stitch_device_ns", ctypes.c_uint64), ("result_wait_ns", ctypes.c_uint64),
                ("kernel_device_ns", ctypes.c_uint64), ("last_kernel_ns", ctypes.c_uint64),
                ("last_stitch_ns", ctypes.c_uint64), ("reruns_behind_queue", ctypes.c_uint64),
                ("reserved", ctypes.c_uint64 * 3)]


STATS_FIELDS = ("results", "rounds", "reruns", "stitch_device_ns", "result_wait_ns", "kernel_device_ns",
                "last_kernel_ns", "last_stitch_ns", "reruns_behind_queue")
KERNEL_DONE, REPORTS_IN = 1, 2  # MSJ_SHARDED_* bits of msj_sharded_ticket_state


class MsjShardedOps(ctypes.Structure):
    _fields_ = [
        ("user", ctypes.c_void_p),
        ("alloc", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p))),
        ("free", ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)),
        ("copy", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64,
                                  ctypes.c_int, ctypes.c_void_p)),
        ("sync", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)),
        ("run_shard", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                       ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32)),
        # optional (NULL: operations that complete before they return): events and the exchange's own stream
        ("event_create", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p))),
        ("event_destroy", ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)),
        ("event_record", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)),
        ("event_wait", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)),
        ("stream_wait", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)),
        ("event_query", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p)),
        ("event_elapsed_ns", ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.POINTER(ctypes.c_uint64))),
        ("side_stream", ctypes.c_void_p),
    ]


_bound = False


def lib():
    """libmsj_stage1.so with the sharded entry points' signatures."""
    global _bound
    L = _lib.load()
    if not _bound:
        vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int32
        L.msj_shard_speculate.restype = i32
        L.msj_shard_speculate.argtypes = [ctypes.c_char_p, u64, ctypes.c_char_p, u64, ctypes.POINTER(MsjCarry)]
        L.msj_shard_verify.restype = i32
        L.msj_shard_verify.argtypes = [ctypes.POINTER(MsjShardReport), u32, ctypes.POINTER(MsjCarry), ctypes.POINTER(u64)]
        L.msj_shard_global_code.restype = i32
        L.msj_shard_global_code.argtypes = [ctypes.POINTER(MsjShardReport), u32, u32, ctypes.POINTER(u64)]
        L.msj_exchange_rccl.restype = i32
        L.msj_exchange_rccl.argtypes = [vp, u32, u32, ctypes.c_char_p, ctypes.POINTER(MsjExchange)]
        L.msj_sharded_create.restype = i32
        L.msj_sharded_create.argtypes = [vp, ctypes.POINTER(MsjExchange), ctypes.POINTER(MsjShardedOps), ctypes.POINTER(vp)]
        L.msj_sharded_destroy.restype = None
        L.msj_sharded_destroy.argtypes = [vp]
        L.msj_sharded_reruns.restype = u64
        L.msj_sharded_reruns.argtypes = [vp]
        L.msj_sharded_rounds.restype = u64
        L.msj_sharded_rounds.argtypes = [vp]
        L.msj_stage1_sharded_submit.restype = i32
        L.msj_stage1_sharded_submit.argtypes = [vp, vp, u64, vp, u64, u64, i32, ctypes.POINTER(MsjCarry), vp, u32, vp, u32,
                                                ctypes.POINTER(u32)]
        L.msj_stage1_sharded_result.restype = i32
        L.msj_stage1_sharded_result.argtypes = [vp, u32, ctypes.POINTER(i32), ctypes.POINTER(u64), ctypes.POINTER(MsjCarry),
                                                ctypes.POINTER(MsjCarry), ctypes.POINTER(MsjShardPlacement)]
        L.msj_sharded_get_stats.restype = i32
        L.msj_sharded_get_stats.argtypes = [vp, ctypes.POINTER(MsjShardedStats)]
        L.msj_sharded_ticket_state.restype = i32
        L.msj_sharded_ticket_state.argtypes = [vp, u32]
        L.msj_exchange_release.restype = None
        L.msj_exchange_release.argtypes = [ctypes.POINTER(MsjExchange)]
        L.msj_debug_set_segment_bytes.restype = i32
        L.msj_debug_set_segment_bytes.argtypes = [vp, u64]
        L.msj_copy_to_host.restype = i32
        L.msj_copy_to_host.argtypes = [vp, vp, vp, u64, vp]
        L.msj_copy_to_device.restype = i32
        L.msj_copy_to_device.argtypes = [vp, vp, ctypes.c_char_p, u64, vp]
        _bound = True
    return L


def speculate_bytes(halo, head):
    """``msj_shard_speculate``: (in_string, next_is_escaped, prev_scalar) a shard may assume from the <= 64 stream
    bytes in front of it and its own first bytes."""
    c = MsjCarry()
    rc = lib().msj_shard_speculate(bytes(halo), len(halo), bytes(head), len(head), ctypes.byref(c))
    assert rc == 0, rc
    return (int(c.in_string), int(c.next_is_escaped), int(c.prev_scalar))


def verify_reports(reports, counts=None, offsets=False):
    """``msj_shard_verify`` on a list of (used, out) pairs of (in_string, next_is_escaped, prev_scalar[, internal_error])
    tuples; counts: optional per-rank (structurals, bytes) of the launches.  Returns (known, rerun_mask, exact_in) with
    exact_in[g] = (s, e, ps) for g < known -- with offsets=True (s, e, ps, index_begin, byte_base)."""
    world = len(reports)
    arr = (MsjShardReport * world)()
    for g, (used, out) in enumerate(reports):
        arr[g].used.in_string, arr[g].used.next_is_escaped, arr[g].used.prev_scalar = used[:3]
        arr[g].out.in_string, arr[g].out.next_is_escaped, arr[g].out.prev_scalar = out[:3]
        if len(out) > 3:
            arr[g].out.internal_error = out[3]
        if counts is not None:
            arr[g].out.count, arr[g].out.bytes = counts[g]
    exact = (MsjCarry * world)()
    mask = ctypes.c_uint64(0)
    known = lib().msj_shard_verify(arr, world, exact, ctypes.byref(mask))
    assert known >= 0, known
    return known, int(mask.value), [(int(exact[g].in_string), int(exact[g].next_is_escaped), int(exact[g].prev_scalar)) +
                                    ((int(exact[g].count), int(exact[g].bytes)) if offsets else ())
                                    for g in range(known)]


def _torch_rccl_path():
    return os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def _all_ok(ok, device, group):
    """True iff `ok` holds on every rank (one small all-reduce): the ranks take every turn together."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t.item()) == 1


def _rccl_communicator(rank, world, device, group):
    """An RCCL communicator of this module's own (the one inside torch's process group is not reachable):
    ncclGetUniqueId on rank 0, the 128-byte id broadcast through torch.distributed, ncclCommInitRank.

    Every step that can fail on one rank alone (loading the library, ncclGetUniqueId, ncclCommInitRank) is followed
    by an agreement of all ranks, so that nobody enters a collective the others have left.  Returns
    (rccl, comm, path, ncclCommCount) or None -- the same on every rank."""
    path = _torch_rccl_path()
    rccl, uid, why = None, _NcclUniqueId(), None
    try:
        rccl = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        if rank == 0:
            rc = rccl.ncclGetUniqueId(ctypes.byref(uid))
            if rc != 0:
                why = f"ncclGetUniqueId: {rc}"
    except OSError as e:
        why = repr(e)
    if not _all_ok(why is None, device, group):
        if why:
            print(f"mojo_simdjson_amd.sharded: no RCCL communicator of our own ({why})", file=sys.stderr)
        return None
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ctypes.memmove(ctypes.byref(uid), t.cpu().numpy().tobytes(), 128)
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int]
    rc = rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
    count = ctypes.c_int(0)
    if rc == 0:
        rccl.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        rc = rccl.ncclCommCount(comm, ctypes.byref(count))
    if rc != 0:
        print(f"mojo_simdjson_amd.sharded: ncclCommInitRank / ncclCommCount: {rc}", file=sys.stderr)
    if not _all_ok(rc == 0 and count.value == world, device, group):
        if comm:
            rccl.ncclCommDestroy(comm)
        return None
    return rccl, comm, path, int(count.value)


class ShardedStage1:
    """This rank's side of the sharded path (one process per GPU).

    ``submit`` enqueues kernel + all-gather + read-back and returns at once; ``result`` waits for the gathered
    reports, replays the chain and -- only if this rank's speculation was refuted -- indexes the shard again.
    ``run`` = ``result(submit(...))``.  Up to ``DEPTH`` submissions may be in flight."""

    DEPTH = 3

    def __init__(self, dev, rank, world, group=None, always_gather=False, exchange="rccl"):
        """always_gather: take the collective path even for world == 1 (lets a one-GPU box exercise the RCCL
        plumbing).  exchange: "rccl" = the library calls ncclAllGather on a communicator of this object's own when
        the process group's backend is "nccl"; "torch" = the 128 bytes go through torch.distributed (what a "gloo"
        group always does)."""
        self.dev, self.rank, self.world, self.group = dev, rank, world, group
        self.always_gather = always_gather
        self.exchange = exchange
        self.exchange_used = None  # "rccl" or "torch", known after the first submission
        self.rccl_ranks = 0        # ncclCommCount of the stitch's communicator (0: no communicator of our own)
        self.last_spec = None  # the exact carry-in of the last verified run: pass it when re-submitting the same shard
        self.last_placement = None  # (index_begin, byte_base, count, bytes) of the last verified run
        self._h = None
        self._rccl = None
        self._tickets = {}

    # ---- the exchange
    def _torch_allgather(self):
        """The exchange as a callback through torch.distributed: the 128 bytes go through host memory (and, when
        the process group is RCCL, back through device tensors of torch's).  The CPU tests' and shared-GPU runs'
        exchange, and the fallback when this module cannot get an RCCL communicator of its own."""
        L = self.dev.lib
        ctx = self.dev.ctx
        group, world = self.group, self.world
        via = self.dev.device if dist.get_backend(group) == "nccl" else torch.device("cpu")

        def allgather(_comm, d_send, d_recv, nbytes, stream):
            try:
                buf = (ctypes.c_uint8 * nbytes)()
                if L.msj_copy_to_host(ctx, buf, ctypes.c_void_p(d_send), nbytes, ctypes.c_void_p(stream)) != 0:
                    return -3
                mine = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).to(via)
                gathered = torch.empty(world * nbytes, dtype=torch.uint8, device=via)
                dist.all_gather_into_tensor(gathered, mine, group=group)
                blob = gathered.cpu().numpy().tobytes()
                if L.msj_copy_to_device(ctx, ctypes.c_void_p(d_recv), blob, len(blob), ctypes.c_void_p(stream)) != 0:
                    return -3
                return 0
            except Exception:  # never let an exception cross the C frame
                return -3

        return ALLGATHER_FN(allgather)

    def _create(self):
        L = lib()
        x = MsjExchange()
        native = dist.get_backend(self.group) == "nccl" and self.exchange == "rccl"
        if native:
            # every rank must end up with the same kind of exchange: each step is agreed on before the next one
            self._rccl = _rccl_communicator(self.rank, self.world, self.dev.device, self.group)
            ok = self._rccl is not None
            if ok:
                ok = L.msj_exchange_rccl(self._rccl[1], self.rank, self.world, self._rccl[2].encode(), ctypes.byref(x)) == 0
                if not _all_ok(ok, self.dev.device, self.group):
                    if ok:
                        L.msj_exchange_release(ctypes.byref(x))  # it worked here but not everywhere: give it back
                    self._rccl[0].ncclCommDestroy(self._rccl[1])
                    self._rccl = None
                    ok = False
            if ok:
                self.rccl_ranks = self._rccl[3]
            else:
                if self.rank == 0:
                    print("mojo_simdjson_amd.sharded: falling back to the exchange through torch.distributed", file=sys.stderr)
                native = False
        if not native:
            self._cb = self._torch_allgather()  # keep the callback object alive
            x = MsjExchange()
            x.comm, x.allgather, x.rank, x.world = None, self._cb, self.rank, self.world
        self.exchange_used = "rccl" if native else "torch"
        h = ctypes.c_void_p()
        rc = L.msj_sharded_create(self.dev.ctx, ctypes.byref(x), None, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"msj_sharded_create failed: {rc}")
        self._h = h

    def close(self):
        if self._h:
            lib().msj_sharded_destroy(self._h)
            self._h = None
        if self._rccl:
            self._rccl[0].ncclCommDestroy(self._rccl[1])
            self._rccl = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def reruns(self):
        return int(lib().msj_sharded_reruns(self._h)) if self._h else 0

    @property
    def rounds(self):
        return int(lib().msj_sharded_rounds(self._h)) if self._h else 0

    def stats(self):
        """``msj_sharded_get_stats`` as a dict: cumulative results, rounds, reruns, stitch_device_ns, result_wait_ns,
        kernel_device_ns, reruns_behind_queue; last_kernel_ns / last_stitch_ns of the last completed result."""
        st = MsjShardedStats()
        if self._h:
            lib().msj_sharded_get_stats(self._h, ctypes.byref(st))
        return {k: int(getattr(st, k)) for k in STATS_FIELDS}

    def ticket_state(self, ticket):
        """``msj_sharded_ticket_state``: KERNEL_DONE | REPORTS_IN bits of a submission in flight, without waiting."""
        if "single" in ticket:
            return KERNEL_DONE | REPORTS_IN
        rc = lib().msj_sharded_ticket_state(self._h, ticket["ticket"])
        if rc < 0:
            raise RuntimeError(f"msj_sharded_ticket_state failed: {rc}")
        return rc

    # ---- speculation from local bytes only (host logic; compute once per placed shard)
    def speculate(self, has_prefix, d_shard=None, shard_len=0, d_halo=None, host_halo=None, host_head=None):
        """(in_string, next_is_escaped, prev_scalar) assumed at the shard's first byte.

        host_halo / host_head: host copies of the 64 stream bytes before the shard and of its first <= 4096
        bytes -- whoever placed the shard on the GPU had them in host memory; without them they are fetched
        from the device."""
        if not has_prefix:
            return (0, 0, 0)
        if host_halo is None:
            if d_halo is None:
                d_halo = torch.as_strided(d_shard, (64,), (1,), d_shard.storage_offset() - 64)
            host_halo = d_halo.cpu().numpy().tobytes()
        if host_head is None:
            host_head = d_shard[: min(4096, shard_len)].cpu().numpy().tobytes()
        return speculate_bytes(host_halo, host_head)

    def submit(self, d_shard, shard_len, d_idx, total_len, has_prefix, flags=0, segments=None,
               d_halo=None, host_halo=None, host_head=None, speculation=None):
        """Enqueue this rank's shard (kernel + all-gather of the reports); returns a ticket."""
        dev = self.dev
        if self.world == 1 and not self.always_gather:
            cin = dev.make_carry(0, 0, 0)
            cout = dev.new_carry()
            dev.shard(d_shard, shard_len, d_idx, cin, cout, segments=segments, has_prefix=has_prefix,
                      is_final=True, trailer_len=total_len, flags=flags)
            return dict(single=cout)
        if self._h is None:
            self._create()
        if speculation is None and (host_halo is not None or d_halo is not None):
            speculation = self.speculate(has_prefix, d_shard, shard_len, d_halo, host_halo, host_head)
        spec = None
        if speculation is not None:
            spec = MsjCarry()
            spec.in_string, spec.next_is_escaped, spec.prev_scalar = speculation
        ticket = ctypes.c_uint32(0)
        rc = lib().msj_stage1_sharded_submit(
            self._h, ctypes.c_void_p(d_shard.data_ptr()), int(shard_len),
            ctypes.c_void_p(d_idx.data_ptr()) if d_idx is not None else None, d_idx.numel() if d_idx is not None else 0,
            int(total_len), int(bool(has_prefix)), ctypes.byref(spec) if spec is not None else None,
            ctypes.c_void_p(segments.data_ptr()) if segments is not None else None,
            (segments.numel() // 32) if segments is not None else 0, dev._stream(), flags, ctypes.byref(ticket))
        if rc != 0:
            raise RuntimeError(f"msj_stage1_sharded_submit failed: {rc}")
        return dict(ticket=int(ticket.value), keep=(d_shard, d_idx, segments))

    def result(self, ticket, flags=None):
        """Wait for a submission; returns (code, total_count, local msj_carry).  The stitched offsets of this
        shard -- (index_begin, byte_base, count, bytes), ``msj_shard_placement`` -- are in ``last_placement``."""
        if "single" in ticket:
            c = self.dev.fetch(ticket["single"])
            self.last_placement = (0, 0, int(c.count), int(c.bytes))
            return int(c.code), int(c.count), c
        code, total = ctypes.c_int32(0), ctypes.c_uint64(0)
        local, used, place = MsjCarry(), MsjCarry(), MsjShardPlacement()
        rc = lib().msj_stage1_sharded_result(self._h, ticket["ticket"], ctypes.byref(code), ctypes.byref(total),
                                             ctypes.byref(local), ctypes.byref(used), ctypes.byref(place))
        if rc != 0:
            raise RuntimeError(f"msj_stage1_sharded_result failed: {rc}")
        self.last_spec = (int(used.in_string), int(used.next_is_escaped), int(used.prev_scalar))
        self.last_placement = (int(place.index_begin), int(place.byte_base), int(place.count), int(place.bytes))
        return int(code.value), int(total.value), local

    def run(self, d_shard, shard_len, d_idx, total_len, has_prefix, flags=0, segments=None,
            d_halo=None, host_halo=None, host_head=None, speculation=None):
        """Index this rank's shard.  Returns (code, total_count, local msj_carry)."""
        return self.result(self.submit(d_shard, shard_len, d_idx, total_len, has_prefix, flags=flags,
                                       segments=segments, d_halo=d_halo, host_halo=host_halo,
                                       host_head=host_head, speculation=speculation))
