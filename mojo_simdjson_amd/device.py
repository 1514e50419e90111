"""Device-resident entry points (what bench.py and the GPU parity tests drive).

PyTorch is plumbing only: it owns device memory (tensors) and the HIP stream;
every byte of the hot path runs in libmsj_stage1.so's HIP kernels through the C
ABI (``msj_stage1_device`` / ``msj_stage1_shard_device``, include/msj_stage1.h).
"""
import ctypes

import torch

from . import _lib

CARRY_BYTES = 64
SEGMENT_BYTES = 32


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class Stage1Device:
    """One ``msj_ctx`` bound to one GPU (one process per GPU)."""

    def __init__(self, device_index=0):
        self.lib = _lib.load()
        if self.lib.msj_device_count() <= 0:
            raise RuntimeError("no HIP device: mojo_simdjson_amd has no CPU fallback")
        self.device_index = device_index
        self.device = torch.device("cuda", device_index)
        h = ctypes.c_void_p()
        rc = self.lib.msj_ctx_create(device_index, ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"msj_ctx_create failed: {rc}")
        self.ctx = h

    def close(self):
        if self.ctx:
            self.lib.msj_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def new_carry(self):
        return torch.zeros(CARRY_BYTES, dtype=torch.uint8, device=self.device)

    def make_carry(self, in_string=0, next_is_escaped=0, prev_scalar=0, count=0, nbytes=0):
        c = _lib.MsjCarry()
        c.in_string, c.next_is_escaped, c.prev_scalar = in_string, next_is_escaped, prev_scalar
        c.count, c.bytes = count, nbytes
        host = torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8)
        return host.to(self.device)

    def index(self, d_buf, d_idx, d_result, flags=0, length=None):
        """Enqueue stage 1 over a device-resident buffer (asynchronous).

        d_buf: uint8 CUDA tensor (16-byte aligned storage); d_idx: int32/uint32
        CUDA tensor with room for n + 3 entries; d_result: 64-byte CUDA tensor
        receiving the final ``msj_carry`` (count, code, utf8_error ...).
        """
        n = int(d_buf.numel() if length is None else length)
        rc = self.lib.msj_stage1_device(self.ctx, _ptr(d_buf), n, _ptr(d_idx), d_idx.numel(),
                                        _ptr(d_result), self._stream(), flags)
        if rc != 0:  # nothing was enqueued (argument / launch error; 1 = longer than one uint32 segment)
            raise RuntimeError(f"msj_stage1_device failed: {rc}")
        return rc

    def index_types(self, d_buf, d_idx, d_types, d_result, flags=0, length=None):
        """PROTOTYPE (``msj_stage1_types_device``): ``index`` that also writes d_types[k] = d_buf[d_idx[k]] beside every index."""
        n = int(d_buf.numel() if length is None else length)
        rc = self.lib.msj_stage1_types_device(self.ctx, _ptr(d_buf), n, _ptr(d_idx), d_idx.numel(), _ptr(d_types), _ptr(d_result),
                                              self._stream(), flags)
        if rc != 0:
            raise RuntimeError(f"msj_stage1_types_device failed: {rc}")
        return rc

    def stage2_prep_pairs(self, d_buf, length, d_idx, n, spans=True, d_prev=None, d_result=None):
        """``msj_stage2_prep_pairs_device`` (spans=True) / ``msj_tokens_pairs_device``: bracket partners as a compact list --
        d_pairs[k] = (token of the k-th opening bracket, token that closes it or 0xFFFFFFFF) -- instead of an index per
        token.  Returns (d_type, d_depth, d_pairs int32[n_cap, 2], d_end or None, d_flags or None, d_result); asynchronous:
        the number of records is msj_tokens_result.reserved."""
        n = int(n)
        dv = self.device
        d_type = torch.empty(max(n, 8), dtype=torch.uint8, device=dv)
        d_depth = torch.empty(max(n, 4), dtype=torch.int32, device=dv)
        d_pairs = torch.empty((max(n, 1), 2), dtype=torch.int32, device=dv)
        d_res = d_result if d_result is not None else torch.zeros(24, dtype=torch.uint8, device=dv)
        prev = _ptr(d_prev) if d_prev is not None else None
        if spans:
            d_end = torch.empty(max(n, 2), dtype=torch.int32, device=dv)
            d_flags = torch.empty(max(n, 2), dtype=torch.uint8, device=dv)
            rc = self.lib.msj_stage2_prep_pairs_device(self.ctx, _ptr(d_buf), int(length), _ptr(d_idx), n, _ptr(d_type), _ptr(d_depth),
                                                       _ptr(d_pairs), _ptr(d_end), _ptr(d_flags), _ptr(d_res), prev, self._stream())
        else:
            d_end = d_flags = None
            rc = self.lib.msj_tokens_pairs_device(self.ctx, _ptr(d_buf), int(length), _ptr(d_idx), n, _ptr(d_type), _ptr(d_depth),
                                                  _ptr(d_pairs), _ptr(d_res), prev, self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_*_pairs_device failed: {rc}")
        return d_type[:n], d_depth[:n], d_pairs, d_end, d_flags, d_res

    def depth_from_types(self, d_type, n, d_depth=None, d_match=None, match=False, d_result=None, d_prev=None):
        """PROTOTYPE (``msj_depth_from_types_device``): depth (and partners) of every token from type bytes stage 1 wrote.
        Asynchronous; returns (d_depth, d_match or None, d_result)."""
        n = int(n)
        if d_depth is None:
            d_depth = torch.empty(max(n, 4), dtype=torch.int32, device=self.device)
        if match and d_match is None:
            d_match = torch.empty(max(n, 4), dtype=torch.int32, device=self.device)
        d_res = d_result if d_result is not None else torch.zeros(24, dtype=torch.uint8, device=self.device)
        rc = self.lib.msj_depth_from_types_device(self.ctx, _ptr(d_type), n, _ptr(d_depth), _ptr(d_match) if d_match is not None else None,
                                                  _ptr(d_res), _ptr(d_prev) if d_prev is not None else None, self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_depth_from_types_device failed: {rc}")
        return d_depth, d_match, d_res

    def shard(self, d_buf, length, d_idx, carry_in, carry_out, segments=None, has_prefix=False,
              is_final=False, no_emit=False, trailer_len=0, flags=0):
        nseg = ctypes.c_uint32(0)
        rc = self.lib.msj_stage1_shard_device(
            self.ctx, _ptr(d_buf), int(length),
            _ptr(d_idx) if d_idx is not None else None,
            d_idx.numel() if d_idx is not None else 0,
            _ptr(carry_in), _ptr(carry_out),
            _ptr(segments) if segments is not None else None,
            (segments.numel() // SEGMENT_BYTES) if segments is not None else 0,
            ctypes.byref(nseg), int(has_prefix), int(is_final), int(no_emit), int(trailer_len),
            self._stream(), flags)
        if rc != 0:  # nothing (or not everything) was enqueued
            raise RuntimeError(f"msj_stage1_shard_device failed: {rc}")
        return rc, nseg.value

    def tokens(self, d_buf, length, d_idx, n, d_type=None, d_depth=None, d_match=None, match=False, d_result=None, sync=True,
               d_prev=None):
        """Token stream for stage 2 (``msj_tokens_device``): type byte and nesting depth of every
        structural, optionally (match=True or d_match given) the partner index of every bracket.
        Returns (d_type uint8[n], d_depth int32[n], msj_tokens_result[, d_match int32[n]]); blocking
        only for the 24-byte result (sync=False: not at all, the result stays in d_result on the device)."""
        n = int(n)
        if d_type is None:
            d_type = torch.empty(max(n, 1), dtype=torch.uint8, device=self.device)
        if d_depth is None:
            d_depth = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        if match and d_match is None:
            d_match = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        d_res = d_result if d_result is not None else torch.zeros(24, dtype=torch.uint8, device=self.device)
        # d_prev: the device msj_tokens_result of the call for the tokens in front (``msj_tokens_chain_device``)
        rc = self.lib.msj_tokens_chain_device(self.ctx, _ptr(d_buf), int(length), _ptr(d_idx), n, _ptr(d_type),
                                              _ptr(d_depth), _ptr(d_match) if d_match is not None else None,
                                              _ptr(d_res), _ptr(d_prev) if d_prev is not None else None, self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_tokens_device failed: {rc}")
        # sync=False: nothing is waited for; the third element is the device tensor holding the msj_tokens_result
        res = _lib.MsjTokensResult.from_buffer_copy(d_res.cpu().numpy().tobytes()) if sync else d_res
        if d_match is not None:
            return d_type[:n], d_depth[:n], res, d_match[:n]
        return d_type[:n], d_depth[:n], res

    def token_spans(self, d_buf, length, d_idx, n):
        """Closing quote / escape flag of every string token, end / float flag of every number token
        (``msj_token_spans_device``).  Returns (d_end int32[n] viewed as uint32, d_flags uint8[n]); asynchronous."""
        n = int(n)
        d_end = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        d_flags = torch.empty(max(n, 1), dtype=torch.uint8, device=self.device)
        rc = self.lib.msj_token_spans_device(self.ctx, _ptr(d_buf), int(length), _ptr(d_idx), n, _ptr(d_end),
                                             _ptr(d_flags), self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_token_spans_device failed: {rc}")
        return d_end[:n], d_flags[:n]

    def stage2_prep(self, d_buf, length, d_idx, n, match=False, d_prev=None, d_result=None):
        """``tokens`` and ``token_spans`` in one go (``msj_stage2_prep_device``), identical results:
        returns (d_type, d_depth, msj_tokens_result, d_match or None, d_end, d_flags).  d_prev: the device
        msj_tokens_result of the call for the tokens in front (``msj_stage2_prep_chain_device``); d_result: where
        this call's goes (to hand on as the next call's d_prev)."""
        n = int(n)
        dv = self.device
        d_type = torch.empty(max(n, 1), dtype=torch.uint8, device=dv)
        d_depth = torch.empty(max(n, 1), dtype=torch.int32, device=dv)
        d_match = torch.empty(max(n, 1), dtype=torch.int32, device=dv) if match else None
        d_end = torch.empty(max(n, 1), dtype=torch.int32, device=dv)
        d_flags = torch.empty(max(n, 1), dtype=torch.uint8, device=dv)
        d_res = d_result if d_result is not None else torch.zeros(24, dtype=torch.uint8, device=dv)
        rc = self.lib.msj_stage2_prep_chain_device(self.ctx, _ptr(d_buf), int(length), _ptr(d_idx), n, _ptr(d_type), _ptr(d_depth),
                                                   _ptr(d_match) if match else None, _ptr(d_end), _ptr(d_flags), _ptr(d_res),
                                                   _ptr(d_prev) if d_prev is not None else None, self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_stage2_prep_device failed: {rc}")
        res = _lib.MsjTokensResult.from_buffer_copy(d_res.cpu().numpy().tobytes())
        return d_type[:n], d_depth[:n], res, (d_match[:n] if match else None), d_end[:n], d_flags[:n]

    def stage2_prep_segments(self, d_buf, segments, d_idx, match=False, d_prev=None):
        """Rows f1 + f2 + f4 for a shard of several uint32 segments (``msj_stage2_prep_segments``).  segments: list of
        (byte_base, byte_len, index_begin, count) -- a host copy of the msj_segment table the shard call wrote.
        Returns (offsets, d_type, d_depth, d_match or None, d_end, d_flags, results): segment s's arrays are the slices
        [offsets[s], offsets[s] + count_s); results = one msj_tokens_result per segment (the last describes the shard)."""
        nseg = len(segments)
        table = (_lib.MsjSegment * nseg)()
        for k, (bb, bl, ib, cnt) in enumerate(segments):
            table[k].byte_base, table[k].byte_len, table[k].index_begin, table[k].count = bb, bl, ib, cnt
        total = sum(((c + 7) // 8) * 8 for _, _, _, c in segments)
        dv = self.device
        d_type = torch.empty(max(total, 8), dtype=torch.uint8, device=dv)
        d_depth = torch.empty(max(total, 8), dtype=torch.int32, device=dv)
        d_match = torch.empty(max(total, 8), dtype=torch.int32, device=dv) if match else None
        d_end = torch.empty(max(total, 8), dtype=torch.int32, device=dv)
        d_flags = torch.empty(max(total, 8), dtype=torch.uint8, device=dv)
        d_res = torch.zeros(24 * nseg, dtype=torch.uint8, device=dv)
        offs = (ctypes.c_uint64 * nseg)()
        rc = self.lib.msj_stage2_prep_segments(self.ctx, _ptr(d_buf), ctypes.byref(table), nseg, _ptr(d_idx), _ptr(d_type), _ptr(d_depth),
                                               _ptr(d_match) if match else None, _ptr(d_end), _ptr(d_flags), _ptr(d_res),
                                               _ptr(d_prev) if d_prev is not None else None, offs, self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_stage2_prep_segments failed: {rc}")
        raw = d_res.cpu().numpy().tobytes()
        results = [_lib.MsjTokensResult.from_buffer_copy(raw[24 * k: 24 * k + 24]) for k in range(nseg)]
        return [int(o) for o in offs], d_type, d_depth, d_match, d_end, d_flags, results

    def documents(self, d_buf, length, d_idx, n, d_type, d_depth, is_final=False, d_carry=None, d_doc_first=None,
                  d_result=None, sync=True, after_tokens=False):
        """Document split of one window of a stream of concatenated documents (``msj_documents_device``):
        the token index at which each document starts, and how far the complete documents reach.
        d_buf / length: the window; is_final: it ends the stream; d_type / d_depth: from ``tokens`` for the
        same d_idx; d_carry: the window's stage-1 carry_out; after_tokens: d_type / d_depth are what the last
        ``tokens`` / ``stage2_prep`` call of this device wrote, untouched since (MSJ_DOCS_AFTER_TOKENS).
        Returns (d_doc_first int32[capacity], msj_documents_result) -- blocking for the 32-byte result --
        or, with sync=False, (d_doc_first, d_result) with nothing waited for."""
        n = int(n)
        if d_doc_first is None:
            d_doc_first = torch.empty(max(n, 1), dtype=torch.int32, device=self.device)
        if d_result is None:
            d_result = torch.zeros(32, dtype=torch.uint8, device=self.device)
        rc = self.lib.msj_documents_device(self.ctx, _ptr(d_buf), int(length), int(bool(is_final)) | (2 if after_tokens else 0), _ptr(d_idx), n,
                                           _ptr(d_type),
                                           _ptr(d_depth), _ptr(d_carry) if d_carry is not None else None, _ptr(d_doc_first),
                                           d_doc_first.numel(), _ptr(d_result), self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_documents_device failed: {rc}")
        if not sync:
            return d_doc_first, d_result
        return d_doc_first, _lib.MsjDocumentsResult.from_buffer_copy(d_result.cpu().numpy().tobytes())

    def set_wait_ticks(self, ticks):
        """Test hook: bound of the single-pass kernel's inter-workgroup waits in 10 ns ticks (default 2 s)."""
        self.lib.msj_debug_set_wait_ticks(self.ctx, int(ticks))

    def fallback_count(self):
        """How often this context re-issued a call through the two-pass kernels after an expired wait."""
        return int(self.lib.msj_fallback_count(self.ctx))

    def fetch(self, d_carry):
        """Blocking read-back of a device ``msj_carry``."""
        out = _lib.MsjCarry()
        rc = self.lib.msj_carry_fetch(self.ctx, _ptr(d_carry), ctypes.byref(out), self._stream())
        if rc != 0:
            raise RuntimeError(f"msj_carry_fetch failed: {rc}")
        return out

