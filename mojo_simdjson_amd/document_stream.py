"""Multi-document mode: windows over a device-resident stream of concatenated JSON documents.

The reference parses one document per call and marks streaming as to do
(``src/mojo_simdjson/generic/stage2/tape_builder.mojo:25``; the hooks of upstream simdjson's
``stage1_mode::streaming_partial`` are visible in ``generic/stage1/json_structural_indexer.mojo:153,169``).
Upstream's ``document_stream`` cuts the input into batches, indexes each batch, walks the structurals
backwards to find where the last complete document ends and starts the next batch there.  Here the
same happens with three device passes per window -- stage 1 over the window (nothing is an error yet at
its end), the token pre-pass (type byte and depth per structural), the document split (a document
starts at every depth-0 token that is not a closing bracket) -- and the host only reads two small
result structs per window.

Every window starts at a document.  Stage 1 wants a 16-byte aligned base, so the window's base is
the document's offset rounded down and the up to 15 bytes in front (the end of the previous document)
read as blanks (``MSJ_FLAG_SKIP``).  Offsets in ``d_idx`` are relative to ``Window.base``.
"""
from dataclasses import dataclass

import torch

from . import _lib, errors

MAX_WINDOW = 1 << 31


def _skip_flag(n):
    return (n & 15) << 24


class DocumentStreamError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"{message} (error {code})")
        self.code = code


@dataclass
class Window:
    base: int           # byte offset of the window in the stream (16-byte aligned)
    length: int         # bytes indexed from there
    consumed: int       # bytes that belong to this window's complete documents (the next window starts at base + consumed)
    n_tokens: int       # structurals of the complete documents
    n_documents: int    # complete documents
    utf8_error: bool    # stage 1's UTF-8 verdict for the window (informational unless strict)
    d_idx: torch.Tensor        # int32[n_tokens]: offsets relative to base
    d_type: torch.Tensor       # uint8[n_tokens]
    d_depth: torch.Tensor      # int32[n_tokens]
    d_doc_first: torch.Tensor  # int32[n_documents]: token index of each document's first token

    def document_offsets(self):
        """Absolute byte offset of every complete document (host list; reads the device arrays)."""
        first = self.d_doc_first.cpu().numpy().astype("int64")
        idx = self.d_idx.cpu().numpy().view("uint32").astype("int64")
        return [self.base + int(idx[t]) for t in first]


class DocumentStream:
    """Iterate over windows of complete documents.

    dev: Stage1Device; d_buf: uint8 device tensor (16-byte aligned) holding the stream; length: bytes;
    window: bytes indexed per step (a document must fit in one window, like upstream's batch_size);
    index_capacity: structurals a window may hold (default: one per byte up to 64 MiB windows, one per
    two bytes beyond).  The arrays a Window carries are reused by the next one.
    """

    def __init__(self, dev, d_buf, length=None, window=1 << 28, flags=0, index_capacity=None, reuse_counts=True):
        self.dev = dev
        self.d_buf = d_buf
        self.length = int(d_buf.numel() if length is None else length)
        self.window = int(min(window, MAX_WINDOW))
        if self.window < 64 or self.window % 16:
            raise ValueError("window must be a multiple of 16 bytes, at least 64")
        if d_buf.data_ptr() % 16:
            raise ValueError("the stream must be 16-byte aligned")
        self.flags = int(flags) & 3
        self.reuse_counts = bool(reuse_counts)  # MSJ_DOCS_AFTER_TOKENS: the split starts from the pre-pass's block counts
        w = min(self.window + 16, max(self.length, 16))
        if index_capacity is None:
            index_capacity = w + 3 if w <= (64 << 20) else w // 2 + 1024
        self.capacity = int(index_capacity)
        dvc = dev.device
        self._idx = torch.empty(self.capacity, dtype=torch.int32, device=dvc)
        self._type = torch.empty(self.capacity, dtype=torch.uint8, device=dvc)
        self._depth = torch.empty(self.capacity, dtype=torch.int32, device=dvc)
        self._first = torch.empty(self.capacity, dtype=torch.int32, device=dvc)
        self._zero = dev.new_carry()
        self._carry = dev.new_carry()
        self._results = torch.zeros(64, dtype=torch.uint8, device=dvc)  # msj_tokens_result | msj_documents_result
        self.windows = 0

    def __iter__(self):
        dev = self.dev
        pos = 0
        while pos < self.length:
            base = pos & ~15
            skip = pos - base
            wlen = min(self.window + skip, self.length - base)
            last = base + wlen == self.length
            d_win = self.d_buf[base:base + wlen]
            # a window is a non-final shard with zero carries: no return code, no trailer, an unclosed
            # string or a cut UTF-8 character at its end is not an error (the next window starts before it)
            dev.shard(d_win, wlen, self._idx, self._zero, self._carry, is_final=False, flags=self.flags | _skip_flag(skip))
            carry = dev.fetch(self._carry)
            if carry.internal_error:
                raise DocumentStreamError(errors.CAPACITY, f"window at {base}: more than {self.capacity} structurals")
            n = int(carry.count)
            # the two small result structs of the token pre-pass and the split come back in one read
            d_type, d_depth, _ = dev.tokens(d_win, wlen, self._idx, n, d_type=self._type, d_depth=self._depth,
                                            d_result=self._results[:24], sync=False)
            d_first, _ = dev.documents(d_win, wlen, self._idx, n, d_type, d_depth, is_final=last, d_carry=self._carry,
                                       d_doc_first=self._first, d_result=self._results[32:64], sync=False, after_tokens=self.reuse_counts)
            blob = self._results.cpu().numpy().tobytes()
            tok = _lib.MsjTokensResult.from_buffer_copy(blob[:24])
            res = _lib.MsjDocumentsResult.from_buffer_copy(blob[32:64])
            cut = res.n_complete < res.n_documents
            if carry.unescaped_error:
                raise DocumentStreamError(errors.UNESCAPED_CHARS, f"window at {base}: control character inside a string")
            if tok.min_depth < 0:
                raise DocumentStreamError(errors.TAPE_ERROR, f"window at {base}: closing bracket without an opening one")
            if last and cut:
                code = errors.UNCLOSED_STRING if carry.in_string else errors.TAPE_ERROR
                raise DocumentStreamError(code, f"the stream ends inside the document at {base + res.resume_offset}")
            if cut and res.n_complete == 0:
                raise DocumentStreamError(errors.CAPACITY, f"the document at {base + res.resume_offset} does not fit in a window of {self.window} bytes")
            self.windows += 1
            nt, nd = int(res.tokens_complete), int(res.n_complete)
            consumed = int(res.resume_offset) if cut else wlen
            yield Window(base=base, length=wlen, consumed=consumed, n_tokens=nt, n_documents=nd,
                         utf8_error=bool(carry.utf8_error), d_idx=self._idx[:nt], d_type=d_type[:nt],
                         d_depth=d_depth[:nt], d_doc_first=d_first[:nd])
            pos = base + consumed
