#!/usr/bin/env python3
"""Rate of the multi-document mode (row f3) on 1 GiB of NDJSON: the document split alone, and the
windowed DocumentStream end to end (stage 1 + token pre-pass + split per window, host reads included)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd.device import Stage1Device  # noqa: E402
from mojo_simdjson_amd.document_stream import DocumentStream  # noqa: E402

dev = Stage1Device(0)
torch.cuda.set_device(0)
block = b"".join(json.dumps({"id": i, "text": "t" * (i % 50), "tags": [i, i + 1], "user": {"name": "n", "ok": True}},
                            separators=(",", ":")).encode() + b"\n" for i in range(12000))
nrep = (1 << 30) // len(block)
d_buf = torch.from_numpy(np.frombuffer(block, dtype=np.uint8).copy()).to(dev.device).repeat(nrep)
nbytes = d_buf.numel()
d_idx = torch.empty(int(nbytes * 0.4), dtype=torch.int32, device=dev.device)
cin, cout = dev.new_carry(), dev.new_carry()
dev.shard(d_buf, nbytes, d_idx, cin, cout, is_final=False)
n = int(dev.fetch(cout).count)
t, d, tok = dev.tokens(d_buf, nbytes, d_idx, n)
d_first = torch.empty(12000 * nrep + 16, dtype=torch.int32, device=dev.device)
d_res = torch.zeros(32, dtype=torch.uint8, device=dev.device)
for _ in range(3):
    dev.documents(d_buf, nbytes, d_idx, n, t, d, is_final=True, d_carry=cout, d_doc_first=d_first, d_result=d_res, sync=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    dev.documents(d_buf, nbytes, d_idx, n, t, d, is_final=True, d_carry=cout, d_doc_first=d_first, d_result=d_res, sync=False)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
_, res = dev.documents(d_buf, nbytes, d_idx, n, t, d, is_final=True, d_carry=cout, d_doc_first=d_first)
alg = 2 * 5 * n + 4 * res.n_documents  # type + depth read by the count and by the write pass, one uint32 per document
print(f"document split: {nbytes} B, {n} structurals, {res.n_documents} documents ({res.n_complete} complete): {ms:.3f} ms, "
      f"{n / ms / 1e6:.1f} G structurals/s, {alg / ms / 1e6:.0f} GB/s of its own traffic, {nbytes / ms / 1e6:.0f} GB/s of JSON")
del d_idx, t, d, d_first
for window in (64 << 20, 256 << 20, 1 << 30):
    best = {}
    for rep in range(6):
        reuse = rep % 2 == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        docs = 0
        stream = DocumentStream(dev, d_buf, nbytes, window=window, index_capacity=int((window + 16) * 0.4), reuse_counts=reuse)
        for w in stream:
            docs += w.n_documents
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep >= 2:
            best[reuse] = min(best.get(reuse, 1e9), dt)
    dt = best[True]
    print(f"DocumentStream window {window >> 20:5d} MiB: {stream.windows} windows, {docs} documents, {dt * 1e3:.2f} ms, "
          f"{nbytes / dt / 1e9:.0f} GB/s of JSON, {docs / dt / 1e6:.0f} M documents/s "
          f"({best[False] * 1e3:.2f} ms when the split counts the document starts itself)")
dev.close()
