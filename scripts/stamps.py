#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of the worker loop (s_memtime stamps, -DMSJ_STAMPS build:
make -C mojo_simdjson_amd/csrc stamps).  Shares, not absolute times, are what to read.

Rows of the stamp buffer are tiles.  Per tile (row = the tile): 0 loop top, 1 compute entered, 2 window carries,
3 planes + classes + escape ballots, 4 strings + scalars + errors, 5 utf8, 6 counts + scan, 7 aggregate published.
Per range and wave (row = the wave's SECOND tile of the range): 8 both tiles computed, 9 barrier passed, 10 folded +
range aggregate published, 12 next range known + loads issued + old prefix in hand, 13 tile A staged, 14 tile A
stored, 15 tile B staged, 11 next range's bytes arrived; the iteration ends at the next iteration's stamp 0.
usage: stamps.py [workload] [flags]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mojo_simdjson_amd import _lib, synth  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "minified"
    flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    _lib._share_torch_hip_runtime()
    lib = ctypes.CDLL(os.environ.get("MSJ_STAMPS_LIB", os.path.join(ROOT, "scripts", "libmsj_stage1_stamps.so")))
    dev = torch.device("cuda", 0)
    u = synth.workload(workload, 64 << 20)
    gib = float(os.environ.get('MSJ_GIB', '1'))
    d_buf = torch.from_numpy(u).to(dev).repeat(int(gib * (1 << 30)) // u.size)
    n = d_buf.numel()
    ntiles = (n + 4095) // 4096
    stamps = torch.zeros((ntiles + 8192) * 16, dtype=torch.int64, device=dev)
    d_idx = torch.empty(int(n * 0.75), dtype=torch.int32, device=dev)
    d_res = torch.zeros(64, dtype=torch.uint8, device=dev)
    h = ctypes.c_void_p()
    assert lib.msj_ctx_create(0, ctypes.byref(h)) == 0
    lib.msj_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    lib.msj_stage1_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    for _ in range(3):
        stamps.zero_()
        assert lib.msj_stage1_device(h, d_buf.data_ptr(), n, d_idx.data_ptr(), d_idx.numel(), d_res.data_ptr(), None, flags) == 0
        torch.cuda.synchronize()
    allraw = stamps.cpu().numpy().reshape(ntiles + 8192, 16).astype(np.int64)
    raw = allraw[:ntiles]
    okt = np.all(raw[:, :8] > 0, axis=1)
    print(f"workload {workload} flags {flags}: {ntiles} tiles, {okt.sum()} stamped")
    names = {1: "loop top -> compute entered", 2: "window carries", 3: "planes + classes + escape ballots",
             4: "strings + scalars + errors", 5: "utf8", 6: "counts + scan", 7: "publish tile aggregate"}
    tot = (raw[:, 7] - raw[:, 0])[okt]
    print(f"per tile: median {np.median(tot):.0f} mean {tot.mean():.0f}")
    for k in range(1, 8):
        d = (raw[:, k] - raw[:, k - 1])[okt]
        print(f"   {names[k]:36s} median {np.median(d):6.0f} mean {d.mean():7.0f}")
    # range rows: the wave's second tile; first tile of the same wave = row - 4; next iteration top = any later row's 0
    second = np.zeros(ntiles, dtype=bool)
    second[(np.arange(ntiles) % 8) >= 4] = True
    A0 = np.zeros(ntiles, dtype=np.int64)
    A0[4:] = raw[:-4, 0]
    okr = second & okt & np.all(raw[:, [8, 9, 10, 11, 12, 13, 14, 15]] > 0, axis=1) & (A0 > 0)
    seq = [("compute tile A + B", A0, raw[:, 8]), ("barrier wait", raw[:, 8], raw[:, 9]), ("ticket hand-off + fold + publish", raw[:, 9], raw[:, 10]),
           ("hand-off wait + loads + prefix", raw[:, 10], raw[:, 12]), ("prepare + stage A", raw[:, 12], raw[:, 13]),
           ("copy out A", raw[:, 13], raw[:, 14]), ("prepare + stage B", raw[:, 14], raw[:, 15]),
           ("wait for next bytes", raw[:, 15], raw[:, 11])]
    span = (raw[:, 11] - A0)[okr]
    print(f"per range iteration up to the bytes wait ({okr.sum()} rows): median {np.median(span):.0f} mean {span.mean():.0f}  (+ copy out B + park, not stamped)")
    for nm, x, y in seq:
        d = (y - x)[okr]
        d = d[(d >= 0) & (d < 10**6)]
        print(f"   {nm:36s} median {np.median(d):6.0f} mean {d.mean():7.0f}  share {100 * d.sum() / span.sum():5.1f} %")
    # ---- real-time (100 MHz) view of the launch: start-up, tail, resolver chain
    wg = allraw[ntiles + 4096: ntiles + 4096 + 1100]
    wg = wg[wg[:, 0] > 0]
    k0 = wg[:, 0].min()
    us = lambda x: (x - k0) * 0.01
    workers = wg[wg[:, 3] > 0]
    print(f"workgroups: {len(wg)} started within {us(wg[:, 0]).max():.2f} us of the first")
    for nm, c in (("prologue done", 2), ("first bytes in registers", 3), ("last range computed", 4), ("drained", 5)):
        col = workers[:, c]
        col = col[col > 0]
        if col.size:
            print(f"  {nm:26s} median {np.median(us(col)):8.2f} us   min {us(col).min():8.2f}   p10 {np.percentile(us(col), 10):8.2f}  p90 {np.percentile(us(col), 90):8.2f}  max {us(col).max():8.2f}")
    res = allraw[ntiles:ntiles + 4096]  # resolver chunk rows
    nch = int((res[:, 3] > 0).sum())
    if nch:
        rr = res[:nch]
        print(f"resolver: {nch} chunks; first entered {us(rr[0, 0]):.2f} us, first done {us(rr[0, 3]):.2f}, last full {us(rr[-1, 1]):.2f}, last done {us(rr[-1, 3]):.2f} us")
        q = [0, nch // 4, nch // 2, 3 * nch // 4, nch - 1]
        print("  chunk done at (us): " + "  ".join(f"#{i}: {us(rr[i, 3]):.1f} (full {us(rr[i, 1]):.1f})" for i in q))
    # range publish times (real time) along the launch
    lo = np.arange(0, ntiles - 8, 8)
    pub = raw[lo, 8]
    mk = pub > 0
    if mk.any():
        pu = us(pub[mk])
        idx = np.arange(lo.size)[mk]
        for f in [k / 16 for k in range(17)] + [0.99]:
            i = min(len(pu) - 1, int(f * (len(pu) - 1)))
            print(f"  range {idx[i]:6d} ({100 * f:5.1f} %) aggregate published at {pu[i]:8.2f} us")
        print(f"  all publishes within [{pu.min():.2f}, {pu.max():.2f}] us")
    lib.msj_ctx_destroy(h)


if __name__ == "__main__":
    main()
