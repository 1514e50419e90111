#!/usr/bin/env python3
"""Diagnostic: where a tile spends its cycles (per-phase s_memtime stamps).

Uses the -DMSJ_STAMPS build (make -C mojo_simdjson_amd/csrc stamps), loaded
directly with ctypes -- the product library never contains stamps.  Shares, not
absolute times, are what to read (the stamps themselves perturb the kernel).
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mojo_simdjson_amd import _lib, synth  # noqa: E402

TILE_ORDER = [0, 1, 2, 3, 4, 5, 6, 7]
RANGE_ORDER = [7, 8, 9, 10, 12, 13, 14, 15]  # on the wave's last tile row; 12..15 of the first tile row: inside the emission
NAMES = {0: "tile loop top", 1: "compute_tile entered", 2: "window carries",
         3: "planes+classify+escape carry", 4: "strings+scalars", 5: "utf8", 6: "count scan+ballots",
         7: "publish tile aggregate", 8: "prefix word arrived (vmcnt 0)", 9: "barrier wait",
         10: "fold + publish range aggregate + ticket request", 12: "range prefix word (poll if late)",
         13: "stage A, ticket hand-over, loads, copy A, stage B, bytes wait", 14: "copy out B",
         15: "park new tiles in LDS"}


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "minified"
    _lib._share_torch_hip_runtime()
    lib = ctypes.CDLL(os.environ.get("MSJ_STAMPS_LIB", os.path.join(ROOT, "mojo_simdjson_amd", "libmsj_stage1_stamps.so")))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    u = synth.workload(workload, 64 << 20)
    d_unit = torch.from_numpy(u).to(dev)
    d_buf = d_unit.repeat((1 << 30) // u.size)
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
    flags = int(os.environ.get("MSJ_STAMPS_FLAGS", "0"))
    for _ in range(3):
        stamps.zero_()
        rc = lib.msj_stage1_device(h, d_buf.data_ptr(), n, d_idx.data_ptr(), d_idx.numel(),
                                   d_res.data_ptr(), None, flags)
        assert rc == 0
        torch.cuda.synchronize()
    allraw = stamps.cpu().numpy().reshape(ntiles + 8192, 16).astype(np.int64)
    raw = allraw[:ntiles]
    res = allraw[ntiles:]
    okt = np.all(raw[:, TILE_ORDER] > 0, axis=1)
    okr = okt & np.all(raw[:, RANGE_ORDER] > 0, axis=1)   # the wave's last tile of a range, steady state
    print(f"workload {workload}: {ntiles} tiles, {okt.sum()} tile rows, {okr.sum()} range rows")
    tt = (raw[:, 7] - raw[:, 0])[okt]
    print(f"per tile (loop top -> aggregate published): median {np.median(tt):.0f}  mean {tt.mean():.0f}  p90 {np.percentile(tt, 90):.0f}")
    for a_, b_ in zip(TILE_ORDER[:-1], TILE_ORDER[1:]):
        d = (raw[:, b_] - raw[:, a_])[okt]
        print(f"  -> {NAMES[b_]:44s} median {np.median(d):7.0f}  mean {d.mean():8.0f}  p90 {np.percentile(d, 90):7.0f}")
    # range part; the whole iteration = 2 tiles + range part: first tile of the wave is row - 4
    first0 = np.zeros(ntiles, dtype=np.int64)
    first0[4:] = raw[:-4, 0]
    it = (raw[:, 15] - first0)
    m = okr & (first0 > 0) & (it > 0) & (it < 10**7)
    print(f"per range iteration (2 tiles): median {np.median(it[m]):.0f}  mean {it[m].mean():.0f}  p90 {np.percentile(it[m], 90):.0f}")
    for a_, b_ in zip(RANGE_ORDER[:-1], RANGE_ORDER[1:]):
        d = (raw[:, b_] - raw[:, a_])[m]
        d = d[(d >= 0) & (d < 10**6)]
        print(f"  -> {NAMES[b_]:44s} median {np.median(d):7.0f}  mean {d.mean():8.0f}  p90 {np.percentile(d, 90):7.0f}  share {100 * d.sum() / it[m].sum():5.1f} %")
    # inside the emission (stamps on the wave's first tile row of the range: row - 4)
    fr = np.zeros((ntiles, 16), dtype=np.int64); fr[4:] = raw[:-4]
    me = m & np.all(fr[:, [12, 13, 14, 15]] > 0, axis=1)
    for nm, x, y in (("prepare + stage tile A", raw[:, 12], fr[:, 12]), ("ticket hand-over + issue loads", fr[:, 12], fr[:, 13]),
                     ("copy out A", fr[:, 13], fr[:, 14]), ("prepare + stage tile B", fr[:, 14], fr[:, 15]),
                     ("wait for next range's bytes", fr[:, 15], raw[:, 13])):
        d = (y - x)[me]
        d = d[(d >= 0) & (d < 10**6)]
        print(f"     . {nm:32s} median {np.median(d):7.0f}  mean {d.mean():8.0f}  p90 {np.percentile(d, 90):7.0f}")
    ct = (raw[:, 7] - first0)[m]
    ct = ct[(ct >= 0) & (ct < 10**6)]
    print(f"  (the two computes: median {np.median(ct):.0f} mean {ct.mean():.0f}  share {100 * ct.sum() / it[m].sum():5.1f} %)")
    # ---- real-time (100 MHz) view of the resolver chain
    nch = int((res[:, 3] > 0).sum())
    rr = res[:nch]
    t0 = rr[0, 0]
    us = lambda x: (x - t0) * 0.01
    print(f"resolver: {nch} chunks of 256 ranges; first chunk entered at 0 us, last done at {us(rr[-1, 3]):.1f} us")
    print(f"  rounds per chunk: mean {rr[:, 4].mean():.2f} max {rr[:, 4].max()};  published by partial progress: mean {rr[:, 5].mean():.1f}")
    print(f"  entry -> full   : median {np.median(us(rr[:, 1]) - us(rr[:, 0])):.2f} us mean {np.mean(us(rr[:, 1]) - us(rr[:, 0])):.2f}")
    print(f"  full  -> done   : median {np.median(us(rr[:, 3]) - us(rr[:, 1])):.2f} us mean {np.mean(us(rr[:, 3]) - us(rr[:, 1])):.2f}")
    print(f"  state -> done   : median {np.median(us(rr[:, 3]) - us(rr[:, 2])):.2f} us")
    print(f"  done(c) - done(c-1): median {np.median(np.diff(us(rr[:, 3]))):.2f} us  mean {np.mean(np.diff(us(rr[:, 3]))):.2f}")
    # per range: publish time (worker) vs its chunk's full / done time vs when the owner had the prefix in hand
    lo = np.arange(0, ntiles - 8, 8)
    pub = raw[lo, 8]; got = raw[lo, 9]
    mk = (pub > 0)
    chunk = (lo // 8) // 256
    cdone = np.where(chunk < nch, rr[np.minimum(chunk, nch - 1), 3], 0)
    cfull = np.where(chunk < nch, rr[np.minimum(chunk, nch - 1), 1], 0)
    mk2 = mk & (cdone > 0)
    print(f"range publish -> its chunk full : median {np.median((cfull - pub)[mk2]) * 0.01:.2f} us  p90 {np.percentile((cfull - pub)[mk2], 90) * 0.01:.2f}")
    print(f"range publish -> its chunk done : median {np.median((cdone - pub)[mk2]) * 0.01:.2f} us  p90 {np.percentile((cdone - pub)[mk2], 90) * 0.01:.2f}")
    mk3 = mk2 & (got > 0)
    print(f"range publish -> prefix in hand : median {np.median((got - pub)[mk3]) * 0.01:.2f} us  p10 {np.percentile((got - pub)[mk3], 10) * 0.01:.2f}")
    print(f"chunk done -> prefix in hand    : median {np.median((got - cdone)[mk3]) * 0.01:.2f} us  p10 {np.percentile((got - cdone)[mk3], 10) * 0.01:.2f}  frac<1us {np.mean((got - cdone)[mk3] < 100):.3f}")
    smp = raw[lo, 10]; rdy = raw[lo, 11]
    mk4 = mk3 & (smp > 0) & (rdy > 0)
    print(f"first sample ready: {np.mean(rdy[mk4] > 1):.3f} of ranges;  publish -> sample: median {np.median((smp - pub)[mk4]) * 0.01:.2f} us;  chunk done -> sample: median {np.median((smp - cdone)[mk4]) * 0.01:.2f} us")
    nr = mk4 & (rdy == 1)
    if nr.any():
        print(f"  not ready at sample ({nr.sum()}): chunk done -> sample median {np.median((smp - cdone)[nr]) * 0.01:.2f} us p90 {np.percentile((smp - cdone)[nr], 90) * 0.01:.2f};  sample -> in hand median {np.median((got - smp)[nr]) * 0.01:.2f} us")
        print(f"  position in chunk of the not-ready ones: mean {np.mean(((lo // 8) % 256)[nr]):.1f}")
    # publish times relative to range order: how far out of order do ranges publish
    order_lag = (pub[mk] - np.maximum.accumulate(pub[mk]))
    print(f"publish time behind the running max of earlier ranges: median {np.median(order_lag) * 0.01:.2f} us  p1 {np.percentile(order_lag, 1) * 0.01:.2f} us")
    cm = np.maximum.accumulate(pub[mk])
    print(f"kernel real-time span by publishes: {(pub[mk].max() - pub[mk].min()) * 0.01:.1f} us")
    # ---- start-up and tail per workgroup (real time)
    wg = allraw[ntiles + 4096: ntiles + 4096 + 1100]
    wg = wg[wg[:, 0] > 0]
    k0 = wg[:, 0].min()
    u = lambda x: (x - k0) * 0.01
    workers = wg[wg[:, 3] > 0]
    print(f"workgroups: {len(wg)} started within {u(wg[:, 0]).max():.2f} us of the first")
    for nm, c in (("prologue done", 2), ("first bytes in registers", 3), ("last range computed", 4), ("drained", 5)):
        col = workers[:, c]
        col = col[col > 0]
        if col.size == 0:
            continue
        print(f"  {nm:26s} median {np.median(u(col)):8.2f} us   min {u(col).min():8.2f}   max {u(col).max():8.2f}")
    lib.msj_ctx_destroy(h)


if __name__ == "__main__":
    main()
