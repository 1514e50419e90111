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

PHASES = ["window carries", "planes+classify+escape", "strings+scalars", "utf8", "count scan",
          "waits+publish agg", "deferred (2 tiles)", "prefix word", "emit (LDS staged)"]


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "minified"
    _lib._share_torch_hip_runtime()
    lib = ctypes.CDLL(os.path.join(ROOT, "mojo_simdjson_amd", "libmsj_stage1_stamps.so"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    u = synth.workload(workload, 64 << 20)
    d_unit = torch.from_numpy(u).to(dev)
    d_buf = d_unit.repeat((1 << 30) // u.size)
    n = d_buf.numel()
    ntiles = (n + 4095) // 4096
    stamps = torch.zeros(ntiles * 16, dtype=torch.int64, device=dev)
    d_idx = torch.empty(int(n * 0.75), dtype=torch.int32, device=dev)
    d_res = torch.zeros(64, dtype=torch.uint8, device=dev)
    h = ctypes.c_void_p()
    assert lib.msj_ctx_create(0, ctypes.byref(h)) == 0
    lib.msj_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    lib.msj_stage1_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    for _ in range(3):
        rc = lib.msj_stage1_device(h, d_buf.data_ptr(), n, d_idx.data_ptr(), d_idx.numel(),
                                   d_res.data_ptr(), None, 0)
        assert rc == 0
        torch.cuda.synchronize()
    raw = stamps.cpu().numpy().reshape(ntiles, 16).astype(np.int64)
    ok = raw[:, 0] > 0
    print(f"loop top -> prefetch issued : median {np.median((raw[:, 11] - raw[:, 0])[ok]):.0f}")
    print(f"prefetch issued -> compute  : median {np.median((raw[:, 1] - raw[:, 11])[ok]):.0f}")
    okp = (raw[:, 15] > 0) & (raw[:, 8] > 0)
    print(f"range-prefix wait (cycles): median {np.median((raw[:, 8] - raw[:, 15])[okp]):.0f} mean {np.mean((raw[:, 8] - raw[:, 15])[okp]):.0f}")
    print(f"publish -> emit slot reached (cycles): median {np.median((raw[:, 15] - raw[:, 7])[okp]):.0f}")
    w = (raw[:, 8] - raw[:, 15])[okp]
    print("range-prefix wait percentiles (cycles):", {q: int(np.percentile(w, q)) for q in (50, 75, 90, 95, 99, 99.9)})
    print(f"sum of waits / sum of lifetimes: {w.clip(0, 10**9).sum() / max(1, (raw[:, 10] - raw[:, 1])[okp].clip(0, 10**9).sum()):.3f}")
    tt = raw[:, 14][okp]; tt = (tt - tt.min()) * 10e-3
    for lo in range(0, int(tt.max()) + 1, 100):
        sel = (tt >= lo) & (tt < lo + 100)
        if sel.any():
            print(f"  t=[{lo:5d},{lo+100:5d}) us: tiles {sel.sum():6d} median wait {np.median(w[sel]):8.0f} mean {w[sel].clip(0,10**9).mean():10.0f}")
    rt = raw[:, 12:15]
    okr = (rt[:, 0] > 0) & (rt[:, 1] > 0) & (rt[:, 2] > 0)
    lat = (rt[:, 1] - rt[:, 0])[okr] * 10e-3  # 100 MHz ticks -> us
    slack = (rt[:, 2] - rt[:, 1])[okr] * 10e-3
    print(f"resolver latency publish->prefix: median {np.median(lat):.2f} us  p90 {np.percentile(lat, 90):.2f}  max {lat.max():.2f}  (n={okr.sum()})")
    print(f"slack prefix->needed            : median {np.median(slack):.2f} us  p10 {np.percentile(slack, 10):.2f}  frac<0 {np.mean(slack < 0):.3f}")
    span_rt = (rt[:, 2][okr].max() - rt[:, 0][okr].min()) * 10e-3
    print(f"real-time span {span_rt:.1f} us")
    # iteration accounting for the first tile of each range-wave pair (t and t+4 belong to the same wave)
    t0 = raw[:, 0]
    per = t0[4:] - t0[:-4]
    okk = (t0[4:] > 0) & (t0[:-4] > 0) & (per > 0) & (per < 10**7) & ((np.arange(len(per)) % 8) < 4)
    print(f"iteration period (cycles): median {np.median(per[okk]):.0f}")
    def seg(a, b, src=raw):
        d = src[:, b] - src[:, a]
        m = (src[:, a] > 0) & (src[:, b] > 0) & (d >= 0) & (d < 10**7)
        return np.median(d[m]) if m.any() else float('nan')
    print(f"  issue loads 0->11 {seg(0,11):.0f} | compute 1->6 {seg(1,6):.0f} | wait+publish 6->7 {seg(6,7):.0f}")
    s = raw[:, 1:11]
    d = np.diff(s, axis=1)
    life = s[:, 9] - s[:, 0]
    print(f"workload {workload}: {ntiles} tiles; tile lifetime median {np.median(life):.0f} ticks, "
          f"p90 {np.percentile(life, 90):.0f}")
    span = s[:, 9].max() - s[:, 0].min()
    print(f"kernel span {span} ticks; sum of lifetimes / span = {life.sum() / span:.1f} tiles in flight")
    for k, name in enumerate(PHASES):
        print(f"  {name:20s} median {np.median(d[:, k]):9.0f}  mean {d[:, k].mean():9.0f}  "
              f"share {100 * d[:, k].sum() / life.sum():5.1f} %")
    lib.msj_ctx_destroy(h)


if __name__ == "__main__":
    main()
