#!/usr/bin/env python3
"""Real-time (100 MHz) timeline of back-to-back launches: the light diagnostic build
(make -C mojo_simdjson_amd/csrc stamps_light: one real-time stamp per range, a few per workgroup).
Answers: what does a launch cost besides its steady state -- start-up, tail, the gap to the next launch?
usage: timeline.py [workload]   (MSJ_GIB=size per launch, default 1)"""
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
    _lib._share_torch_hip_runtime()
    lib = ctypes.CDLL(os.environ.get("MSJ_STAMPS_LIB", os.path.join(ROOT, "scripts", "libmsj_stage1_stamps_light.so")))
    dev = torch.device("cuda", 0)
    u = synth.workload(workload, 64 << 20)
    gib = float(os.environ.get("MSJ_GIB", "1"))
    d_buf = torch.from_numpy(u).to(dev).repeat(int(gib * (1 << 30)) // u.size)
    n = d_buf.numel()
    ntiles = (n + 4095) // 4096
    bufs = [torch.zeros((ntiles + 8192) * 16, dtype=torch.int64, device=dev) for _ in range(3)]
    d_idx = torch.empty(int(n * 0.75), dtype=torch.int32, device=dev)
    d_res = torch.zeros(64, dtype=torch.uint8, device=dev)
    h = ctypes.c_void_p()
    assert lib.msj_ctx_create(0, ctypes.byref(h)) == 0
    lib.msj_stage1_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p,
                                      ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]

    def launch(b):
        lib.msj_debug_set_stamps(ctypes.c_void_p(b.data_ptr()) if b is not None else None)
        assert lib.msj_stage1_device(h, d_buf.data_ptr(), n, d_idx.data_ptr(), d_idx.numel(), d_res.data_ptr(), None, 0) == 0

    for _ in range(6):
        launch(bufs[0])
    torch.cuda.synchronize()
    for b in bufs:
        b.zero_()
    torch.cuda.synchronize()
    for k in range(9):     # 9 back-to-back launches; the last three keep their stamps
        launch(bufs[k - 6] if k >= 6 else bufs[0])
    torch.cuda.synchronize()
    prev_end = None
    for k, b in enumerate(bufs):
        allraw = b.cpu().numpy().reshape(ntiles + 8192, 16).astype(np.int64)
        raw = allraw[:ntiles]
        wg = allraw[ntiles + 4096: ntiles + 4096 + 1100]
        wg = wg[wg[:, 0] > 0]
        k0 = wg[:, 0].min()
        us = lambda x: (x - k0) * 0.01
        workers = wg[wg[:, 3] > 0]
        end = max(workers[:, 5].max(), wg[:, 0].max())
        lo = np.arange(0, ntiles - 8, 8)
        pub = raw[lo, 8]
        pu = np.sort(us(pub[pub > 0]))
        nr = len(pu)
        marks = [pu[min(nr - 1, int(f * (nr - 1)))] for f in np.linspace(0, 1, 9)]
        mid = (marks[6] - marks[2]) * 2   # the middle half of the ranges, scaled to the whole launch
        print(f"launch {k}: {nr} ranges; gap from the previous launch's last workgroup to this one's first: "
              f"{'n/a' if prev_end is None else f'{(k0 - prev_end) * 0.01:.2f} us'}")
        print(f"   workgroups started within {us(wg[:, 0]).max():.2f} us; first bytes median {np.median(us(workers[:, 3])):.2f} us; "
              f"first publish {pu[0]:.2f} us; last publish {pu[-1]:.2f} us; last workgroup drained {us(end):.2f} us")
        print("   publishes at 0/8 .. 8/8 of the ranges (us): " + " ".join(f"{m:.1f}" for m in marks))
        print(f"   steady state (middle half x 2): {mid:.1f} us  ->  launch minus steady state: {us(end) - mid:.1f} us")
        prev_end = end
    lib.msj_ctx_destroy(h)


if __name__ == "__main__":
    main()
