#!/usr/bin/env python3
"""GB/s against structural density d = S/N.

1 GiB streams (a 64 MiB unit repeated on the device), resident in HBM; per workload 0.3 s of untimed
launches (the clocks settle), then 50 timed ones (HIP events); the 20 launches right
after 5 warm-ups are timed too ("unsettled").  The structural count of every run is checked against the
oracle's count of the unit (checker only, outside the timed region).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import helpers  # noqa: E402
from mojo_simdjson_amd import synth  # noqa: E402
from mojo_simdjson_amd.device import Stage1Device  # noqa: E402
import bench  # noqa: E402  (same_box_ceilings: the trivial kernels of scripts/ubench/hbm_ceilings.hip on this box)

UNIT = 64 << 20
CASES = [("spaces + one scalar", lambda: synth.extreme(UNIT, 3)), ("one giant string", lambda: synth.extreme(UNIT, 2)),
         ("pretty, indent 8", lambda: synth.workload("pretty8", UNIT)), ("pretty, indent 4", lambda: synth.workload("pretty4", UNIT)),
         ("pretty, tab + CRLF", lambda: synth.workload("pretty_tab_crlf", UNIT)),
         ("pretty, indent 2", lambda: synth.workload("pretty2", UNIT)), ("UTF-8 heavy", lambda: synth.workload("utf8", UNIT)),
         ("minified", lambda: synth.workload("minified", UNIT)), ("[1234,1234,...]", lambda: synth.extreme(UNIT, 5)), ("[123,1234,123,...]", lambda: synth.extreme(UNIT, 6)),
         ("[123,123,...]", lambda: synth.extreme(UNIT, 4)), ("[12,123,12,...]", lambda: synth.extreme(UNIT, 7)),
         ("[10,10,...]", lambda: synth.extreme(UNIT, 1)), ("[[[[...]]]]", lambda: synth.extreme(UNIT, 0))]


def main():
    if len(sys.argv) > 1:  # A/B of kernel builds: tests/density_sweep.py path/to/libmsj_stage1.so
        from mojo_simdjson_amd import _lib

        _lib.LIB_PATH = os.path.abspath(sys.argv[1])
        print("library:", _lib.LIB_PATH)
    oracle = helpers.load_oracle()
    dev = Stage1Device(0)
    torch.cuda.set_device(0)
    print(f"{'workload':24s} {'bytes':>12s} {'d = S/N':>8s} {'ms':>8s} {'ingest GB/s':>12s} {'(N+4S)/t GB/s':>14s} {'frac of 8 TB/s':>14s}")
    for name, gen in CASES:
        u = gen()
        code, n_unit, _ = helpers.run_oracle(oracle.msj_oracle_stage1, u.tobytes())
        d_unit = torch.from_numpy(u).to(dev.device)
        reps = (1 << 30) // u.size
        d_buf = d_unit.repeat(reps)
        n = d_buf.numel()
        want = n_unit * reps
        ntiles = n // 4096
        wquads = min(1024, -(-(-(-4 * want // ntiles)) // 128) * 8)  # what the ceiling kernels write per tile (whole lines)
        d_idx = torch.empty(max(want + 16, ntiles * wquads * 4), dtype=torch.int32, device=dev.device)
        d_res = dev.new_carry()
        import time
        for _ in range(5):
            dev.index(d_buf, d_idx, d_res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dev.index(d_buf, d_idx, d_res)
        e1.record()
        torch.cuda.synchronize()
        ms_unsettled = e0.elapsed_time(e1) / 20
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(25):
                dev.index(d_buf, d_idx, d_res)
            torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            dev.index(d_buf, d_idx, d_res)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        res = dev.fetch(d_res)
        assert int(res.count) == want and res.internal_error == 0, (name, int(res.count), want)
        # the reference's code for the repeated stream: only meaningful for the valid documents
        alg = n + 4 * want
        # the same bytes moved by trivial kernels on this box, right behind the product's window (the index array is scratch now)
        ceil = None
        try:
            ceil = bench.same_box_ceilings(torch, dev.device, d_buf, ntiles * 4096, want, d_idx)
        except Exception as exc:
            print("ceilings:", repr(exc))
        cs = ""
        if ceil:
            cs = (f"   same-mix ceiling {ceil['same_mix_best']:.0f} GB/s (w/r {ceil['w_per_r']:.2f}; mixes "
                  f"{ceil.get('same_mix_plain_nt', 0):.0f}/{ceil.get('same_mix_nt_nt', 0):.0f}/{ceil.get('same_mix_deferred_nt_nt', 0):.0f}, "
                  f"serial sum {ceil.get('serial_sum_of_pure_streams', 0):.0f}) -> {alg / ms / 1e6 / ceil['same_mix_best']:.3f} of it")
        print(f"{name:24s} {n:12d} {want / n:8.4f} {ms:8.4f} {n / ms / 1e6:12.1f} {alg / ms / 1e6:14.1f} {alg / ms / 1e6 / 8000:14.3f}"
              f"   unsettled {ms_unsettled:.4f} ms ({alg / ms_unsettled / 1e6 / 8000:.3f}){cs}", flush=True)
        del d_buf, d_idx
    dev.close()


if __name__ == "__main__":
    main()
