#!/usr/bin/env python3
"""what a BALANCED split of a lane's 64-bit structural mask between its two scatter chains
would save (stage1_kernel.hip, scatter_bits32: a chain takes two indices per step -- the lowest and the highest set
bit -- and runs until the lane with the most bits is done; a tile's emission is chain(low halves) + chain(high halves)).

Exact count on the real workloads' masks (the oracle's indices of a 16 MiB unit; CPU only, uses oracle/: a checker-side
tool like tests/density_sweep.py), per 4 KiB tile = per wave:
  now       max over lanes ceil(lo / 2) + max over lanes ceil(hi / 2)
  swap      the larger half of every lane in the first chain, the smaller in the second
  balanced  a lane's bits split at its median bit: ceil(ceil(n / 2) / 2) + ceil(floor(n / 2) / 2), max over lanes each
  perfect   every lane with the same number of indices: ceil(S / 128)
A step is 8 vector instructions (profiles/r04/valu_sections_utf8.txt: emission 168 per tile minified = 15 steps x 8 + 48).
    python3 tests/emission_balance.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import synth  # noqa: E402
from tests import helpers  # noqa: E402


def main():
    f = helpers.load_oracle_fast()
    print(f"{'workload':10s} {'idx/tile':>8s} {'now':>6s} {'swap':>6s} {'balanced':>8s} {'perfect':>8s}   balanced saves (steps, x 8 = vector instructions per tile)")
    for name in ("minified", "utf8", "pretty4", "pretty8"):
        data = synth.workload(name, 16 << 20).tobytes()
        idx = np.zeros(len(data) + 3, dtype=np.uint32)
        nn = ctypes.c_uint64(0)
        assert f.msj_fast_stage1(data, len(data), idx.ctypes.data, idx.size, ctypes.byref(nn)) == 0
        idx = idx[: nn.value]
        nt = len(data) // 4096
        h = np.bincount(idx // 32, minlength=nt * 128)[: nt * 128].reshape(nt, 64, 2)  # tile, lane, half
        lo, hi = h[:, :, 0], h[:, :, 1]
        n = lo + hi
        half = lambda x: (x + 1) // 2  # noqa: E731
        now = half(lo).max(1) + half(hi).max(1)
        swap = half(np.maximum(lo, hi)).max(1) + half(np.minimum(lo, hi)).max(1)
        bal = half((n + 1) // 2).max(1) + half(n // 2).max(1)
        perfect = (n.sum(1) + 127) // 128
        d = now.mean() - bal.mean()
        print(f"{name:10s} {len(idx) / nt:8.0f} {now.mean():6.2f} {swap.mean():6.2f} {bal.mean():8.2f} {perfect.mean():8.2f}   {d:.2f} steps = {8 * d:.0f}")
    print("(finding a lane's median set bit is a select-by-rank in a 64-bit word: 5 popcount-and-halve rounds of ~4 vector\n"
          " instructions, and the two chains would no longer work on aligned 32-bit words: the split costs what it saves)")


if __name__ == "__main__":
    main()
