"""Deterministic synthetic JSON workloads (BASELINE.json configs 2-4).

Thin wrapper over ``libmsj_gen.so`` (csrc/synth_gen.c, host-only C).  Seeds and
shapes follow SURVEY.md section 8d.
"""

import numpy as np

from . import _lib

SEED_MINIFIED = 0x5EED0001
SEED_UTF8 = 0x5EED0002
SEED_PRETTY = 0x5EED0003


def unit(target_bytes, seed=SEED_MINIFIED, mode=0, indent=0, crlf=False):
    """One complete JSON document of ~target_bytes whose length is 77 mod 128."""
    g = _lib.load_gen()
    cap = int(target_bytes) + (1 << 20)
    out = np.empty(cap, dtype=np.uint8)
    n = g.msj_gen_unit(out.ctypes.data, cap, int(target_bytes), seed, mode, indent, int(crlf))
    if n == 0:
        raise RuntimeError("synthetic generator overflowed its buffer")
    return out[:n]


def workload(name, target_bytes):
    """Named workloads used by bench.py and the parity tests."""
    if name == "minified":
        return unit(target_bytes, SEED_MINIFIED, 0, 0)
    if name == "utf8":
        return unit(target_bytes, SEED_UTF8, 1, 0)
    if name == "pretty2":
        return unit(target_bytes, SEED_PRETTY, 0, 2)
    if name == "pretty4":
        return unit(target_bytes, SEED_PRETTY, 0, 4)
    if name == "pretty8":
        return unit(target_bytes, SEED_PRETTY, 0, 8)
    if name == "pretty_tab_crlf":
        return unit(target_bytes, SEED_PRETTY, 0, -1, True)
    raise ValueError(name)


def extreme(n, kind):
    g = _lib.load_gen()
    out = np.empty(int(n), dtype=np.uint8)
    m = g.msj_gen_extreme(out.ctypes.data, int(n), kind)
    return out[:m]


def stream_shard(d_unit, unit_len, start, length):
    """Bytes [start, start + length) of the endless repetition of a device-resident unit (a torch uint8 tensor):
    how bench.py and the full-size tests place a byte-range shard of BASELINE config 5's stream on its GPU
    (start may be negative by up to one unit: the 64-byte halo in front of a shard)."""
    import torch

    parts = []
    off = start % unit_len
    remaining = length
    first = min(unit_len - off, remaining)
    parts.append(d_unit[off:off + first])
    remaining -= first
    full = remaining // unit_len
    if full:
        parts.append(d_unit.repeat(full))
    remaining -= full * unit_len
    if remaining:
        parts.append(d_unit[:remaining])
    return torch.cat(parts)
