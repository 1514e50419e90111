"""Device-side checker for streams made of one unit repeated.

TEST INFRASTRUCTURE (a checker, like oracle/): used by tests/ and by bench.py's verification step after the
timed window, never by the product.  The expected index array of R repetitions of a unit that ends with all
carries at zero is

    G[j] = unit_idx[j mod n] + (j div n) * unit_len,      j = 0 .. R*n - 1

(unit_idx from the oracle on ONE unit), so a shard that covers stream bytes [start, start + len) -- cut anywhere,
also inside a unit or inside a string -- must hold G[index_begin .. index_begin + count) minus its byte base,
minus the byte base of the uint32 segment an index falls into (msj_segment).  `check_shard` compares every index
on the device (the arrays never travel to the host) and returns a 64-bit hash of what the shard holds,

    h = sum over its indices of (stream_offset + 1) * (2 * j + 1)   mod 2^64,   j = the index's global ordinal,

computed from the values the library WROTE, placed with the offsets the library RETURNED (msj_shard_placement,
msj_segment): the sum of the ranks' hashes equals `stream_hash` -- a closed form of the unit's indices alone --
only if every rank's index_begin and byte_base are right as well (the reference's harness compares every index
and the trailer, tests/test_stage_1.mojo:43-82; this is that comparison for arrays of 10^10 entries).
"""
import numpy as np

MASK64 = (1 << 64) - 1


def stream_hash(unit_idx, unit_len, reps):
    """Hash of the whole stream's index array (R repetitions), from one unit's indices: a closed form mod 2^64."""
    u = np.asarray(unit_idx, dtype=np.uint64)
    n = int(u.size)
    with np.errstate(over="ignore"):
        r = np.arange(n, dtype=np.uint64)
        a = int(((u + np.uint64(1)) * (np.uint64(2) * r + np.uint64(1))).sum(dtype=np.uint64))
        b = int((u + np.uint64(1)).sum(dtype=np.uint64))
    s1 = reps * (reps - 1) // 2
    s2 = (reps - 1) * reps * (2 * reps - 1) // 6
    return (reps * a + 2 * n * b * s1 + unit_len * n * n * s1 + 2 * unit_len * n * n * s2) & MASK64


def expected_index_begin(unit_idx, unit_len, start):
    """Number of structurals of the stream in front of byte `start`."""
    u = np.asarray(unit_idx)
    return (start // unit_len) * int(u.size) + int(np.searchsorted(u, start % unit_len, side="left"))


def check_shard(torch, d_idx, count, d_unit_idx, unit_len, byte_base, index_begin, segments=None, chunk=1 << 26):
    """Compare local indices d_idx[0..count) of a shard with the stream's expected array.

    d_unit_idx: int64 device tensor, the unit's indices; byte_base / index_begin: the shard's placement as the
    library returned it; segments: list of (byte_base, index_begin, count) per uint32 segment (msj_segment rows),
    None = one segment at 0.  Returns (mismatches, hash) with hash as described in the module docstring."""
    n = int(d_unit_idx.numel())
    dev = d_idx.device
    if segments:
        seg_base = torch.tensor([int(s[0]) for s in segments], dtype=torch.int64, device=dev)
        seg_first = torch.tensor([int(s[1]) for s in segments[1:]], dtype=torch.int64, device=dev)
    bad = 0
    h = 0
    for a in range(0, int(count), chunk):
        b = min(int(count), a + chunk)
        k = torch.arange(a, b, dtype=torch.int64, device=dev)
        j = k + int(index_begin)
        q = torch.div(j, n, rounding_mode="floor")
        want = d_unit_idx[j - q * n] + q * int(unit_len)  # stream offset
        got = (d_idx[a:b].to(torch.int64) & 0xFFFFFFFF) + int(byte_base)
        if segments:
            sid = torch.bucketize(k, seg_first, right=True) if seg_first.numel() else torch.zeros_like(k)
            got = got + seg_base[sid]
        bad += int((got != want).sum().item())
        h = (h + int(((got + 1) * (2 * j + 1)).sum().item())) & MASK64  # int64 arithmetic wraps: mod 2^64
        del k, j, q, want, got
    return bad, h
