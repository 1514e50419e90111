#!/usr/bin/env python3
"""bench.py -- stage-1 GB/s of JSON ingested on N MI355X (BASELINE.json metric).

A "step" is one full stage-1 pass (structural indexing + UTF-8 verdict) over the
rank's device-resident shard of the synthetic stream:

  N = 1 : BASELINE.json configs[1], 1 GiB minified ASCII twitter-like JSON
          (a ~64 MiB generated unit repeated; exact size printed in `config`).
  N > 1 : BASELINE.json configs[4]'s per-rank share: the same stream grown to
          N x 8 GiB (64 GiB at N = 8), cut into N byte-range shards that start
          and end anywhere inside a unit (weak scaling; `--gib-per-gpu` overrides);
          every step includes the RCCL stitch (sharded.py: one all-gather of 128
          bytes per rank, the library calls ncclAllGather itself).

`python bench.py --gpus N` without a launcher environment starts the N ranks itself
(torch.distributed.run as a CHILD process, before anything here touches a GPU) and
relays their output and exit code; under the driver's launcher (WORLD_SIZE set) it
is one of the ranks.

After the timed window every rank verifies its own index array ON THE DEVICE against
the replication property (tests/replication.py: every index, placed with the offsets
the stitch returned) and contributes a 64-bit hash to one all-gather; rank 0 compares
the sum with the closed form of the oracle's unit indices: `config.verified`.

Output: ONE JSON line on rank 0 (contract in the task statement), with
`roofline` for the dominant kernel (stage1_kernel) and `cpu_baseline` (the
oracle's block-for-block restatement of the reference, 1 thread, bounded sample).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mojo_simdjson_amd import synth  # noqa: E402
from mojo_simdjson_amd.device import Stage1Device  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
UNIT_BYTES = 64 << 20
SHARD_ALIGN = 16384  # shard bases are multiples of this (four of the kernel's 4 KiB tiles: 16-byte aligned bases)


def unit_indices(unit):
    """The oracle's structural indices of one unit (checker only: what the GPU result is verified against after
    the timed window).  oracle/stage1_fast.c gives the port's results bit for bit (tests/test_oracle_fast.py) in a
    fraction of its time, and every rank needs them."""
    import ctypes

    from tests import helpers

    f = helpers.load_oracle_fast()
    data = unit.tobytes()
    idx = np.zeros(len(data) + 3, dtype=np.uint32)
    nn = ctypes.c_uint64(0)
    rc = f.msj_fast_stage1(data, len(data), idx.ctypes.data, idx.size, ctypes.byref(nn))
    assert rc == 0, f"the oracle rejects the synthetic unit: {rc}"
    return idx[: int(nn.value)].copy()


def cpu_baseline(unit, budget_s=10.0):
    """Reference-faithful CPU port (oracle/stage1_oracle.c, 1 thread) on a bounded sample: the timed baseline.
    (bench.py touches oracle/ here and in unit_indices(), nowhere else, and never inside a timed window.)"""
    import ctypes

    from tests import helpers

    o = helpers.load_oracle()
    o.msj_oracle_stage1_repeat.restype = ctypes.c_int32
    o.msj_oracle_stage1_repeat.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_void_p,
                                           ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                           ctypes.c_int]
    data = unit.tobytes()
    idx = np.zeros(len(data) + 3, dtype=np.uint32)
    n = ctypes.c_uint64(0)
    probe = data[: 8 << 20]
    t0 = time.perf_counter()
    o.msj_oracle_stage1_repeat(probe, len(probe), idx.ctypes.data, idx.size, ctypes.byref(n), 1)
    t_probe = time.perf_counter() - t0
    rate = len(probe) / t_probe
    reps = max(1, int(budget_s * rate / len(data)))
    t0 = time.perf_counter()
    rc = o.msj_oracle_stage1_repeat(data, len(data), idx.ctypes.data, idx.size, ctypes.byref(n), reps)
    dt = time.perf_counter() - t0
    assert rc == 0
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {
        "value": round(reps * len(data) / dt / 1e9, 4),
        "unit": "GB/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} x {len(data)} B unit of the same workload ({dt:.1f} s), host CPU {model}, "
                  f"{os.cpu_count()} logical cores present",
    }
    # two more CPU figures, labelled as NOT the reference -- the same
    # algorithm with carry-less-multiply prefix_xor and AVX2 classification (oracle/stage1_fast.c),
    # on one thread and on all cores (two-pass chunked run); same results as the port
    try:
        f = helpers.load_oracle_fast()
        nn = ctypes.c_uint64(0)

        def best(fn, runs):
            ts = []
            for _ in range(runs):
                t0 = time.perf_counter()
                rc = fn()
                ts.append(time.perf_counter() - t0)
                assert rc == 0 and nn.value == n.value
            return min(ts)

        t1 = best(lambda: f.msj_fast_stage1(data, len(data), idx.ctypes.data, idx.size, ctypes.byref(nn)), 3)
        # all cores: a larger sample (4 units) and the best thread count of a small sweep -- the
        # job's CPU share on the GPU box is a cgroup quota, not what cpu_count() says
        big = data * 4
        idx4 = np.zeros(len(big) + 3, dtype=np.uint32)
        n4 = ctypes.c_uint64(0)
        tm, threads = None, 1
        for t in (8, 16, 32, 64, 128):
            if t > (os.cpu_count() or 1):
                break
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                rc = f.msj_fast_stage1_mt(big, len(big), idx4.ctypes.data, idx4.size, ctypes.byref(n4), t)
                ts.append(time.perf_counter() - t0)
                assert rc == 0 and n4.value == 4 * n.value
            if tm is None or min(ts) < tm:
                tm, threads = min(ts), t
        tm = tm / 4 if tm else t1  # per unit, like t1
        out["not_the_reference"] = {
            "optimised_1_thread": {"value": round(len(data) / t1 / 1e9, 3), "unit": "GB/s", "cores": 1},
            "optimised_all_cores": {"value": round(len(data) / tm / 1e9, 3), "unit": "GB/s", "cores": threads},
            "note": "oracle/stage1_fast.c: pclmul prefix_xor + AVX2 classify, identical results; best of 3 runs "
                    "(1 thread: one unit; all cores: 4 units, two-pass chunked scan, best of 8..128 threads)",
        }
    except Exception as exc:  # measurement extra: never fails the bench
        out["not_the_reference"] = {"error": repr(exc)}
    return out


def end_to_end(unit, lib, n_bytes=256 << 20, runs=5):
    """"also report (clearly separated) end-to-end including H2D".  msj_stage1 -- the host-pointer
    entry point the Mojo shim binds -- on `n_bytes` of the same workload in ordinary pageable host memory: PCIe up,
    kernel, indices down over PCIe, all inside the call.  NEVER the metric: a separate record beside it."""
    import ctypes

    reps = max(1, n_bytes // int(unit.size))
    data = np.tile(unit, reps)
    idx = np.zeros(data.size + 3, dtype=np.uint32)  # touched once, reused: the parser's list (allocate(), :85-89)
    n, verdict = ctypes.c_uint64(0), ctypes.c_int32(0)
    ts = []
    for k in range(runs + 1):
        t0 = time.perf_counter()
        rc = lib.msj_stage1(data.ctypes.data, data.size, idx.ctypes.data, idx.size, ctypes.byref(n), ctypes.byref(verdict), 0)
        ts.append(time.perf_counter() - t0)
        assert rc == 0, rc
    ts = sorted(ts[1:])  # the first call sets the pinned rings up
    med = ts[len(ts) // 2]
    # where the host side ran (28.6 GB/s on the driver's box against 43 - 45 on the builder's, nothing in
    # the record to say why): the GPU's NUMA node, the node of the caller's two buffers and of the library's pinned rings,
    # how many copy workers are bound to the GPU's node, the PCIe link (msj_host_placement; -1 / "" = the kernel does not say)
    placement = None
    try:
        lib.msj_host_placement.restype = ctypes.c_int32
        lib.msj_host_placement.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64]
        lib.msj_debug_numa_node_of.restype = ctypes.c_int32
        lib.msj_debug_numa_node_of.argtypes = [ctypes.c_void_p]
        text = ctypes.create_string_buffer(1024)
        if lib.msj_host_placement(None, text, 1024) == 0:
            placement = json.loads(text.value.decode())
            placement["caller_input_node"] = int(lib.msj_debug_numa_node_of(data.ctypes.data + data.size // 2))
            placement["caller_index_node"] = int(lib.msj_debug_numa_node_of(idx.ctypes.data + idx.nbytes // 4))
            placement["cpus_allowed"] = len(os.sched_getaffinity(0))
    except Exception as exc:  # a measurement extra
        placement = {"error": repr(exc)}
    return {"value": round(data.size / med / 1e9, 2), "unit": "GB/s", "best": round(data.size / ts[0] / 1e9, 2),
            "ms_median": round(med * 1e3, 3), "bytes": int(data.size), "structurals": int(n.value), "runs": runs,
            "placement": placement,
            "what": "msj_stage1 (host pointers, pageable caller memory): H2D of the input + kernel + D2H of n + 3 indices per "
                    "call, median of the runs; NOT the metric (inputs of `value` are resident in HBM)"}


def same_box_ceilings(torch_mod, device, d_in, n_bytes, structurals, d_scratch, grid=None, brief=False):
    """What the memory system of THIS box gives this launch's bytes when nothing is computed: the trivial kernels of
    scripts/ubench/hbm_ceilings.hip (built as scripts/bin/libhbm_ceilings.so by __graft_entry__.build()) on the
    product kernel's persistent grid, the product's tile walk and store shape, in the same process right behind the
    product's windows (boxes of the pool differ by 3-5 %: a ceiling recorded on another box can sit below the product).
    Reads the shard once per launch; writes ceil(4 S / tiles) bytes per tile, rounded UP to whole 128-byte lines.
    Returns GB/s per variant (settled: 300 untimed launches, then the faster of two windows of 300 timed ones) or None.
    `same_mix_best` is the fastest way found to move the launch's bytes without computing anything: the best of four
    trivial read + write kernels and of (best pure read time + best pure write time).
    `best_trivial_kernel` is the fastest trivial kernel that EXISTS for these bytes (no serial-sum model).
    brief=True (the sub-records of `other_configs`): non-temporal read + the two non-temporal mixes, 150 launches per
    window -- a kernel that exists for the row, in a second of box time."""
    import ctypes

    path = os.path.join(ROOT, "scripts", "bin", "libhbm_ceilings.so")
    if not os.path.exists(path):
        return None
    L = ctypes.CDLL(path)
    L.msj_ceiling_launch.restype = ctypes.c_int
    L.msj_ceiling_launch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32,
                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
    ntiles = n_bytes // 4096
    wquads = min(1024, -(-(-(-4 * structurals // ntiles)) // 128) * 8)  # bytes per tile -> whole lines -> 16-byte quads
    if d_scratch.numel() * d_scratch.element_size() < ntiles * wquads * 16:
        return None
    grid = grid or 4 * torch_mod.cuda.get_device_properties(device).multi_processor_count
    sink = torch_mod.zeros(64, dtype=torch_mod.uint8, device=device)
    stream = ctypes.c_void_p(torch_mod.cuda.current_stream(device).cuda_stream)
    nl = 150 if brief else 300
    out = {"w_per_r": round(wquads * 16 / 4096, 4), "grid": grid, "launches": nl}
    n_rd, n_wr = ntiles * 4096, ntiles * wquads * 16

    def ms(wq, policy):
        def go(reps):
            rc = L.msj_ceiling_launch(d_in.data_ptr(), d_scratch.data_ptr(), sink.data_ptr(), ntiles, wq, policy, grid, reps, stream)
            assert rc == 0, rc
        go(nl)
        best = None
        for _ in range(2):  # a ceiling is the BEST the box does: the faster of two windows of `nl` launches
            e0, e1 = torch_mod.cuda.Event(enable_timing=True), torch_mod.cuda.Event(enable_timing=True)
            e0.record()
            go(nl)
            e1.record()
            torch_mod.cuda.synchronize()
            t = e0.elapsed_time(e1) / nl
            best = t if best is None else min(best, t)
        return best

    def gbps(nbytes, t_ms):
        return round(nbytes / (t_ms * 1e-3) / 1e9, 1)

    if brief:
        out["read_nt"] = gbps(n_rd, ms(0, 1))
        if wquads:
            t_mix = {"same_mix_nt_nt": ms(wquads, 1), "same_mix_deferred_nt_nt": ms(wquads, 9)}
            for k, t in t_mix.items():
                out[k] = gbps(n_rd + n_wr, t)
            out["best_trivial_kernel"] = gbps(n_rd + n_wr, min(t_mix.values()))
        else:
            out["best_trivial_kernel"] = out["read_nt"]
        return out
    t_read = {"read_plain": ms(0, 0), "read_nt": ms(0, 1)}
    for k, t in t_read.items():
        out[k] = gbps(n_rd, t)
    if wquads:
        # the same bytes by trivial kernels: load next / store this (the product's policies: plain loads, non-temporal
        # whole-line stores; and with non-temporal loads), stores deferred by two ranges like the product's emission
        t_mix = {"same_mix_plain_nt": ms(wquads, 0), "same_mix_nt_nt": ms(wquads, 1),
                 "same_mix_deferred_plain_nt": ms(wquads, 8), "same_mix_deferred_nt_nt": ms(wquads, 9)}
        t_wr = {"write_only_nt": ms(wquads, 4), "write_only_plain": ms(wquads, 6)}
        for k, t in t_mix.items():
            out[k] = gbps(n_rd + n_wr, t)
        for k, t in t_wr.items():
            out[k] = gbps(n_wr, t)
        # HBM's data bus is half duplex: a launch cannot take less than its reads at the best pure read rate plus its
        # writes at the best pure write rate.  The write-only launches of this launch's own (small) output are short
        # enough for their start and tail to count (0.16 ms; the product pays one start and tail, not two), so the pure
        # write rate is also taken from a launch that writes 3 N (the whole scratch buffer) and the better one counts
        wq_big = min(1024, (d_scratch.numel() * d_scratch.element_size() // ntiles // 16) // 8 * 8)
        peak = 0.0
        for wq_p in sorted({256, wq_big}):  # 1 N and as much as the scratch buffer holds (3 N in bench.py)
            if wq_p > wquads and wq_p <= wq_big:
                rate_p = gbps(ntiles * wq_p * 16, min(ms(wq_p, 4), ms(wq_p, 6)))
                if rate_p > peak:
                    peak = rate_p
                    out["write_only_peak_w_per_r"] = round(wq_p * 16 / 4096, 4)
        if peak:
            out["write_only_peak"] = peak
        wr_rate = max(out.get("write_only_peak", 0.0), out["write_only_nt"], out["write_only_plain"])
        t_sum = min(t_read.values()) + n_wr / (wr_rate * 1e9) * 1e3
        out["serial_sum_of_pure_streams"] = gbps(n_rd + n_wr, t_sum)
        out["same_mix_ms"] = round(min(min(t_mix.values()), t_sum), 4)
        out["same_mix_best"] = gbps(n_rd + n_wr, min(min(t_mix.values()), t_sum))
        out["best_trivial_kernel"] = gbps(n_rd + n_wr, min(t_mix.values()))  # a kernel that exists (no model)
    else:
        out["same_mix_best"] = max(out["read_plain"], out["read_nt"])
        out["best_trivial_kernel"] = out["same_mix_best"]
    return out


# BASELINE.json configs 3 and 4 (and config 4's synthetic extremes) as sub-records of the default N = 1 line: the same
# protocol as the headline (W warm-up steps, the K steps right behind them as `unsettled`, `settle_ms` of untimed
# passes, K timed steps between synchronisations, HIP events on the launch stream for the kernel time), every index
# verified on the device afterwards against the oracle's unit indices (checker only, outside every timed window).
OTHER_CONFIGS = [
    ("utf8", "BASELINE config 3: 1 GiB UTF-8-heavy JSON (multi-byte code points + escaped strings), UTF-8 validation on",
     lambda: synth.workload("utf8", UNIT_BYTES)),
    # the same stream as the REFERENCE computes it: its UTF-8 checker is a stub that always succeeds
    # (generic/stage1/json_structural_indexer.mojo:16-30), so MSJ_FLAG_NO_UTF8 is the reference's semantics; the headline
    # and `utf8` above keep the validator on (more work than the reference does)
    ("utf8_validator_off", "BASELINE config 3 with MSJ_FLAG_NO_UTF8 (the reference validates nothing: its checker is a stub)",
     lambda: synth.workload("utf8", UNIT_BYTES), 2),
    ("pretty4", "BASELINE config 4: 1 GiB pretty-printed JSON, indent 4",
     lambda: synth.workload("pretty4", UNIT_BYTES)),
    ("d0_blanks", "config 4 extreme d ~ 0: blanks + one scalar per 64 MiB unit",
     lambda: synth.extreme(UNIT_BYTES - 52, 3)),
    ("d0.5_[123,", "config 4 extreme d = 0.5: [123,123,...]",
     lambda: synth.extreme(UNIT_BYTES - 52, 4)),
    ("d1.0_[[[[", "config 4 extreme d = 1.0: [[[[...]]]]",
     lambda: synth.extreme(UNIT_BYTES - 52, 0)),
]


def other_config_record(dev, device, name, what, gen, steps, warmup, settle_ms, flags, with_ceilings):
    """One sub-record of `other_configs` (N = 1 only): a 1 GiB stream of the workload's unit, resident in HBM."""
    from tests import replication

    unit = gen()
    unit_len = int(unit.size)
    u_idx = unit_indices(unit)  # the oracle's indices of one unit: the expected result (checker only)
    unit_n = int(u_idx.size)
    d_unit = torch.from_numpy(unit).to(device)
    reps = (1 << 30) // unit_len
    d_buf = d_unit.repeat(reps)
    total = reps * unit_len
    d_idx = torch.empty(unit_n * reps + 1024, dtype=torch.int32, device=device)
    d_res = dev.new_carry()
    assert d_buf.data_ptr() % 16 == 0

    def step():
        dev.index(d_buf, d_idx, d_res, flags=flags, length=total)

    def window(k):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(k):
            step()
        e1.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, e0.elapsed_time(e1)

    for _ in range(warmup):
        step()
    u_dt, u_ev = window(steps)
    settle_steps = 0
    t_end = time.perf_counter() + settle_ms * 1e-3
    while settle_ms > 0 and time.perf_counter() < t_end:
        for _ in range(25):
            step()
        settle_steps += 25
        torch.cuda.synchronize()
    dt, ev = window(steps)
    res = dev.fetch(d_res)
    count = int(res.count)
    alg = total + 4 * count
    k_ms = ev / steps
    achieved = alg / (k_ms * 1e-3) / 1e9
    # ---- every index, on the device (tests/replication.py), + the trailer + the closed-form hash
    d_uidx = torch.from_numpy(u_idx.astype(np.int64)).to(device)
    bad, h = replication.check_shard(torch, d_idx, min(count, unit_n * reps), d_uidx, unit_len, 0, 0, None)
    tail = (d_idx[count:count + 3].to(torch.int64) & 0xFFFFFFFF).tolist() if count + 3 <= d_idx.numel() else None
    ok = (int(res.code) == 0 and res.internal_error == 0 and count == unit_n * reps and bad == 0
          and h == replication.stream_hash(u_idx, unit_len, reps) and tail == [total & 0xFFFFFFFF, total & 0xFFFFFFFF, 0]
          and int(res.bytes) == total)
    verified = "indices" if ok else (f"FAILED (code {int(res.code)}, count {count} want {unit_n * reps}, mismatches {bad}, "
                                     f"trailer {tail})")
    del d_uidx
    rec = {
        "workload": f"1 GiB/GPU synthetic {name} (unit {unit_len} B x {reps}); {what}",
        "bytes_total": total, "structurals_total": count, "density": round(count / total, 5),
        "utf8_validation": not (flags & 2),
        "ms_per_step": round(dt / steps * 1e3, 4),
        "value": round(total * steps / dt / 1e9, 2), "unit": "GB/s",
        "untimed_passes_before_window": warmup + steps + settle_steps,
        "verified": verified,
        "utf8_error": int(res.utf8_error),
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "frac_unsettled": round(alg / (u_ev / steps * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "kernel": "stage1_kernel", "kernel_ms": round(k_ms, 4), "kernel_ms_unsettled": round(u_ev / steps, 4),
                     "algorithmic_bytes_per_launch": alg, "traffic": None},
    }
    if with_ceilings:
        try:
            ntiles = total // 4096
            wquads = min(1024, -(-(-(-4 * count // ntiles)) // 128) * 8)
            need = ntiles * wquads * 16
            d_scratch = d_idx if d_idx.numel() * 4 >= need else torch.empty(need, dtype=torch.uint8, device=device)
            c = same_box_ceilings(torch, device, d_buf, ntiles * 4096, count, d_scratch, brief=True)
            if c:
                rec["roofline"]["ceilings"] = c
                rec["roofline"]["frac_of_measured_read"] = round(achieved / c["read_nt"], 4)
                rec["roofline"]["frac_of_best_trivial_kernel"] = round(achieved / c["best_trivial_kernel"], 4)
        except Exception as exc:  # a measurement extra: never fails the bench
            rec["roofline"]["ceilings"] = {"error": repr(exc)}
    del d_buf, d_idx, d_unit
    torch.cuda.empty_cache()
    return rec, ok


def ceiling_fields(achieved, ingest, box, recorded, workload):
    """The roofline record's measured-ceiling fields (GB/s): from this box's own run when there is one."""
    src = None
    if isinstance(box, dict) and "read_nt" in box:
        c, src = box, "this box, this process (scripts/ubench/hbm_ceilings.hip, 300 settled launches per variant)"
    else:
        r = (recorded or {}).get("by_workload", {}).get(workload)
        c = dict(r) if isinstance(r, dict) else None
        if c:
            c.update({k: recorded[k] for k in ("read_plain", "read_nt") if k in recorded})
            src = f"profiles/traffic.json, recorded {recorded.get('date')} on another box of the pool (boxes differ by 3-5 %)"
    if not c:
        return {"ceilings": box if isinstance(box, dict) else None, "measured_read_peak": None, "frac_of_measured_read": None,
                "ingest_frac_of_measured_read": None, "model_envelope": None, "frac_of_model_envelope": None,
                "best_trivial_kernel": None, "frac_of_best_trivial_kernel": None}
    read = max(c["read_plain"], c["read_nt"])
    # `model_envelope` = max(the trivial mixes, the SERIAL SUM of the best pure read and the best pure write): the second
    # term is a model (a half-duplex data bus), not a kernel; `best_trivial_kernel` is the fastest kernel that exists
    mix = c["same_mix_best"]
    best = c.get("best_trivial_kernel")
    return {"ceilings": {**c, "source": src},
            "measured_read_peak": read,
            "frac_of_measured_read": round(achieved / read, 4),
            "ingest_frac_of_measured_read": round(ingest / read, 4),
            "model_envelope": mix,
            "frac_of_model_envelope": round(achieved / mix, 4),
            "best_trivial_kernel": best,
            "frac_of_best_trivial_kernel": round(achieved / best, 4) if best else None}


def self_launch(args):
    """`python bench.py --gpus N` from a plain command line: start the N ranks as a child process tree
    (torch.distributed.run, one process per GPU) BEFORE anything in this process touches a GPU, relay their
    output (rank 0 prints the JSON line) and leave with their exit code.  Never an exec: a process that has
    initialised the GPU must not be replaced, and this one has not even done that."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="minified",
                    choices=["minified", "utf8", "pretty2", "pretty4", "pretty8", "pretty_tab_crlf"])
    ap.add_argument("--gib-per-gpu", type=float, default=None,
                    help="stream bytes per GPU; default 1 (BASELINE config 2) at N = 1, 8 (config 5: 64 GiB over 8 GPUs) at N > 1")
    ap.add_argument("--lib", default=None,
                    help="measurement aid: load this build of libmsj_stage1.so instead of the package's (A/B of kernel "
                         "variants in one GPU session); reported as config.library")
    ap.add_argument("--no-verify", action="store_true", help="skip the index-by-index verification after the timed window")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-utf8", action="store_true", help="skip UTF-8 validation (not the headline config)")
    ap.add_argument("--settle-ms", type=float, default=400.0,
                    help="untimed back-to-back passes (this many ms of them) between the W warm-up steps and the K timed "
                         "steps, so that the timed window sees the clocks the GPU holds under load instead of the "
                         "transient after idle (DESIGN.md section 4); 0 = none.  The K steps right after the warm-up "
                         "are timed too and reported as 'unsettled'")
    ap.add_argument("--rehearse-sharded", action="store_true",
                    help="N = 1 only: run the N > 1 code path -- process group over RCCL with one rank, msj_stage1_sharded_submit / "
                         "_result with the library's own ncclAllGather, three submissions in flight, the per-rank fields of the "
                         "line -- on one GPU (a rehearsal of the line the driver gets at N > 1, never the headline)")
    ap.add_argument("--index-capacity-frac", type=float, default=0.0,
                    help="index slots per input byte (default: what the workload needs, from the unit's own count)")
    ap.add_argument("--no-ceilings", action="store_true",
                    help="skip the trivial HBM-ceiling kernels behind the verification (profiling passes: 6 000 launches less)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the sub-records of BASELINE configs 3 and 4 (`other_configs`; N = 1, default workload only)")
    ap.add_argument("--no-emit", action="store_true",
                    help="diagnostic: summary pass only, no index writes (never a reported number)")
    args = ap.parse_args()
    if args.gib_per_gpu is None:
        args.gib_per_gpu = 1.0 if args.gpus == 1 else 8.0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))  # nothing above or in the imports has touched a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.lib:
        from mojo_simdjson_amd import _lib

        _lib.LIB_PATH = os.path.abspath(args.lib)
    # one process per GPU; MSJ_BENCH_BACKEND=gloo lets several ranks share one GPU to
    # rehearse the N>1 path on a one-GPU box (never used for reported numbers)
    backend = os.environ.get("MSJ_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    sharded = world > 1 or args.rehearse_sharded  # the N > 1 code path (always at N > 1)
    if sharded:
        import torch.distributed as dist_mod

        dist = dist_mod
        if world == 1:  # --rehearse-sharded from a plain command line: a one-rank group
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    dev = Stage1Device(dev_index)
    flags = 2 if args.no_utf8 else 0

    # ---- synthetic stream: one generated unit, repeated (identical on every rank)
    seeds = {"minified": (synth.SEED_MINIFIED, 0, 0, False), "utf8": (synth.SEED_UTF8, 1, 0, False),
             "pretty2": (synth.SEED_PRETTY, 0, 2, False), "pretty4": (synth.SEED_PRETTY, 0, 4, False),
             "pretty8": (synth.SEED_PRETTY, 0, 8, False), "pretty_tab_crlf": (synth.SEED_PRETTY, 0, -1, True)}
    seed, mode, indent, crlf = seeds[args.workload]
    unit = synth.unit(UNIT_BYTES, seed, mode, indent, crlf)
    unit_len = int(unit.size)
    d_unit = torch.from_numpy(unit).to(device)
    per_gpu = int(args.gib_per_gpu * (1 << 30))
    total_len = (world * per_gpu // unit_len) * unit_len       # whole units: a valid document stream
    shard_len_nominal = -(-total_len // world)
    shard_len_nominal = -(-shard_len_nominal // SHARD_ALIGN) * SHARD_ALIGN
    start = rank * shard_len_nominal
    shard_len = max(0, min(total_len, start + shard_len_nominal) - start)
    assert shard_len > 0
    halo = 64 if rank > 0 else 0
    d_alloc = synth.stream_shard(d_unit, unit_len, start - halo, shard_len + halo)
    d_shard = d_alloc[halo:]
    assert d_shard.data_ptr() % 16 == 0

    # ---- expected result: the oracle's indices of ONE unit (checker only, outside every timed window)
    u_idx = unit_indices(unit)
    unit_n = int(u_idx.size)
    e2e = None
    cpu = None  # the timed CPU baseline runs BEHIND the GPU windows (below): ten seconds of it in front of them would
    #             put the GPU into a deeper idle state than anything a caller's process does before its first parse

    # index slots for this shard: from the unit's own count (a shard holds at most ceil(shard_len / unit_len) + 1 units'
    # worth of structurals), never from an assumed density
    cap = (-(-shard_len // unit_len) + 1) * unit_n + 1024
    if args.index_capacity_frac:  # measurement aid: where the index buffer ends up relative to the input matters a little
        cap = max(cap, int(shard_len * args.index_capacity_frac) + 1024)
    d_idx = torch.empty(cap, dtype=torch.int32, device=device)
    d_res = dev.new_carry()
    n_seg_max = 8
    d_seg = torch.zeros(n_seg_max * 32, dtype=torch.uint8, device=device)  # msj_segment table of the shard
    sh = None

    if not sharded and args.no_emit:
        d_zero = dev.new_carry()

        def step():
            dev.shard(d_shard, shard_len, d_idx, d_zero, d_res, is_final=True, no_emit=True, trailer_len=total_len,
                      flags=flags)
            return None
    elif not sharded and shard_len <= 0xFFFFFFFF:
        def step():
            dev.index(d_shard, d_idx, d_res, flags=flags, length=shard_len)
            return None
    elif not sharded:
        # longer than one uint32 segment: the shard entry point chains segments on the device
        d_zero = dev.new_carry()

        def step():
            dev.shard(d_shard, shard_len, d_idx, d_zero, d_res, segments=d_seg, is_final=True, trailer_len=total_len,
                      flags=flags)
            return None
    else:
        from mojo_simdjson_amd.sharded import ShardedStage1

        sh = ShardedStage1(dev, rank, world, always_gather=(world == 1))
        # the host that placed the shard had these bytes in host memory (they are part of
        # what it copied to the GPU): the 64-byte halo and the shard's first 4 KiB
        host_halo = d_alloc[:halo].cpu().numpy().tobytes() if halo else None
        host_head = d_shard[:4096].cpu().numpy().tobytes() if halo else None

        # speculative carries from the shard's own bytes: host logic, once per placed shard
        spec = sh.speculate(rank > 0, d_shard, shard_len, host_halo=host_halo, host_head=host_head)

        def submit():
            # after the first verified run the exact carry-in of this (unchanged) shard is known: a
            # refuted guess costs one re-run once, not one per step
            return sh.submit(d_shard, shard_len, d_idx, total_len, has_prefix=(rank > 0), flags=flags,
                             segments=d_seg, speculation=sh.last_spec or spec)

        def step():
            return sh.result(submit())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    per_step = []  # N > 1: (kernel-only ns, kernel end -> reports in ns) of every result, from the slot's HIP events

    def collect(ticket):
        r = sh.result(ticket)
        st = sh.stats()
        per_step.append((st["last_kernel_ns"], st["last_stitch_ns"]))
        return r

    def timed_window():
        """EXACTLY K steps between two barriers; returns (wall seconds, HIP-event ms, last result)."""
        last = None
        per_step.clear()
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        if not sharded:
            for _ in range(args.steps):
                last = step()
        else:
            # steps are independent passes: up to DEPTH submissions in flight.  result(k) waits for the event behind
            # ITS read-back only (csrc/sharded.cpp), and the all-gather + read-back of step k sit on the library's
            # high-priority side stream behind kernel k's event: kernel k+1 starts behind kernel k, the stitch of
            # step k and the host's verification run beside it.  All K results are in hand before the closing barrier
            pending = []
            for _ in range(args.steps):
                pending.append(submit())
                if len(pending) >= sh.DEPTH:
                    last = collect(pending.pop(0))
            while pending:
                last = collect(pending.pop(0))
        ev1.record()
        barrier()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1), last

    last = None
    for _ in range(args.warmup):
        last = step()
    unsettled = None
    settle_steps = 0
    if args.settle_ms > 0:
        # (a) the K steps as they run right after W warm-up steps on a GPU that was idle: reported, not `value`
        u_dt, u_ev, _ = timed_window()
        unsettled = (u_dt, u_ev)
        # (b) keep the GPU under the same load until its clocks have settled (untimed)
        t_end = time.perf_counter() + args.settle_ms * 1e-3
        while True:
            for _ in range(25):
                last = step()
            settle_steps += 25
            torch.cuda.synchronize()
            flag = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], dtype=torch.float64,
                                device=device if (dist is None or backend == "nccl") else "cpu")
            if dist is not None:
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # every rank leaves in the same round
            if float(flag.item()) == 0.0:
                break
    stats0 = sh.stats() if sh is not None else None
    dt, ev_ms, last_timed = timed_window()
    stats1 = sh.stats() if sh is not None else None
    if last_timed is not None:
        last = last_timed
    timed_per_step = list(per_step)

    # ---- "median / min of >= 10 runs".  A second window right behind the timed one (same
    #      clocks), every step bracketed by its own pair of HIP events; `ms_per_step` stays the K-step window's mean
    def quantiles(ms):
        a = np.sort(np.asarray(ms, dtype=np.float64))
        return {"n": int(a.size), "median": round(float(np.median(a)), 4), "min": round(float(a[0]), 4),
                "p95": round(float(a[min(a.size - 1, int(np.ceil(0.95 * a.size)) - 1)]), 4),
                "max": round(float(a[-1]), 4), "mean": round(float(a.mean()), 4)}

    step_dist = None
    standalone = None
    if not sharded:
        n_dist = max(args.steps, 50)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_dist + 1)]
        evs[0].record()
        for k in range(n_dist):
            step()
            evs[k + 1].record()
        torch.cuda.synchronize()
        step_dist = quantiles([evs[k].elapsed_time(evs[k + 1]) for k in range(n_dist)])
    else:
        # this rank's shard WITHOUT the stitch (the exact carry its last result proved, no exchange, no host turn):
        # what the rank's GPU does alone -- the basis of `scaling_efficiency` below
        s_in, e_in, ps_in = sh.last_spec
        d_exact = dev.make_carry(s_in, e_in, ps_in)
        d_alone = dev.new_carry()

        def alone():
            dev.shard(d_shard, shard_len, d_idx, d_exact, d_alone, segments=d_seg, has_prefix=(rank > 0),
                      is_final=(rank == world - 1), trailer_len=total_len, flags=flags)

        for _ in range(2):
            alone()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            alone()
        e1.record()
        torch.cuda.synchronize()
        standalone = e0.elapsed_time(e1) / args.steps  # ms per pass of this rank's shard, nothing else on the GPU
        barrier()
    # ---- the box's own ceilings for this launch's bytes (N = 1), right behind the product's windows while the GPU is
    #      still under load (behind the ten idle seconds of the CPU leg the same trivial kernels read 4 % less), into a
    #      scratch buffer of their own (the index array is verified later)
    ceil_box = None
    if world == 1 and rank == 0 and not args.no_emit and not args.no_ceilings and not args.rehearse_sharded:
        try:
            n_c = min(shard_len, 0xFFFF0000) // 4096 * 4096
            d_scratch = torch.empty(3 * n_c + 4096, dtype=torch.uint8, device=device)
            torch.cuda.synchronize()
            res_c = dev.fetch(d_res)
            ceil_box = same_box_ceilings(torch, device, d_shard, n_c, int(int(res_c.count) * (n_c / shard_len)), d_scratch)
            del d_scratch
        except Exception as exc:  # a measurement extra: never fails the bench
            ceil_box = {"error": repr(exc)}
    # ---- BASELINE configs 3 and 4 under the same clock (N = 1, default workload at 1 GiB only): sub-records, each with its
    #      own warm-up, unsettled window, settle and K-step window; behind the headline's windows, before the CPU leg
    others, others_failed = None, False
    if (world == 1 and rank == 0 and not sharded and args.workload == "minified" and args.gib_per_gpu == 1.0
            and not (args.no_other_configs or args.no_emit or args.no_verify)):
        others = {}
        for name, what, gen, *extra in OTHER_CONFIGS:
            try:
                rec, ok = other_config_record(dev, device, name, what, gen, args.steps, args.warmup, args.settle_ms,
                                              flags | (extra[0] if extra else 0), with_ceilings=not args.no_ceilings)
                others[name] = rec
                others_failed = others_failed or not ok
            except Exception as exc:  # the headline is still printed; the exit code says that a sub-record broke
                others[name] = {"error": repr(exc)}
                others_failed = True
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.rehearse_sharded:
        torch.cuda.synchronize()
        cpu = cpu_baseline(unit)  # timed CPU baseline on rank 0 at N = 1 only; the GPU is idle meanwhile
        try:
            e2e = end_to_end(unit, dev.lib)
        except Exception as exc:  # a measurement extra: never fails the bench
            e2e = {"error": repr(exc)}

    # ---- result check (outside the timed region)
    if not sharded:
        res = dev.fetch(d_res)
        code, total_count = int(res.code), int(res.count)
        assert res.internal_error == 0
        assert int(res.bytes) == shard_len and total_count > 0, (int(res.bytes), shard_len, total_count)
        placement = (0, 0, total_count, shard_len)
    else:
        code, total_count, res = last
        placement = sh.last_placement
    reps = total_len // unit_len
    # (wrong results do not raise here: they go into the printed line as a failed verification, exit code 3)
    count_ok = code == 0 and total_count == unit_n * reps
    local_count = int(res.count) if sharded else total_count

    # ---- every index of this rank's shard, on the device, placed with the offsets the stitch returned
    #      (tests/replication.py; SURVEY.md section 8d config 5: device-side checker + 64-bit hash)
    verified, verify_detail = "count", None
    if not (args.no_verify or args.no_emit):
        from tests import replication

        torch.cuda.synchronize()
        index_begin, byte_base, cnt, nbytes = placement
        # nothing in here raises before the all-gather below: a rank that found something wrong still takes part
        place_ok = (byte_base, cnt, nbytes) == (start, local_count, shard_len)
        segs = None
        if sharded or shard_len > 0xFFFFFFFF:
            nseg = -(-shard_len // 0xFFFF0000)
            table = np.frombuffer(d_seg.cpu().numpy().tobytes(), dtype=np.uint64).reshape(n_seg_max, 4)[:nseg]
            segs = [(int(r[0]), int(r[2]), int(r[3])) for r in table]
            place_ok = place_ok and sum(sg[2] for sg in segs) == local_count
        d_uidx = torch.from_numpy(u_idx.astype(np.int64)).to(device)
        bad, h = replication.check_shard(torch, d_idx, local_count, d_uidx, unit_len, start, index_begin, segs)
        want_begin = replication.expected_index_begin(u_idx, unit_len, start)
        tail_ok = True
        if rank == world - 1:
            tail = (d_idx[local_count:local_count + 3].to(torch.int64) & 0xFFFFFFFF).tolist()
            tail_ok = tail == [total_len & 0xFFFFFFFF, total_len & 0xFFFFFFFF, 0]
        # one all-gather of (mismatches, hash, index_begin ok, trailer ok) per rank
        mine = torch.tensor([bad, h - (1 << 64) if h >= (1 << 63) else h, int(index_begin == want_begin and place_ok), int(tail_ok)],
                            dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        if dist is not None:
            allv = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
        else:
            allv = [mine]
        rows = [[int(x) for x in v.tolist()] for v in allv]
        h_sum = sum(r[1] for r in rows) & replication.MASK64
        h_want = replication.stream_hash(u_idx, unit_len, reps)
        ok = count_ok and all(r[0] == 0 and r[2] == 1 and r[3] == 1 for r in rows) and h_sum == h_want
        verify_detail = {"mismatches": sum(r[0] for r in rows), "hash": f"{h_sum:016x}", "hash_expected": f"{h_want:016x}",
                         "index_begin_ok": all(r[2] == 1 for r in rows), "trailer_ok": all(r[3] == 1 for r in rows)}
        # a failed verification does not swallow the measurement: the line is printed with the verdict in it and the
        # process then exits with code 3
        verified = "indices" if ok else (f"FAILED (code {code}, count {total_count} want {unit_n * reps}; per rank [mismatches, hash, "
                                         f"placement ok, trailer ok]: {rows})")
        del d_uidx
    elif not count_ok:
        verified = f"FAILED (code {code}, count {total_count} want {unit_n * reps})"

    t = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())
    stitch, scaling_eff = None, None
    if sh is not None:
        # the stitch over the timed window, max over ranks where it is a latency
        st = {k: stats1[k] - stats0[k] for k in stats1}
        v = torch.tensor([st["reruns"], st["rounds"], st["stitch_device_ns"] / max(1, st["rounds"]),
                          st["result_wait_ns"] / max(1, st["results"])], dtype=torch.float64,
                         device=device if backend == "nccl" else "cpu")
        vmax = v.clone()
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        # per step and rank: the kernel alone (event pair around the shard's launches) and kernel end -> reports in.
        # The rank whose kernel ends last sees the bare latency of the exchange, every other rank that + its lead:
        # the spread of the second figure over the ranks of one step is the skew of the ranks' kernel ends
        ps = torch.tensor(timed_per_step if len(timed_per_step) == args.steps else [(0, 0)] * args.steps,
                          dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        alone_t = torch.tensor([standalone, float(shard_len)], dtype=torch.float64, device=ps.device)
        all_ps = [torch.empty_like(ps) for _ in range(world)]
        all_alone = [torch.empty_like(alone_t) for _ in range(world)]
        dist.all_gather(all_ps, ps)
        dist.all_gather(all_alone, alone_t)
        kern = np.stack([t.cpu().numpy()[:, 0] for t in all_ps]) / 1e6    # [rank, step] ms
        arrive = np.stack([t.cpu().numpy()[:, 1] for t in all_ps]) / 1e3  # [rank, step] us
        skew = arrive.max(axis=0) - arrive.min(axis=0)
        alone_ms = [float(t[0].item()) for t in all_alone]
        alone_rate = sum(float(t[1].item()) / (float(t[0].item()) * 1e-3) for t in all_alone) / 1e9  # GB/s, all ranks
        stitch = {"exchange": sh.exchange_used, "rccl_ranks": sh.rccl_ranks, "reruns": int(v[0].item()),
                  "reruns_behind_queue": int(stats1["reruns_behind_queue"] - stats0["reruns_behind_queue"]),
                  "allgather_rounds": int(vmax[1].item()),
                  "in_flight": sh.DEPTH,
                  "stitch_us_per_round": round(float(vmax[2].item()) / 1e3, 2),
                  "result_wait_us_per_step": round(float(vmax[3].item()) / 1e3, 2),
                  "kernel_only_ms": [round(float(x), 4) for x in np.median(kern, axis=1)],
                  "kernel_only_ms_max_step": round(float(kern.max()), 4),
                  "reports_in_us": {"min_rank_median": round(float(np.median(arrive, axis=1).min()), 2),
                                    "max_rank_median": round(float(np.median(arrive, axis=1).max()), 2)},
                  "rank_skew_us": {"median": round(float(np.median(skew)), 2), "max": round(float(skew.max()), 2)},
                  "standalone_ms": [round(x, 4) for x in alone_ms],
                  "note": "timed window only; reruns summed over ranks; stitch = HIP-event time from the end of a round's "
                          "kernel to the gathered reports' arrival in pinned host memory (max over ranks); result_wait = host "
                          "time blocked in msj_stage1_sharded_result per step (max over ranks, with up to 3 steps in flight); "
                          "kernel_only_ms = per rank, median over the steps, HIP events around the shard's launches alone; "
                          "reports_in_us = kernel end -> reports in, per rank (median over steps): the smallest is the "
                          "exchange's bare latency; rank_skew_us = per step, max - min of that figure over the ranks = how "
                          "far apart the ranks' kernels end; standalone_ms = each rank's shard through msj_stage1_shard_device "
                          "with its exact carry, no exchange, no host turn (K steps after the timed window)"}
        scaling_eff = {"value": round(total_len * args.steps / dt_max / 1e9 / alone_rate, 4),
                       "gib_per_gpu": args.gib_per_gpu,
                       "basis": f"whole-job GB/s / sum over the {world} ranks of (shard bytes / standalone_ms): the same "
                                f"{args.gib_per_gpu:g} GiB/GPU shards on the same GPUs in the same process without the stitch "
                                f"(= {world} x the N = 1 configuration AT {args.gib_per_gpu:g} GiB PER GPU, measured here rather "
                                f"than assumed).  NOT comparable with the default N = 1 line, which is BASELINE config 2 at "
                                f"1 GiB per GPU: a launch of 1 GiB pays its ~11 us of start-up and tail 8 x as often per byte "
                                f"as one of 8 GiB (`python bench.py --gpus 1 --gib-per-gpu {args.gib_per_gpu:g}` is the N = 1 "
                                f"point of this curve)",
                       "standalone_aggregate": round(alone_rate, 2)}

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        value = total_len * args.steps / dt_max / 1e9
        # dominant kernel: stage1_kernel, one launch per step on this rank's stream
        # (N>1: one launch per shard unless a speculation was refuted); HIP-event time
        # over the timed region / steps, which also covers the small per-launch
        # descriptor memset and, for N>1, the host side of the all-gather.
        launches = 1
        alg_bytes = shard_len + 4 * local_count
        k_ms = ev_ms / args.steps / launches
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes kept in profiles/traffic.json -- only if they were taken with
        # THIS kernel (the source hash in msj_version()) on this workload and size; otherwise null
        traffic, traffic_note, measured = None, None, {}
        version = dev.lib.msj_version().decode()
        src_hash = version.rsplit("src:", 1)[-1] if "src:" in version else None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                measured = tj.get("_measured_peaks_gbps", {})
                ent = tj.get(args.workload)
                if isinstance(ent, dict) and world == 1 and ent.get("bytes_total") == total_len:
                    if ent.get("kernel_src") == src_hash:
                        traffic = ent.get("hbm_bytes_per_launch")
                    else:
                        traffic_note = (f"profiles/traffic.json holds {ent.get('hbm_bytes_per_launch')} B for kernel "
                                        f"src:{ent.get('kernel_src')}, this is src:{src_hash}: not reported")
            except Exception:
                traffic = None
        out = {
            "metric": "stage-1 GB/s JSON ingested (+ % HBM3E peak), 1/2/4/8 MI355X",
            "value": round(value, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            # what really ran between the start of the process and the timed window: the W warm-up steps, the K steps of
            # the `unsettled` window and the settle passes (config.dvfs_settle)
            "untimed_passes_before_window": args.warmup + ((args.steps + settle_steps) if args.settle_ms > 0 else 0),
            "ms_per_step": round(ms_per_step, 4),
            **({"ms_per_step_median": step_dist["median"], "ms_per_step_min": step_dist["min"],
                "ms_per_step_p95": step_dist["p95"]} if step_dist else {}),
            **({"scaling_efficiency": scaling_eff} if scaling_eff else {}),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{args.gib_per_gpu:g} GiB/GPU synthetic {args.workload} twitter-like JSON "
                            f"(unit {unit_len} B x {total_len // unit_len}), bit-exact vs CPU stage_1 oracle",
                "bytes_total": total_len,
                "structurals_total": total_count,
                "density": round(total_count / total_len, 5),
                "utf8_validation": not args.no_utf8,
                "dvfs_settle": (f"{settle_steps} untimed passes ({args.settle_ms:g} ms) between the {args.warmup} warm-up "
                                f"steps and the {args.steps} timed ones; see 'unsettled'") if args.settle_ms > 0 else "none",
                **({"diagnostic": "no-emit summary pass: not a stage-1 result"} if args.no_emit else {}),
                **({"rehearsal": "--rehearse-sharded: the N > 1 code path on one rank (RCCL world 1): not the headline"}
                   if args.rehearse_sharded else {}),
                "sharding": "single GPU" if world == 1 else f"{world} byte-range shards of {shard_len_nominal} B "
                                                             f"({-(-shard_len_nominal // 0xFFFF0000)} uint32 segments each), cut inside units; "
                                                             f"one all-gather of 128 B per rank and step",
                # after the timed window: "indices" = every index of every rank compared on the device with the
                # oracle's unit indices + k * unit_len, placed with the stitched offsets, trailer included, and the
                # ranks' 64-bit hashes sum to the closed form; "count" = only the total was checked
                "verified": verified,
                **({"verify": verify_detail} if verify_detail else {}),
                **({"stitch": stitch} if stitch else {}),
                **({"per_step_ms": {**step_dist, "note": "a second window right behind the timed one, every step between "
                                    "its own pair of HIP events; ms_per_step is the K-step window's mean"}} if step_dist else {}),
                "library": version + (f" [--lib {args.lib}]" if args.lib else ""),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                **({"traffic_note": traffic_note} if traffic_note else {}),
                "kernel": "stage1_kernel",
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": round(k_ms, 4),
                # north_star's target is quoted against MEASURED read bandwidth: the trivial kernels of
                # scripts/ubench/hbm_ceilings.hip on the product's grid, walk and store shape -- on this box in this
                # process when the helper library is there (N = 1), else the figures recorded in profiles/traffic.json
                # (another box: +-3-5 %).  Both readings of the target: (N + 4S)/t and the input bytes alone, over the
                # measured read bandwidth; and (N + 4S)/t over what a trivial kernel moves the SAME bytes at
                # (same read : write mix), which no kernel on this grid can exceed.
                **ceiling_fields(achieved, value / world, ceil_box, measured, args.workload),
            },
            "cpu_baseline": cpu,
            **({"end_to_end": e2e} if e2e else {}),
            **({"other_configs": others} if others else {}),
        }
        if unsettled is not None:
            # the same K steps timed right after the W warm-up steps, before the clocks have settled
            u_k_ms = unsettled[1] / args.steps
            # the cold figure beside the sustained one: what a caller who parses one document after idle sees
            out["roofline"]["frac_unsettled"] = round(alg_bytes / (u_k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
            out["unsettled"] = {
                "ms_per_step": round(unsettled[0] / args.steps * 1e3, 4),
                "value": round(total_len * args.steps / unsettled[0] / 1e9, 2),
                "kernel_ms": round(u_k_ms, 4),
                "roofline_frac": round(alg_bytes / (u_k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "note": "rank 0's clock; the first ~100 ms after idle run 10-15 % slower than the sustained rate",
            }
        print(json.dumps(out), flush=True)
    verify_failed = verified.startswith("FAILED") or others_failed
    # a run over RCCL whose stitch did not go through the library's own ncclAllGather on a communicator of all N ranks is
    # not the run the line claims to be: the line is printed (rank 0, above) and every rank leaves with exit code 4
    exchange_failed = (sh is not None and backend == "nccl" and
                       (sh.exchange_used != "rccl" or int(sh.rccl_ranks) != world))
    if exchange_failed and rank == 0:
        print(f"bench.py: the stitch ran over '{sh.exchange_used}' with {sh.rccl_ranks} RCCL ranks, not over RCCL with {world}: "
              f"exit code 4", file=sys.stderr, flush=True)
    if sharded:
        barrier()
        sh.close()  # the RCCL communicator of the stitch goes before torch's process group does
    dev.close()
    if dist is not None:
        dist.destroy_process_group()
    if verify_failed:
        sys.exit(3)
    if exchange_failed:
        sys.exit(4)


if __name__ == "__main__":
    main()
