"""What the stitch costs a step on ONE GPU, and whether the pipeline overlaps it.

World 1 over the real RCCL backend (the library's own ncclAllGather on the exchange's side stream):
  (a) the shard through msj_stage1_shard_device alone -- no exchange, no host turn: the kernel's rate;
  (b) submit -> result, one step at a time: kernel + exchange + read-back + host turn, nothing overlapped;
  (c) three submissions in flight, result(k) waiting for ITS slot's event only: what bench.py --gpus N runs.
(c) - (a) is what the stitch still costs per step; (b) - (a) what it costs when nothing hides it.
Usage: python scripts/stitch_overlap.py [GiB ...]   (default 1 8)"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import sharded, synth  # noqa: E402
from mojo_simdjson_amd.device import Stage1Device  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29871")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = Stage1Device(0)
unit = synth.workload("minified", 64 << 20)
d_unit = torch.from_numpy(unit).to(dev.device)
K = 60
for gib in [float(x) for x in sys.argv[1:]] or [1.0, 8.0]:
    reps = int(gib * (1 << 30)) // unit.size
    n_bytes = reps * int(unit.size)
    d_buf = synth.stream_shard(d_unit, int(unit.size), 0, n_bytes)
    d_idx = torch.empty(int(n_bytes * 0.3), dtype=torch.int32, device=dev.device)
    d_seg = torch.zeros(8 * 32, dtype=torch.uint8, device=dev.device)
    d_in, d_out = dev.make_carry(0, 0, 0), dev.new_carry()
    sh = sharded.ShardedStage1(dev, 0, 1, always_gather=True, exchange="rccl")

    def alone():
        dev.shard(d_buf, n_bytes, d_idx, d_in, d_out, segments=d_seg, is_final=True, trailer_len=n_bytes)

    def submit():
        return sh.submit(d_buf, n_bytes, d_idx, n_bytes, has_prefix=False, segments=d_seg)

    def window(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / K, e0.elapsed_time(e1) / K

    def run_alone():
        for _ in range(K):
            alone()

    def run_serial():
        for _ in range(K):
            sh.result(submit())

    def run_pipelined():
        pending = []
        for _ in range(K):
            pending.append(submit())
            if len(pending) >= sh.DEPTH:
                sh.result(pending.pop(0))
        while pending:
            sh.result(pending.pop(0))

    for _ in range(300 if gib <= 2 else 60):  # settle the clocks
        alone()
    sh.result(submit())
    rows = []
    for name, fn in (("kernel alone", run_alone), ("submit -> result, serial", run_serial), ("three in flight", run_pipelined),
                     ("kernel alone (again)", run_alone), ("three in flight (again)", run_pipelined)):
        s0 = sh.stats()
        wall, ev = window(fn)
        s1 = sh.stats()
        d = {k: s1[k] - s0[k] for k in s1}
        extra = ""
        if d["results"]:
            extra = (f"  kernel-only {d['kernel_device_ns'] / d['results'] / 1e6:.4f} ms, kernel end -> reports in "
                     f"{d['stitch_device_ns'] / max(1, d['rounds']) / 1e3:.1f} us, host blocked {d['result_wait_ns'] / d['results'] / 1e3:.1f} us per step")
        rows.append((name, wall, ev))
        print(f"{gib:g} GiB  {name:28s} {wall:.4f} ms/step wall  {ev:.4f} ms/step events{extra}", flush=True)
    base = min(rows[0][1], rows[3][1])
    print(f"{gib:g} GiB  stitch per step: serial +{(rows[1][1] - base) * 1e3:.1f} us, three in flight "
          f"+{(min(rows[2][1], rows[4][1]) - base) * 1e3:.1f} us over the kernel alone ({base:.4f} ms); "
          f"{n_bytes / min(rows[2][1], rows[4][1]) / 1e6:.0f} GB/s pipelined vs {n_bytes / base / 1e6:.0f} GB/s alone", flush=True)
    sh.close()
    del d_buf, d_idx
dev.close()
dist.destroy_process_group()
