#!/usr/bin/env python3
"""Debug aid: msj_stage2_prep_device on replicated workloads in ONE context, depth compared with the definition."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers
from mojo_simdjson_amd import synth
from mojo_simdjson_amd.device import Stage1Device

dev = Stage1Device(0)
oracle = helpers.load_oracle()
cache = {}
def run(wl, reps, mode):
    dev.lib.msj_debug_set_span_mode(dev.ctx, mode)
    if wl not in cache:
        u = synth.workload(wl, 64 << 20)
        data = u.tobytes()
        code, n1, idx_u = helpers.run_oracle(oracle.msj_oracle_stage1, data)
        idx_u = idx_u[:n1]
        wt, wd, (final, mn, mx) = helpers.oracle_tokens(data, idx_u)
        cache[wl] = (u, idx_u, wd, mn, mx)
    u, idx_u, wd, mn, mx = cache[wl]
    nu = len(idx_u)
    d_buf = torch.from_numpy(u).to(dev.device).repeat(reps)
    d_idx = torch.empty(nu * reps + 16, dtype=torch.int32, device=dev.device)
    d_res = dev.new_carry()
    dev.index(d_buf, d_idx, d_res)
    n = int(dev.fetch(d_res).count)
    out = dev.stage2_prep(d_buf, d_buf.numel(), d_idx, n, match=False)
    t, d, res = out[0], out[1], out[2]
    wd_d = torch.from_numpy(wd).to(dev.device)
    bad = (d.view(reps, -1) != wd_d.expand(reps, -1)).view(-1)
    nb = int(bad.sum())
    msg = f"{wl} x{reps} mode {mode}: result {res.final_depth} {res.min_depth} {res.max_depth} want 0 {mn} {mx}; depth mismatches {nb}; min(depth[]) {int(d.view(-1)[:n].min())}; n % 2048 = {n % 2048}"
    if nb:
        first = int(torch.nonzero(bad)[0])
        last = int(torch.nonzero(bad)[-1])
        msg += f"; first bad token {first} (block {first // 2048}, +{first % 2048}) got {d[first:first + 4].tolist()} want {wd[(first % nu):(first % nu) + 4].tolist()}; last bad {last} (block {last // 2048})"
    print(msg, flush=True)

for rnd in range(1):
    for wl, reps, mode in [("minified", 16, 0), ("minified", 16, 1), ("utf8", 16, 0), ("utf8", 16, 2), ("pretty8", 16, 0), ("pretty8", 16, 2), ("minified", 63, 0)]:
        run(wl, reps, mode)
