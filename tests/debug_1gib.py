import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import synth
from mojo_simdjson_amd.device import Stage1Device
from tests import helpers
o = helpers.load_oracle()
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
u = synth.workload("minified", 64 << 20)
code, n, idx = helpers.run_oracle(o.msj_oracle_stage1, u.tobytes())
dev = Stage1Device(0)
d_unit = torch.from_numpy(u).to(dev.device)
reps = max(1, (mb << 20) // u.size)
d_buf = d_unit.repeat(reps)
total = d_buf.numel()
d_idx = torch.full((n * reps + 3,), -1, dtype=torch.int32, device=dev.device)
d_res = dev.new_carry()
for it in range(3):
    d_idx.fill_(-1)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    dev.index(d_buf, d_idx, d_res)
    r = dev.fetch(d_res)
    print(f"launch+fetch {time.perf_counter()-t0:.4f} s")
    print(f"iter {it}: code {r.code} count {r.count} expect {n*reps} internal {r.internal_error} in_string {r.in_string} unesc {r.unescaped_error} utf8 {r.utf8_error}")
    unit_idx = torch.from_numpy(idx[:n].astype(np.int64)).to(dev.device)
    bad = 0
    for k in range(reps):
        got = d_idx[k * n:(k + 1) * n].to(torch.int64) & 0xFFFFFFFF
        neq = (got != unit_idx + k * u.size)
        if bool(neq.any()):
            w = torch.nonzero(neq).flatten()
            print(f"  rep {k}: {int(neq.sum())} mismatches, first at {int(w[0])} got {int(got[w[0]])} want {int(unit_idx[w[0]]) + k*u.size}; unwritten {(d_idx[k*n:(k+1)*n] == -1).sum().item()}")
            bad += 1
            if bad > 3: break
