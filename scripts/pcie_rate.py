"""Host-pointer entry point (msj_stage1): rate including H2D of the input and D2H of the indices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import DomParserImplementation, synth
u = synth.workload("minified", 256 << 20)
data = u.tobytes()
p = DomParserImplementation()
p.stage1(data)
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    rc = p.stage1(data)
dt = (time.perf_counter() - t0) / reps
print(f"msj_stage1 host-pointer form: {len(data)} B, rc {rc}, n {p.n_structural_indexes}, {dt*1e3:.1f} ms, {len(data)/dt/1e9:.2f} GB/s (PCIe + allocation of the result array included)")
