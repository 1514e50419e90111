import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from mojo_simdjson_amd.device import Stage1Device
dev = Stage1Device(0)
def run(data):
    d_buf = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev.device)
    d_idx = torch.empty(len(data) + 7, dtype=torch.int32, device=dev.device)
    cin, cout = dev.new_carry(), dev.new_carry()
    dev.shard(d_buf, len(data), d_idx, cin, cout, is_final=False)
    n = int(dev.fetch(cout).count)
    e, f = dev.token_spans(d_buf, len(data), d_idx, n)
    print(len(data), d_idx[:n].cpu().numpy().tolist()[-4:], e.cpu().numpy().tolist()[-4:], f.cpu().numpy().tolist()[-4:])
base = b'["' + b"a" * 5000 + b'", ' + b"9" * 3000 + b', "tail\\'
run(base)
run(base + b" ")
run(base + b"   ")
run(b'["a", 9, "tail\\')
run(b'["a", 9, "tail')
run(b'["a", 9, "tail\\\\')
run(b"x" * 16 + b'"tail\\')
