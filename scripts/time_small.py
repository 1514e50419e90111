import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mojo_simdjson_amd import DomParserImplementation
p = DomParserImplementation()
for data in (b'[1, 2]', b'{"a":"' + b'x' * 5000 + b'"}', b'[1,' * 30000 + b'1' + b']' * 30000):
    p.stage1(data)
    t0 = time.perf_counter()
    for _ in range(50):
        rc = p.stage1(data)
    dt = (time.perf_counter() - t0) / 50
    print(f"len {len(data):7d}: rc {rc} {dt*1e6:9.1f} us per call")
