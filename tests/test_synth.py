"""CPU checks of the synthetic workloads (BASELINE.json configs 2-4)."""
import json

import numpy as np
import pytest

from mojo_simdjson_amd import synth
from tests import helpers


@pytest.mark.parametrize("name", ["minified", "utf8", "pretty2", "pretty4", "pretty8", "pretty_tab_crlf"])
def test_units_are_valid_json(name, oracle):
    u = synth.workload(name, 1 << 20)
    b = u.tobytes()
    doc = json.loads(b)                      # also proves valid UTF-8
    assert len(doc["statuses"]) > 100
    assert len(b) % 128 == 77                # every block/step phase is exercised on repetition
    assert np.array_equal(u, synth.workload(name, 1 << 20))  # deterministic
    code, n, idx = helpers.run_oracle(oracle.msj_oracle_stage1, b)
    assert code == 0 and n > 0
    assert oracle.msj_oracle_utf8(b, len(b)) == 0
    d = n / len(b)
    if name == "minified":
        assert 0.08 < d < 0.25
        assert b.count(b"\\") > 1000         # escapes present
    if name == "utf8":
        assert sum(1 for c in b if c >= 0x80) > len(b) // 3
    # the unit ends with every carry at zero, so indices of a repeated unit are unit + k*len
    rep = b + b
    c2, n2, idx2 = helpers.run_oracle(oracle.msj_oracle_stage1, rep)
    assert c2 == 0 and n2 == 2 * n
    assert np.array_equal(idx2[n:2 * n], idx[:n] + np.uint32(len(b)))


def test_extremes(oracle):
    for kind, lo, hi in ((0, 0.99, 1.01), (1, 0.6, 0.7), (2, 0.0, 0.001), (3, 0.0, 0.001),
                         (4, 0.49, 0.51), (5, 0.39, 0.41)):
        b = synth.extreme(100000, kind).tobytes()
        code, n, _ = helpers.run_oracle(oracle.msj_oracle_stage1, b)
        assert code == 0
        assert lo <= n / len(b) <= hi
