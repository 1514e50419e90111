#!/bin/bash
# the token tests against another build of the library (a variant under test, or one without a fix: they must fail
# there), then the product build against it on a 1 GiB workload:  scripts/check_against_ref.sh <lib.so> [workload] [pytest -k expression]
cd "$(dirname "$0")/.."
REF=$(readlink -f "$1"); W=${2:-minified}; K=${3:-tile_groups or token_spans or stage2_prep or fixtures}
echo "--- tests/test_tokens.py -k '$K' against $1:"
python3 - "$REF" "$K" <<'PY'
import subprocess, sys
code = f"""
import sys
sys.path.insert(0, ".")
from mojo_simdjson_amd import _lib
_lib.LIB_PATH = {sys.argv[1]!r}
import pytest
sys.exit(pytest.main(["tests/test_tokens.py", "-x", "-q", "-m", "gpu", "-k", {sys.argv[2]!r}]))
"""
r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
print(r.stdout[-900:])
PY
bash scripts/prep_ab.sh libs $W "$1" 2>&1 | grep -v amdgpu | cut -c1-110
