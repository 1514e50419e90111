#!/bin/bash
# the tile-group tests against another build of the library (e.g. one without a fix: they must fail there), then the
# product build against it on the 1 GiB minified workload:  scripts/check_against_ref.sh <lib.so>
cd "$GRAFT_REPO_ROOT"
REF=$(readlink -f "$1")
timeout -k 10 600 python -m pytest tests/test_tokens.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-250
echo "--- test_prep_around_the_tile_groups against $1:"
python3 - "$REF" <<'PY'
import subprocess, sys
code = f"""
import sys
sys.path.insert(0, ".")
from mojo_simdjson_amd import _lib
_lib.LIB_PATH = {sys.argv[1]!r}
import pytest
sys.exit(pytest.main(["tests/test_tokens.py", "-x", "-q", "-m", "gpu", "-k", "around_the_tile_groups"]))
"""
r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
print(r.stdout[-700:])
PY
bash scripts/prep_lib_ab.sh minified "$1" 2>&1 | grep -v amdgpu | cut -c1-110
