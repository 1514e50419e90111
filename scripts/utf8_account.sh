#!/bin/bash
# where the UTF-8-heavy workload's instructions and time go -- dynamic counts per tile with and
# without emission / validation (PMC), per-phase shares of a wave's iteration (stamps build), beside minified.
cd "$(dirname "$0")/.."
bash scripts/valu_probe.sh acct 2>&1 | grep -v "^$"
make -C mojo_simdjson_amd/csrc stamps > /dev/null 2>&1 || echo "stamps build failed"
for w in utf8 minified; do echo "== stamps $w"; timeout -k 10 200 python3 scripts/stamps.py $w 2>&1 | grep -v amdgpu; done
