#!/bin/bash
# GB/s against the input size per launch (device-resident, sustained):  scripts/size_sweep.sh [lib.so] [sizes...]
cd "$(dirname "$0")/.."
LIB=${1:-mojo_simdjson_amd/libmsj_stage1.so}; shift
SIZES=${@:-0.5 0.75 0.9 1 1.1 1.25 1.5 2 3 3.9}
for g in $SIZES; do
  timeout -k 10 120 python bench.py --lib $PWD/$LIB --steps 40 --warmup 5 --no-cpu-baseline --no-verify --no-ceilings --gib-per-gpu $g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('gib $g', d['config']['bytes_total'], 'B', d['ms_per_step'], 'ms', d['value'], 'GB/s', 'alg', d['roofline']['achieved'], 'ms/GiB', round(d['ms_per_step']/(d['config']['bytes_total']/2**30),4))"
done
