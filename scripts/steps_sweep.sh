#!/bin/bash
# ms per launch against the number of back-to-back launches (why bench.py settles the clocks):  scripts/steps_sweep.sh [lib.so]
cd "$(dirname "$0")/.."
LIB=${1:-mojo_simdjson_amd/libmsj_stage1.so}
for st in 20 100 500 2500; do
  timeout -k 10 200 python bench.py --lib $PWD/$LIB --steps $st --warmup 3 --settle-ms 0 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('steps $st', d['ms_per_step'], 'ms', d['value'], 'GB/s frac', d['roofline']['frac'])"
done
timeout -k 10 200 python bench.py --lib $PWD/$LIB --steps 600 --warmup 3 --settle-ms 0 --no-cpu-baseline --no-verify --gib-per-gpu 3.9 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('3.9 GiB steps 600', d['ms_per_step'], 'ms', d['value'], 'GB/s frac', d['roofline']['frac'])"
timeout -k 10 200 python bench.py --lib $PWD/$LIB --steps 20 --warmup 3 --settle-ms 0 --no-cpu-baseline --no-verify --gib-per-gpu 3.9 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('3.9 GiB steps 20', d['ms_per_step'], 'ms', d['value'], 'GB/s frac', d['roofline']['frac'])"
