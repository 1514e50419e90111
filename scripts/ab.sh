#!/bin/bash
# A/B several builds of libmsj_stage1.so in one GPU session:  scripts/ab.sh [bench args] -- variants/*.so
ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for so in "$@"; do
  for rep in 1 2; do
    timeout -k 10 100 python bench.py --lib $PWD/$so --steps 20 --warmup 3 --settle-ms 0 --no-cpu-baseline "${ARGS[@]}" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$so', d['value'], 'GB/s', d['ms_per_step'], 'ms')"
  done
done
