#!/bin/bash
# Sustained A/B of several builds of libmsj_stage1.so in one GPU session, interleaved:
#   scripts/ab3.sh variants/a.so variants/b.so ...
# 1500 timed launches after 300 warm-ups each (the clocks need ~100 ms of load to settle).
cd "$(dirname "$0")/.."
one() {
  timeout -k 10 100 python bench.py --lib $PWD/$1 --steps 1500 --warmup 300 --settle-ms 0 --no-cpu-baseline --workload $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '$2', d['ms_per_step'], 'ms', d['value'], 'GB/s', 'frac', d['roofline']['frac'])"
}
for rep in 1 2; do for so in "$@"; do one $so minified; done; done
for so in "$@"; do one $so utf8; done
for so in "$@"; do one $so pretty4; done
