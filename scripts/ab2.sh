#!/bin/bash
# A/B several builds of libmsj_stage1.so in one GPU session, interleaved:  scripts/ab2.sh variants/a.so variants/b.so ...
# Prints ms per step (20 timed launches after 3 warm-ups) per workload, 3 rounds for minified.
cd "$(dirname "$0")/.."
one() {
  timeout -k 10 100 python bench.py --lib $PWD/$1 --steps 20 --warmup 3 --settle-ms 0 --no-cpu-baseline --workload $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '$2', d['ms_per_step'], 'ms', d['value'], 'GB/s', 'frac', d['roofline']['frac'])"
}
for rep in 1 2 3; do for so in "$@"; do one $so minified; done; done
for so in "$@"; do one $so utf8; done
for so in "$@"; do one $so pretty4; done
