#!/bin/bash
# sustained A/B of several builds of libmsj_stage1.so in one GPU session, interleaved (1 500 timed launches after 300 warm-ups each): scripts/ab.sh "<workloads>" a.so b.so ...  (two alternating rounds each)
cd "$(dirname "$0")/.."
WL=$1; shift
one() {
  timeout -k 10 100 python bench.py --lib $PWD/$1 --steps 1500 --warmup 300 --settle-ms 0 --no-cpu-baseline --workload $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '$2', d['ms_per_step'], 'ms', d['value'], 'GB/s', 'frac', d['roofline']['frac'], d['config']['verified'])"
}
for w in $WL; do for rep in 1 2; do for so in "$@"; do one $so $w; done; done; done
