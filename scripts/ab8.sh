#!/bin/bash
# 8 GiB per launch pair (one rank's share of config 5), several builds and index-buffer sizes in turn:  scripts/ab8.sh "<capacity fracs>" a.so b.so ...
cd "$(dirname "$0")/.."
FR=$1; shift
one() { timeout -k 10 200 python bench.py --lib $PWD/$1 --gib-per-gpu 8 --steps 100 --warmup 10 --no-cpu-baseline --no-ceilings --index-capacity-frac $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', '8 GiB, index capacity frac $2', d['ms_per_step'], 'ms', d['value'], 'GB/s', 'frac', d['roofline']['frac'], d['config']['verified'])"; }
for rep in 1 2; do for f in $FR; do for so in "$@"; do one $so $f; done; done; done
