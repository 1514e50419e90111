#!/bin/bash
# Arbitrary PMC groups, one pass each: scripts/pmc_groups.sh <workload> "<group1>" "<group2>" ...
cd "$(dirname "$0")/.."
W=$1; shift
OUT=gpurun_out/pmc_groups; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
i=0
for G in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace -d $OUT/g$i -o g --output-format csv -- python3 bench.py ${MSJ_LIB:+--lib $MSJ_LIB} --steps 3 --warmup 1 --workload $W --no-cpu-baseline > $OUT/g$i.log 2>&1 || { echo "group $i failed: $G"; tail -3 $OUT/g$i.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_groups/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stage1_kernel' in r.get('Kernel_Name', ''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:30s} per launch {sum(v)/len(v):16.0f}   per 4 KiB tile {sum(v)/len(v)/262144:10.2f}")
PY
