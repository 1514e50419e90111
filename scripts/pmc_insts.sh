#!/bin/bash
# Dynamic instruction counts per tile for one build: scripts/pmc_insts.sh [workload] (MSJ_LIB=path selects the build: passed to bench.py as --lib)
cd "$(dirname "$0")/.."
W=${1:-minified}
OUT=gpurun_out/pmc_insts; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM --kernel-trace -d $OUT/g -o g --output-format csv -- python3 bench.py ${MSJ_LIB:+--lib $MSJ_LIB} --steps 3 --warmup 1 --settle-ms 0 --workload $W --no-cpu-baseline > $OUT/g.log 2>&1 || echo "failed"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_insts/g/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stage1_kernel' in r.get('Kernel_Name', ''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:24s} per 4 KiB tile {sum(v)/len(v)/262144:10.1f}")
PY
