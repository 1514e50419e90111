#!/bin/bash
# LDS counters of the stage-1 kernel per 4 KiB tile: scripts/pmc_lds.sh [workload]   (MSJ_LIB=path selects the build)
cd "$(dirname "$0")/.."
W=${1:-minified}
OUT=gpurun_out/pmc_lds; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $OUT/g -o g --output-format csv -- python3 bench.py ${MSJ_LIB:+--lib $MSJ_LIB} --steps 3 --warmup 1 --settle-ms 0 --workload $W --no-cpu-baseline --no-verify > $OUT/g.log 2>&1 || echo "failed"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_lds/g/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stage1_kernel' in r.get('Kernel_Name', ''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:26s} per 4 KiB tile {sum(v)/len(v)/262144:10.1f}")
PY
