#!/bin/bash
# SQ issue/stall counters for the stage-1 kernel, one group per pass (never with trace domains).
# usage: scripts/pmc_sq.sh [workload]
set -e
cd "$(dirname "$0")/.."
W=${1:-minified}
OUT=gpurun_out/pmc_sq
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
         "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $G --kernel-trace -d $OUT/g$i -o g$i --output-format csv -- python3 bench.py ${MSJ_LIB:+--lib $MSJ_LIB} --steps 3 --warmup 1 --workload $W --no-cpu-baseline > $OUT/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob('gpurun_out/pmc_sq/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stage1_kernel' in r.get('Kernel_Name', ''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(f"{k:28s} mean/launch {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
