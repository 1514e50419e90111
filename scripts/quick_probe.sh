#!/bin/bash
# Short form of valu_probe.sh: the three BASELINE workloads only (MSJ_LIB=path selects the build: passed to bench.py as --lib).
cd "$(dirname "$0")/.."
TAG=${1:-quick}
OUT=/tmp/valu_probe/$TAG; rm -rf $OUT; mkdir -p $OUT  # raw counter files stay on the box (gpurun_out/ is limited to 64 MiB): the summary lines go to stdout
export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
      --kernel-trace -d $OUT/$name -o g --output-format csv -- python3 bench.py ${MSJ_LIB:+--lib $MSJ_LIB} --steps 6 --warmup 2 --settle-ms 0 --no-cpu-baseline "$@" > $OUT/$name.log 2>&1 || echo "$name failed"
  python3 - "$OUT/$name" "$name" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stage1_kernel' in r.get('Kernel_Name', ''):
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
dur = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'stage1_kernel' in r.get('Kernel_Name', ''):
            dur.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print(f"{sys.argv[2]:22s}", " ".join(f"{k.replace('SQ_','')}={sum(v)/len(v)/262144:.1f}" for k, v in sorted(agg.items())),
      f"avg_us={sum(dur)/max(1,len(dur))/1e3:.1f} (n={len(dur)})")
PY
}
run minified
run minified_noemit --no-emit
run utf8 --workload utf8
run pretty4 --workload pretty4
