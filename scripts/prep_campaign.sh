#!/bin/bash
# round 4: the token rows -- tests, the uint32-edge test, msj_stage2_prep_device with bracket partners: rate and split
cd "$(dirname "$0")/.."
out=gpurun_out/r04f
mkdir -p $out
python -m pytest tests/test_tokens.py tests/test_documents.py -x -q -m gpu > $out/gpu_tests_tokens.txt 2>&1 || { tail -40 $out/gpu_tests_tokens.txt; exit 1; }
tail -3 $out/gpu_tests_tokens.txt
python -m pytest tests/test_stage1_gpu.py -x -q -m gpu -k "uint32_edge" > $out/gpu_tests_edge.txt 2>&1 || { tail -40 $out/gpu_tests_edge.txt; exit 1; }
tail -3 $out/gpu_tests_edge.txt
for w in minified utf8 pretty4; do
  python scripts/prep_prof.py $w > $out/prep_$w.txt 2>&1; tail -1 $out/prep_$w.txt
  python scripts/prep_prof.py $w --match > $out/prep_match_$w.txt 2>&1; tail -1 $out/prep_match_$w.txt
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prep_kt -- python3 scripts/prep_prof.py minified --match --iters 100 --warm 50 > /tmp/prep_kt.log 2>&1
python3 - <<'PY' > gpurun_out/r04f/prep_match_minified_split.txt
import csv, glob
for f in glob.glob('/tmp/prep_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.2f}")
PY
cat gpurun_out/r04f/prep_match_minified_split.txt
