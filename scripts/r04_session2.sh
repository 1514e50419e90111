#!/bin/bash
# round 4: the ceilings table and the three BASELINE workloads' bench lines on ONE box
cd "$(dirname "$0")/.."
out=gpurun_out/r04b
mkdir -p $out
timeout -k 10 300 scripts/bin/hbm_ceilings 1 > $out/hbm_ceilings_1gib.txt 2>&1 || { tail -20 $out/hbm_ceilings_1gib.txt; exit 1; }
grep -v CEILINGS $out/hbm_ceilings_1gib.txt | head -17
for w in minified utf8 pretty4; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --workload $w > $out/bench_$w.json 2> $out/bench_$w.err || { tail -20 $out/bench_$w.err; exit 1; }
  python - <<PY
import json
r = json.load(open("$out/bench_$w.json"))
print("$w", r["value"], r["ms_per_step"], {k: r["roofline"][k] for k in ("achieved", "frac", "frac_unsettled", "measured_read_peak", "frac_of_measured_read", "ingest_frac_of_measured_read", "measured_same_mix_peak", "frac_of_same_mix")}, r["roofline"]["ceilings"])
PY
done
