#!/bin/bash
# Round-3 measurement campaign (run on the GPU box in two gpurun calls: part 1 = profiles, part 2 = sweeps and stress).
#   scripts/r03_campaign.sh 1|2
cd "$(dirname "$0")/.."
O=gpurun_out/r03; mkdir -p $O
if [ "$1" = "1" ]; then
  for w in minified utf8 pretty4; do
    bash scripts/prof.sh r03_$w --workload $w > $O/prof_$w.txt 2>&1
    cp gpurun_out/prof/r03_$w/summary.txt $O/summary_r03_${w}_1gib.txt 2>/dev/null
    cp gpurun_out/prof/r03_$w/bench_line.json $O/bench_r03_${w}_1gib.json 2>/dev/null
    for f in gpurun_out/prof/r03_$w/kt/*/*_kernel_stats.csv; do cp $f $O/kernel_stats_r03_${w}_1gib.csv; done
    echo "== $w"; tail -45 $O/prof_$w.txt | cut -c1-200
  done
  bash scripts/valu_probe.sh r03 > $O/valu_probe_r03.txt 2>&1; cat $O/valu_probe_r03.txt | cut -c1-250
else
  timeout -k 10 300 python tests/density_sweep.py > $O/density_sweep_r03.txt 2>&1; grep -v amdgpu $O/density_sweep_r03.txt | cut -c1-170
  bash scripts/steps_sweep.sh > $O/steps_sweep_r03.txt 2>&1; cat $O/steps_sweep_r03.txt
  bash scripts/size_sweep.sh mojo_simdjson_amd/libmsj_stage1.so 0.25 0.5 1 2 3.9 > $O/size_sweep_r03.txt 2>&1; cat $O/size_sweep_r03.txt
  timeout -k 10 200 python bench.py --gib-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_r03_minified_8gib.json 2> $O/bench_8gib.err; cut -c1-400 $O/bench_r03_minified_8gib.json
  MSJ_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_r03_n2_gloo_one_gpu.json 2> $O/bench_n2.err; cut -c1-300 $O/bench_r03_n2_gloo_one_gpu.json
  for w in minified utf8 pretty4; do timeout -k 10 120 python scripts/prep_prof.py $w 2>/dev/null | tail -1; done > $O/stage2_prep_r03_rates.txt; cat $O/stage2_prep_r03_rates.txt
  bash scripts/prep_prof.sh r03_prep_minified minified > $O/stage2_prep_r03_minified_1gib.txt 2>&1; tail -12 $O/stage2_prep_r03_minified_1gib.txt | cut -c1-160
  timeout -k 10 150 python tests/stress.py 100 31 > $O/stress_31.txt 2>&1; tail -1 $O/stress_31.txt
  MSJ_STRESS_FLAGS=0x100 timeout -k 10 100 python tests/stress.py 50 32 > $O/stress_twopass_32.txt 2>&1; tail -1 $O/stress_twopass_32.txt
  timeout -k 10 120 python tests/stress_sharded.py 70 33 > $O/stress_sharded_33.txt 2>&1; tail -1 $O/stress_sharded_33.txt
  timeout -k 10 120 python tests/stress_tokens.py 60 34 > $O/stress_tokens_34.txt 2>&1; tail -1 $O/stress_tokens_34.txt
  timeout -k 10 120 python tests/stress_documents.py 60 35 > $O/stress_documents_35.txt 2>&1; tail -1 $O/stress_documents_35.txt
  timeout -k 10 120 python tests/stress_host.py 60 36 > $O/stress_host_36.txt 2>&1; tail -1 $O/stress_host_36.txt
  if [ -f variants/r3_base.so ]; then  # round 2's kernel (+ the capacity flag) against this build, same box, interleaved
    cp mojo_simdjson_amd/libmsj_stage1.so variants/r3_ship.so
    bash scripts/ab3.sh variants/r3_base.so variants/r3_ship.so > $O/ab_sustained_r02_r03.txt 2>&1; cat $O/ab_sustained_r02_r03.txt
    bash scripts/ab2.sh variants/r3_base.so variants/r3_ship.so > $O/ab_unsettled_r02_r03.txt 2>&1; cat $O/ab_unsettled_r02_r03.txt
  fi
  bash scripts/clock_probe.sh minified > $O/clock_probe_r03_minified.txt 2>&1; tail -30 $O/clock_probe_r03_minified.txt
fi
