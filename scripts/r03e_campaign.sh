#!/bin/bash
# after a change of the stage-1 kernel: the GPU tests and stress tools, then the profiles of the three BASELINE workloads
# (kernel stats, PMC, bench lines) and the traffic entries for the new source hash:  scripts/r03e_campaign.sh <tag>
cd "$(dirname "$0")/.."
T=${1:-r03e}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt | cut -c1-200
timeout -k 10 150 python tests/stress.py 70 101 > $O/stress_101.txt 2>&1; tail -1 $O/stress_101.txt
MSJ_STRESS_FLAGS=0x100 timeout -k 10 100 python tests/stress.py 30 102 > $O/stress_twopass_102.txt 2>&1; tail -1 $O/stress_twopass_102.txt
timeout -k 10 120 python tests/stress_sharded.py 50 103 > $O/stress_sharded_103.txt 2>&1; tail -1 $O/stress_sharded_103.txt
for w in minified utf8 pretty4; do
  bash scripts/prof.sh ${T}_$w --workload $w > $O/prof_$w.txt 2>&1
  cp gpurun_out/prof/${T}_$w/summary.txt $O/summary_${T}_${w}_1gib.txt 2>/dev/null
  cp gpurun_out/prof/${T}_$w/bench_line.json $O/bench_${T}_${w}_1gib.json 2>/dev/null
  for f in gpurun_out/prof/${T}_$w/kt/*/*_kernel_stats.csv; do cp $f $O/kernel_stats_${T}_${w}_1gib.csv; done
  python3 scripts/traffic_update.py gpurun_out/prof/${T}_$w $w profiles/r03/summary_${T}_${w}_1gib.txt > $O/traffic_$w.txt 2>&1; cat $O/traffic_$w.txt | cut -c1-200
done
cp profiles/traffic.json $O/traffic.json
timeout -k 10 300 python tests/density_sweep.py > $O/density_sweep_$T.txt 2>&1; grep -v amdgpu $O/density_sweep_$T.txt | cut -c1-170
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_${T}_default.json 2> $O/bench.err; cut -c1-300 $O/bench_${T}_default.json
timeout -k 10 200 python bench.py --gib-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${T}_minified_8gib.json 2> $O/bench8.err; cut -c1-200 $O/bench_${T}_minified_8gib.json
