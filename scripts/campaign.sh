#!/bin/bash
# After a change of the stage-1 kernel (round 4): GPU tests, every stress tool, rocprofv3 stats + PMC of the three BASELINE
# workloads with the traffic entries for the new source hash, the ceilings table, density / size sweeps with same-box
# ceilings, the default and the 8 GiB bench lines, the stitch overlap on one GPU.   scripts/campaign.sh <tag> [part]
cd "$(dirname "$0")/.."
T=${1:-r05}; PART=${2:-all}; O=gpurun_out/$T; mkdir -p $O
if [ $PART = all ] || [ $PART = 1 ]; then
timeout -k 10 800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; tail -2 $O/gpu_tests.txt | cut -c1-200
timeout -k 10 150 python tests/stress.py 60 201 > $O/stress_201.txt 2>&1; tail -1 $O/stress_201.txt
MSJ_STRESS_FLAGS=0x100 timeout -k 10 100 python tests/stress.py 25 202 > $O/stress_twopass_202.txt 2>&1; tail -1 $O/stress_twopass_202.txt
timeout -k 10 120 python tests/stress_sharded.py 45 203 > $O/stress_sharded_203.txt 2>&1; tail -1 $O/stress_sharded_203.txt
timeout -k 10 120 python tests/stress_tokens.py 45 204 > $O/stress_tokens_204.txt 2>&1; tail -1 $O/stress_tokens_204.txt
timeout -k 10 120 python tests/stress_documents.py 45 205 > $O/stress_documents_205.txt 2>&1; tail -1 $O/stress_documents_205.txt
timeout -k 10 100 python tests/stress_host.py 30 206 > $O/stress_host_206.txt 2>&1; tail -1 $O/stress_host_206.txt
fi
if [ $PART = all ] || [ $PART = 2 ]; then
for w in minified utf8 pretty4; do
  bash scripts/prof.sh ${T}_$w --workload $w > $O/prof_$w.txt 2>&1
  cp gpurun_out/prof/${T}_$w/summary.txt $O/summary_${T}_${w}_1gib.txt 2>/dev/null
  cp gpurun_out/prof/${T}_$w/bench_line.json $O/bench_${T}_${w}_1gib.json 2>/dev/null
  cp gpurun_out/prof/${T}_$w/kernel_stats.csv $O/kernel_stats_${T}_${w}_1gib.csv 2>/dev/null
  python3 scripts/traffic_update.py /tmp/prof/${T}_$w $w profiles/${T}/summary_${T}_${w}_1gib.txt > $O/traffic_$w.txt 2>&1; cut -c1-200 $O/traffic_$w.txt
done
rm -rf gpurun_out/prof
timeout -k 10 120 scripts/bin/hbm_ceilings 1 > $O/hbm_ceilings_1gib.txt 2>&1; python3 scripts/ceilings_update.py $O/hbm_ceilings_1gib.txt > $O/ceilings_update.txt 2>&1; tail -3 $O/ceilings_update.txt
cp profiles/traffic.json $O/traffic.json
fi
if [ $PART = all ] || [ $PART = 3 ]; then
timeout -k 10 400 python tests/density_sweep.py > $O/density_sweep_$T.txt 2>&1; grep -v amdgpu $O/density_sweep_$T.txt | cut -c1-24,60-400
bash scripts/size_sweep.sh mojo_simdjson_amd/libmsj_stage1.so 0.25 0.5 1 2 3.9 > $O/size_sweep_$T.txt 2>&1; tail -8 $O/size_sweep_$T.txt
timeout -k 10 200 python scripts/small_launch.py 2>&1 | grep -v amdgpu > $O/small_launch_$T.txt; cat $O/small_launch_$T.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_${T}_default.json 2> $O/bench.err; cut -c1-400 $O/bench_${T}_default.json
timeout -k 10 200 python bench.py --gib-per-gpu 8 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${T}_minified_8gib.json 2> $O/bench8.err; cut -c1-200 $O/bench_${T}_minified_8gib.json
timeout -k 10 300 python scripts/stitch_overlap.py 1 8 2>&1 | grep "GiB" > $O/stitch_overlap.txt; cat $O/stitch_overlap.txt
MSJ_BENCH_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 10 --warmup 2 --settle-ms 0 > $O/bench_${T}_n2_gloo_one_gpu.json 2> $O/bench_n2.err; cut -c1-300 $O/bench_${T}_n2_gloo_one_gpu.json
fi
