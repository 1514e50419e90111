#!/bin/bash
# round 4, first GPU session: the pipelined stitch (tests + what it costs), the re-measured ceilings, the default bench
cd "$(dirname "$0")/.."
out=gpurun_out/r04a
mkdir -p $out
python -m pytest tests/test_stage1_gpu.py -x -q -m gpu -k "sharded or rccl" -s > $out/gpu_tests_sharded.txt 2>&1 || { tail -30 $out/gpu_tests_sharded.txt; exit 1; }
tail -3 $out/gpu_tests_sharded.txt
timeout -k 10 300 python scripts/stitch_overlap.py 1 8 > $out/stitch_overlap.txt 2>&1 || { tail -20 $out/stitch_overlap.txt; exit 1; }
cat $out/stitch_overlap.txt
timeout -k 10 300 scripts/bin/hbm_ceilings 1 > $out/hbm_ceilings_1gib.txt 2>&1 || { tail -20 $out/hbm_ceilings_1gib.txt; exit 1; }
cat $out/hbm_ceilings_1gib.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
cat $out/bench_default.json
python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1 || { tail -40 $out/gpu_tests.txt; exit 1; }
tail -3 $out/gpu_tests.txt
