#!/bin/bash
# round 4: kernel v1 (cheap escaped-start patch, carry by value + echo) -- GPU tests, A/B against the round-3 kernel, VALU counts
cd "$(dirname "$0")/.."
out=gpurun_out/r04c
mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1 || { tail -40 $out/gpu_tests.txt; exit 1; }
tail -3 $out/gpu_tests.txt
bash scripts/ab4.sh "minified utf8 pretty4" variants/base.so variants/v1.so > $out/ab_v1.txt 2>&1
cat $out/ab_v1.txt
MSJ_LIB=$PWD/variants/base.so bash scripts/quick_probe.sh base > $out/valu_base.txt 2>&1
MSJ_LIB=$PWD/variants/v1.so bash scripts/quick_probe.sh v1 > $out/valu_v1.txt 2>&1
cat $out/valu_base.txt $out/valu_v1.txt
timeout -k 10 300 python scripts/stitch_overlap.py 1 8 > $out/stitch_overlap.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" $out/stitch_overlap.txt
