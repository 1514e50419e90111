#!/bin/bash
# msj_stage2_prep_device by tiles (mode 2) against by tokens (mode 1) on workloads of falling index density, one box:
#   scripts/prep_density_ab.sh    (output under gpurun_out/prep_density_ab.txt)
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prep_density_ab.txt
: > "$OUT"
for w in pretty2 pretty_tab_crlf pretty4 pretty8; do
  for m in 2 1 2 1; do
    timeout -k 10 200 python3 scripts/prep_prof.py $w --mode $m --iters 100 --warm 80 2>&1 | grep -v amdgpu | tee -a "$OUT"
  done
done
