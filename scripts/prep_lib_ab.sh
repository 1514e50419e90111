#!/bin/bash
# msj_stage2_prep_device of the product build against other builds of the library, alternating on one box:
#   scripts/prep_lib_ab.sh <workload> <lib.so> [<lib.so> ...]      (output under gpurun_out/prep_lib_ab.txt)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
W=${1:-minified}; shift
OUT=gpurun_out/prep_lib_ab.txt
for r in 1 2 3; do
  timeout -k 10 200 python3 scripts/prep_prof.py $W --iters 150 --warm 100 | tee -a "$OUT" || exit 1
  for l in "$@"; do
    echo "--lib $l" | tee -a "$OUT"
    timeout -k 10 200 python3 scripts/prep_prof.py $W --iters 150 --warm 100 --lib $l | tee -a "$OUT" || exit 1
  done
done
