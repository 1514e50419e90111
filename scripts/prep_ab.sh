#!/bin/bash
# msj_stage2_prep_device A/B on one box, alternating (output also under gpurun_out/prep_ab.txt):
#   scripts/prep_ab.sh modes  [workloads...]             kernel organised by tiles (mode 2) against by tokens (mode 1)
#   scripts/prep_ab.sh libs <workload> <lib.so> [...]    the product build against other builds of the library
#   add MATCH=1 for the call with bracket partners
set -o pipefail
cd "$(dirname "$0")/.."
KIND=${1:-modes}; shift
OUT=gpurun_out/prep_ab.txt; mkdir -p gpurun_out; : > "$OUT"
M=${MATCH:+--match}
if [ "$KIND" = modes ]; then
  for w in ${@:-minified utf8 pretty4}; do for m in 2 1 2 1; do
    timeout -k 10 200 python3 scripts/prep_prof.py $w --mode $m --iters 150 --warm 100 $M 2>&1 | grep -v amdgpu | tee -a "$OUT" || exit 1
  done; done
else
  W=${1:-minified}; shift
  for r in 1 2 3; do
    timeout -k 10 200 python3 scripts/prep_prof.py $W --iters 150 --warm 100 $M | tee -a "$OUT" || exit 1
    for l in "$@"; do
      echo "--lib $l" | tee -a "$OUT"
      timeout -k 10 200 python3 scripts/prep_prof.py $W --iters 150 --warm 100 $M --lib $l | tee -a "$OUT" || exit 1
    done
  done
fi
