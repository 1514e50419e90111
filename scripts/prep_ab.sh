#!/bin/bash
# msj_stage2_prep_device, kernel organised by tiles (mode 2) against the one organised by tokens (mode 1), same box,
# alternating: scripts/prep_ab.sh  (run on the GPU box; output under gpurun_out/prep_ab.txt)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prep_ab.txt
: > "$OUT"
for w in minified utf8 pretty4; do
  for m in 2 1 2 1; do
    timeout -k 10 200 python3 scripts/prep_prof.py $w --mode $m --iters 150 --warm 100 | tee -a "$OUT" || exit 1
  done
done
