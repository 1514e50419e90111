#!/bin/bash
# token_tiles with issue priorities by phase (builds scripts/libmsj_prio_<classification>_<chunk loop>.so) against the
# product build, alternating on one box: scripts/prep_prio_ab.sh
set -o pipefail
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prep_prio_ab.txt
: > "$OUT"
for r in 1 2; do
  timeout -k 10 200 python3 scripts/prep_prof.py minified --iters 150 --warm 100 | tee -a "$OUT" || exit 1
  for l in scripts/libmsj_prio_*.so; do
    echo "--lib $l" | tee -a "$OUT"
    timeout -k 10 200 python3 scripts/prep_prof.py minified --iters 150 --warm 100 --lib $l | tee -a "$OUT" || exit 1
  done
done
