#!/bin/bash
# kernel split of msj_stage2_prep_device on a 1 GiB workload (run on the GPU box): scripts/prep_prof.sh <tag> [workload] [--match]
set -o pipefail
TAG=${1:-prep}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$REPO"
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
timeout -k 10 300 python3 scripts/prep_prof.py "$@" | tee "$OUT/rate.txt" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 scripts/prep_prof.py "$@" --iters 50 --warm 50 > "$OUT/kt.log" 2>&1 || { tail -5 "$OUT/kt.log"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT/kernels.txt"
import csv, glob, sys
best = None
for f in glob.glob(sys.argv[1] + "/kt/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    if best is None or len(rows) > len(best): best = rows
for r in best:
    if int(r["Calls"]) >= 50:
        print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.1f} per-prep-call_us={float(r["TotalDurationNs"])/1e3/100:9.1f}')
PY
