#!/bin/bash
# per-kernel split of msj_stage2_prep_device with bracket partners (rocprofv3 --kernel-trace --stats), for several builds:
#   scripts/prep_split.sh <workload> [lib.so ...]     (no lib = the product; PAIRS=1: the compact list instead of match[])
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
W=${1:-minified}; shift
for l in "${@:-}"; do
  rm -rf /tmp/prep_kt
  A=""; [ -n "$l" ] && A="--lib $l"
  echo "== ${l:-product} $W"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prep_kt -- python3 scripts/prep_prof.py $W $([ -n "$PAIRS" ] && echo --pairs || echo --match) --iters 100 --warm 50 $A > /tmp/prep_kt.log 2>&1 || { tail -5 /tmp/prep_kt.log; exit 1; }
  grep -v amdgpu /tmp/prep_kt.log | tail -1
  python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/prep_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'msj_tokens' in r['Name'] and float(r['TotalDurationNs']) > 2e6:
            print(f"  {r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
done
