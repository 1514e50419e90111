#!/bin/bash
# per-kernel split of one prep_prof.py run (rocprofv3 --kernel-trace --stats): scripts/split_any.sh "<prep_prof.py arguments>" [name filter]
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
rm -rf /tmp/prep_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prep_kt -- python3 scripts/prep_prof.py --iters 100 --warm 50 $1 > /tmp/prep_kt.log 2>&1 || { tail -5 /tmp/prep_kt.log; exit 1; }
echo "== $1"
FILTER="${2:-msj_tokens}" python3 - <<'PY'
import csv, glob, os
for f in glob.glob('/tmp/prep_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ["FILTER"] in r['Name'] and float(r['TotalDurationNs']) > 2e6:
            print(f"  {r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
