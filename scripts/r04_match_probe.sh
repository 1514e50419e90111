#!/bin/bash
# round 4 probe: what bounds match_brackets (grid per list, length of the linear scan)
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in mojo_simdjson_amd/libmsj_stage1.so variants/match_g8_l256.so variants/match_g128_l256.so variants/match_g32_l1024.so variants/match_g32_l64.so; do
  rm -rf /tmp/mp_kt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp_kt -- python3 scripts/prep_prof.py minified --match --iters 60 --warm 30 --lib $v > /tmp/mp.log 2>&1
  python3 - "$v" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/mp_kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'match_brackets' in r['Name'] or 'apply_depth' in r['Name']:
            print(f"{sys.argv[1]:36s} {r['Name'][:40]:40s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f}")
PY
done
