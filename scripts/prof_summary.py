#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by scripts/prof.sh: per-kernel average
duration (kernel trace) and per-launch counter means (PMC passes)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    for row in csv.DictReader(open(f)):
        name = row["Name"][:60]
        print(f"  {name:60s} calls={row['Calls']:>4s} avg_ns={float(row['AverageNs']):12.0f} min_ns={row['MinNs']}")
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "stage1_kernel" in r.get("Kernel_Name", "")]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if d:
        print(f"== stage1_kernel launches in the trace: {len(d)}; first 25 avg {sum(d[:25]) / len(d[:25]):.0f} ns "
              f"(5 warm-up + 20 unsettled steps); last 20 avg {sum(d[-20:]) / len(d[-20:]):.0f} ns (the timed steps); "
              f"all avg {sum(d) / len(d):.0f} ns")
bl = os.path.join(out, "bench_line.json")
if os.path.exists(bl):
    print("== bench line of the traced run:", open(bl).read().strip()[:1200])
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "stage1" not in k and "msj" not in k:
            continue
        agg[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print("== counters (mean per launch):", k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} {sum(v) / len(v):18.1f}   (n={len(v)})")
