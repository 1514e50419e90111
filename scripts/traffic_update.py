#!/usr/bin/env python3
"""Rewrite one workload's entry of profiles/traffic.json from a scripts/prof.sh output directory:
    scripts/traffic_update.py gpurun_out/prof/<tag> <workload> <committed summary path>
Takes FETCH_SIZE / WRITE_SIZE (KiB, mean per stage1_kernel launch) from the PMC passes, applies the gfx950 correction
(FETCH_SIZE x 2, MI355X_MICROARCH.md section HBM), and stores the kernel's source hash and the stream size from the
bench line of the traced run, so that bench.py only reports the figure for the kernel it belongs to."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, workload, source = sys.argv[1], sys.argv[2], sys.argv[3]
agg = defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "stage1_kernel" in row.get("Kernel_Name", ""):
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
fetch = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"])
write = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"])
line = json.loads(open(os.path.join(out, "bench_line.json")).read())
lib = line["config"]["library"]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "traffic.json")
tj = json.load(open(path))
tj[workload] = {"hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)), "fetch_size_kib": round(fetch, 1),
                "write_size_kib": round(write, 1), "kernel_src": lib.split("src:", 1)[1].split()[0],
                "bytes_total": line["config"]["bytes_total"], "source": source}
json.dump(tj, open(path, "w"), indent=1)
print(workload, tj[workload])
