#!/bin/bash
# Core clock per stage1_kernel launch: GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) against the launch's duration
# from the kernel trace, for the launches right after idle (unsettled) and after 400 ms of load (settled).
#   scripts/clock_probe.sh [workload]      (MSJ_LIB=path selects the build: passed to bench.py as --lib)
cd "$(dirname "$0")/.."
W=${1:-minified}
OUT=gpurun_out/clock_probe; rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
for mode in 0 400; do
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $OUT/m$mode -o g --output-format csv -- \
      python3 bench.py ${MSJ_LIB:+--lib $MSJ_LIB} --steps 20 --warmup 5 --settle-ms $mode --no-cpu-baseline --no-verify --workload $W > $OUT/m$mode.log 2>&1 || echo "mode $mode failed"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mode in ("0", "400"):
    dur = {}
    for f in glob.glob(f"{out}/m{mode}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stage1_kernel" in r.get("Kernel_Name", ""):
                dur[r["Dispatch_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    ctr = collections.defaultdict(dict)
    for f in glob.glob(f"{out}/m{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stage1_kernel" in r.get("Kernel_Name", ""):
                ctr[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    rows = sorted((dur[d][0], d) for d in dur if d in ctr)
    print(f"--settle-ms {mode}: {len(rows)} launches; per launch: duration us, GRBM_GUI_ACTIVE / 8 XCDs, clock GHz, VALU busy share, issue-stall share, wait share")
    sel = rows if mode == "0" else rows[:8] + rows[-20:]
    for k, (t, d) in enumerate(sel):
        c = ctr[d]
        ns = dur[d][1]
        cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
        wc = c.get("SQ_WAVE_CYCLES", 1)
        print(f"  {k:3d} {ns/1e3:8.1f} {cyc:10.0f} {cyc/ns:6.3f}  valu {c.get('SQ_ACTIVE_INST_VALU',0)/wc:5.3f} stall {c.get('SQ_WAIT_INST_ANY',0)/wc:5.3f} wait {c.get('SQ_WAIT_ANY',0)/wc:5.3f}")
PY
