#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box via gpurun).
#   scripts/prof.sh <tag> [bench args...]
# Raw CSVs under /tmp/prof/<tag>/ on the box; summary.txt, bench_line.json and the kernel-stats CSV are copied to
# gpurun_out/prof/<tag>/ (copy what you keep to profiles/).  scripts/traffic_update.py reads /tmp/prof/<tag> on the box.
# Counter passes are separate from --kernel-trace --stats (gpurun refuses mixed
# trace domains with --pmc), and FETCH_SIZE / WRITE_SIZE need separate passes
# (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots").
set -o pipefail
TAG=${1:-run}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$REPO"
OUT=/tmp/prof/$TAG   # the raw CSVs stay on the box (gpurun_out/ is limited to 64 MiB); the summaries are copied at the end
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-other-configs $*"   # the driver's command (bench.py settles the clocks for 400 ms before the timed steps)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $ARGS > "$OUT/kt.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/kt.log"; exit 1; }
grep "\"metric\"" "$OUT/kt.log" | tail -1 > "$OUT/bench_line.json"
ARGS="$ARGS --settle-ms 0 --no-ceilings"               # counters do not depend on the clocks: short runs for the PMC passes
i=0
[ -n "$PROF_SKIP_PMC" ] && { KEEP=gpurun_out/prof/$TAG; mkdir -p "$KEEP"; cp "$OUT/bench_line.json" "$KEEP/"; for f in "$OUT"/kt/*/*_kernel_stats.csv; do cp "$f" "$KEEP/kernel_stats.csv"; done; exit 0; }   # PROF_SKIP_PMC=1: the kernel-trace pass only
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 bench.py $ARGS > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($PMC) failed"; tail -3 "$OUT/pmc$i.log"; }
done
python3 scripts/prof_summary.py "$OUT" | tee "$OUT/summary.txt"
KEEP=gpurun_out/prof/$TAG; mkdir -p "$KEEP"
cp "$OUT/summary.txt" "$OUT/bench_line.json" "$KEEP/" 2>/dev/null
for f in "$OUT"/kt/*/*_kernel_stats.csv; do cp "$f" "$KEEP/kernel_stats.csv" 2>/dev/null; done
