#!/bin/bash
# PMC passes for msj_stage2_prep_device's kernels (run on the GPU box): scripts/prep_pmc.sh <tag> [workload] [--match]
set -o pipefail
TAG=${1:-prep}; shift
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$REPO"
OUT=gpurun_out/prof/$TAG
mkdir -p "$OUT"
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $PMC --output-format csv -d "$OUT/pmc$i" -- python3 scripts/prep_prof.py "$@" --iters 10 --warm 3 > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($PMC) failed"; tail -3 "$OUT/pmc$i.log"; }
done
python3 scripts/prof_summary.py "$OUT" | tee "$OUT/summary.txt"
