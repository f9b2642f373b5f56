#!/bin/bash
# PMC passes for the hot kernels (separate rocprofv3 runs per counter group; never combined with
# tracing other than --kernel-trace).  Usage on the GPU box: bash tools/run_pmc.sh <outdir> <what>
set -u
OUT=${1:-gpurun_out/pmc}; WHAT=${2:-all}; BATCH=${3:-32}; CONFIG=${4:-1.5b-moe}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/prof_kernels.py 3 "$WHAT" "$BATCH" "$CONFIG" > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_traffic.py "$OUT"
