#!/bin/bash
# PMC passes over a probe binary (separate rocprofv3 runs per counter group, --kernel-trace only).
# usage on the GPU box: bash tools/pmc_probe.sh <outdir> <probe command...>   (environment variables pass through)
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift; local ctrs="$1"; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1 || echo "pass $name rc=$?"
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "$@"
run sq2 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "$@"
run tcc1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "$@"
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
python3 tools/pmc_probe_fold.py "$OUT"
