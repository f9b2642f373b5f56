#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PROBE_R3=1 PROBE_ONLY=walk PROBE_WALKS="8,4" PROBE_REPS=12
for b in gemm_probe gemm_probe_new gemm_probe gemm_probe_new gemm_probe_stamps gemm_probe_stamps_new; do
  echo "== $b"; timeout -k 10 120 tools/probes/$b.bin 44 | grep -v "fc1 shape plain" || exit 1
done > gpurun_out/mulpre_ab.log 2>&1
cat gpurun_out/mulpre_ab.log
