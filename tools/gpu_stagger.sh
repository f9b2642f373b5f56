#!/bin/bash
# two-per-CU NT kernel: does a start offset between the two work-groups of a CU (odd wave slots start late) change anything?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PROBE_R3=1 PROBE_ONLY=walk PROBE_WALKS="8,4" PROBE_REPS=9
for sp in 0 8 16 24 0 12 20 32; do
  echo "== stagger $sp us"
  NT_SPREAD=$sp timeout -k 10 120 tools/probes/gemm_probe.bin 44 || exit 1
done > gpurun_out/nt_stagger.log 2>&1
cat gpurun_out/nt_stagger.log
