#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_tests2.log 2>&1; rc=$?
tail -8 gpurun_out/r3_tests2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc $rc)"; exit $rc; fi
PROBE_R3=1 PROBE_ONLY=walk PROBE_REPS=9 timeout -k 10 300 tools/probes/gemm_probe.bin 44 > gpurun_out/r3_probe_walk2.log 2>&1 || exit 1
cat gpurun_out/r3_probe_walk2.log
PROBE_R3=1 PROBE_REPS=2 PROBE_WALKS="0,0;16,11" bash tools/pmc_probe.sh gpurun_out/r3_pmc_gemm tools/probes/gemm_probe.bin 44 > gpurun_out/r3_pmc_gemm.log 2>&1
tail -30 gpurun_out/r3_pmc_gemm.log
exit $rc
