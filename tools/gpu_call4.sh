#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_moe_kernels_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/r3_tests4.log 2>&1; rc=$?
tail -4 gpurun_out/r3_tests4.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc $rc)"; exit $rc; fi
L=gpurun_out/r3_probe_old_new.log; : > $L
for v in old base old base; do echo "== $v" >> $L; PROBE_R3=1 PROBE_WALKS="8,4" PROBE_REPS=7 timeout -k 10 120 tools/probes/gemm_probe_$v.bin 44 >> $L 2>&1 || exit 1;
  PROBE_SHORT=1 PROBE_REPS=7 timeout -k 10 120 tools/probes/gemm_probe_$v.bin 44 >> $L 2>&1 || exit 1; done
cat $L
exit $rc
