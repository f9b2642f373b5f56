#!/bin/bash
# Round 6, eighth GPU call: the look-back forward with LDS-DMA landing buffers (variants 1: xc, z; 2: C, xc, z) against the
# register form: correctness at the bench shape (the suite's look-back tests under each library), then kernel times
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c8; mkdir -p $O
for v in lbdma1 lbdma2; do
  APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so timeout -k 10 300 python -m pytest tests/test_scan_gate_gpu.py -q -x -k "lookback or config4 or stress" > $O/tests_$v.log 2>&1; rc=$?
  echo "$v: $(tail -1 $O/tests_$v.log)"; [ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" $O/tests_$v.log | head -10; exit $rc; }
done
KFILTER=scan_lb bash tools/gpu_prof_libs.sh r6c8_lbdma "tools/prof_scan_gate.py 8 44 1" - lbdma1 lbdma2 > $O/lbdma.log 2>&1 || { tail -5 $O/lbdma.log; exit 1; }
cat gpurun_out/r6c8_lbdma.log
echo call8 done
