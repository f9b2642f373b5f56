#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/r3_scan_lt_ab.log; : > $L
echo "== LT64 tests" >> $L
timeout -k 10 400 python -m pytest tests/test_scan_gate_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $L
echo "== LT32 tests" >> $L
APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_lt32.so timeout -k 10 400 python -m pytest tests/test_scan_gate_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -3 >> $L
for v in "" lt32 "" lt32; do echo "== lib ${v:-default}" >> $L; if [ -n "$v" ]; then export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so; else unset APERTIS_HIP_LIB; fi; timeout -k 10 200 python tools/microbench.py scan_gate 2>&1 | grep "1-launch" >> $L; done
cat $L
