#!/bin/bash
# tools/gpu_ab2.sh <log> <alt-lib-suffix|-> <tests -k expr|-> <probe args> <bins...>: tests on the alternate library build
# (apertis_llm_amd/libapertis_hip_<suffix>.so) after a first probe pass has shown the kernels terminate
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; shift
LIBSFX="$1"; shift; K="$1"; shift; ARGS="$1"; shift
for b in "$@"; do echo "== $b (pass 1)" >> $L; timeout -k 10 200 tools/probes/$b.bin $ARGS >> $L 2>&1 || { echo "probe $b failed rc=$?" >> $L; cat $L; exit 1; }; done
if [ "$K" != "-" ]; then
  if [ "$LIBSFX" != "-" ]; then export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$LIBSFX.so; fi
  timeout -k 10 900 python -m pytest tests/test_moe_kernels_gpu.py -m gpu -x -q -k "$K" > gpurun_out/ab_tests.log 2>&1; rc=$?
  tail -15 gpurun_out/ab_tests.log >> $L
  unset APERTIS_HIP_LIB
  if [ $rc -ne 0 ]; then cat $L; exit $rc; fi
fi
for b in "$@"; do echo "== $b (pass 2)" >> $L; timeout -k 10 200 tools/probes/$b.bin $ARGS >> $L 2>&1 || exit 1; done
cat $L
