#!/bin/bash
# One A/B harness for probe binaries (replaces the round-3 one-off gpu_call*.sh scripts).
#   tools/gpu_ab.sh <log-name> <tests-k-expr|-> <probe-args> <bin1> <bin2> ...   (probe binaries under tools/probes/, run in
#   the order given, twice; env PROBE_* is passed through).  Build the variants with the -D switches their names say.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; shift
K="$1"; shift
ARGS="$1"; shift
if [ "$K" != "-" ]; then
  timeout -k 10 900 python -m pytest tests/test_moe_kernels_gpu.py -m gpu -x -q -k "$K" > gpurun_out/ab_tests.log 2>&1; rc=$?
  tail -4 gpurun_out/ab_tests.log | tee -a $L
  if [ $rc -ne 0 ]; then exit $rc; fi
fi
for rep in 1 2; do for b in "$@"; do echo "== $b (pass $rep)" >> $L; timeout -k 10 300 tools/probes/$b.bin $ARGS >> $L 2>&1 || exit 1; done; done
cat $L
