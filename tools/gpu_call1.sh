#!/bin/bash
# round 3, call 1: GPU test suite, then the tile-walk / TN-ring probe
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests1.log 2>&1; rc=$?
tail -5 gpurun_out/r3_tests1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc $rc)"; exit $rc; fi
PROBE_R3=1 timeout -k 10 240 tools/probes/gemm_probe.bin 44 > gpurun_out/r3_probe_walk_ring.log 2>&1; rc2=$?
cat gpurun_out/r3_probe_walk_ring.log
exit $(( rc != 0 ? rc : rc2 ))
