#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_tests6.log 2>&1; rc=$?
tail -4 gpurun_out/r3_tests6.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/gpu_bench.sh r3_bench_b
