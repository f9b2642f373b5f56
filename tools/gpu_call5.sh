#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r3_tests5.log 2>&1; rc=$?
tail -5 gpurun_out/r3_tests5.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc $rc)"; exit $rc; fi
timeout -k 10 500 python bench.py --decode --decode-dtype bf16 > gpurun_out/r3_decode_bf16.json 2> gpurun_out/r3_decode_bf16.err; rc2=$?
grep "decode B" gpurun_out/r3_decode_bf16.err; tail -2 gpurun_out/r3_decode_bf16.err
exit $(( rc != 0 ? rc : rc2 ))
