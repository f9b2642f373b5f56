#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "generate or decode" > gpurun_out/r3_tests7.log 2>&1; rc=$?
tail -6 gpurun_out/r3_tests7.log; grep -i "decode step graph" gpurun_out/r3_tests7.log | head -3
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 500 python bench.py --decode --decode-dtype bf16 > gpurun_out/r3_decode_bf16.json 2> gpurun_out/r3_decode_bf16.err; rc2=$?
grep -E "decode B|graph" gpurun_out/r3_decode_bf16.err | head; tail -2 gpurun_out/r3_decode_bf16.err
exit $(( rc != 0 ? rc : rc2 ))
