#!/bin/bash
# host-side profile of the H=256 family's step and of the bench config
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python tools/host_profile.py --config 350m-moe > gpurun_out/hostprof_350m.log 2>&1 || { tail -20 gpurun_out/hostprof_350m.log; exit 1; }
grep -n "^== 3 steps" gpurun_out/hostprof_350m.log
timeout -k 10 400 python tools/host_profile.py --config 1.5b-moe --batch 16 --layers 8 > gpurun_out/hostprof_1.5b.log 2>&1 || { tail -20 gpurun_out/hostprof_1.5b.log; exit 1; }
grep -n "^== 3 steps" gpurun_out/hostprof_1.5b.log
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x > gpurun_out/r3_model_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r3_model_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
