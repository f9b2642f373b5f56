#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 400 python tools/host_profile.py --config 350m-moe > gpurun_out/hostprof2_350m.log 2>&1 || { tail -20 gpurun_out/hostprof2_350m.log; exit 1; }
grep -n "^== 3 steps" gpurun_out/hostprof2_350m.log
