#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_moe_kernels_gpu.py tests/test_configs_gpu.py -m gpu -q -x > gpurun_out/plan2_tests.log 2>&1; rc=$?; tail -2 gpurun_out/plan2_tests.log; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for which in preplan tree; do
  if [ $which = tree ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$which.so; fi
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timers > gpurun_out/plan2_${which}_$rep.json 2> gpurun_out/plan2_${which}_$rep.err || { tail -5 gpurun_out/plan2_${which}_$rep.err; exit 1; }
  echo "== 1.5b $which $rep"; python tools/show_bench.py gpurun_out/plan2_${which}_$rep.json | grep -E "tok/s"
done; done
for which in preplan tree preplan tree; do
  if [ $which = tree ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$which.so; fi
  timeout -k 10 300 python bench.py --config 350m-moe --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers > gpurun_out/plan2_c3_${which}.json 2> gpurun_out/plan2_c3_${which}.err || { tail -5 gpurun_out/plan2_c3_${which}.err; exit 1; }
  echo "== 350m-moe $which"; python tools/show_bench.py gpurun_out/plan2_c3_${which}.json | grep -E "tok/s"
done
