#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_moe_kernels_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/mulpre_tests.log 2>&1; rc=$?; tail -2 gpurun_out/mulpre_tests.log; [ $rc -eq 0 ] || exit $rc
for which in old new; do
  root=.; [ $which = old ] && root=.ab_old
  timeout -k 10 400 python $root/bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/mulpre_1.5b_${which}.json 2> gpurun_out/mulpre_1.5b_${which}.err || { tail -5 gpurun_out/mulpre_1.5b_${which}.err; exit 1; }
  python tools/show_bench.py gpurun_out/mulpre_1.5b_${which}.json | head -7
done
