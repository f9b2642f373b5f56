#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_moe_kernels_gpu.py tests/test_configs_gpu.py -m gpu -q -x > gpurun_out/plan_tests.log 2>&1; rc=$?; tail -4 gpurun_out/plan_tests.log; [ $rc -eq 0 ] || exit $rc
for which in old new; do
  root=.; [ $which = old ] && root=.ab_old
  timeout -k 10 300 python $root/bench.py --config 350m-moe --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers > gpurun_out/plan_350m_${which}.json 2> gpurun_out/plan_350m_${which}.err || { tail -5 gpurun_out/plan_350m_${which}.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/plan_350m_${which}.json').read().strip().splitlines()[-1]); print('350m-moe $which', round(d['value']), round(d['ms_per_step'],2))"
done
