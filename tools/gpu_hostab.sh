#!/bin/bash
# A/B of the host-side changes (optimizer tables built once, cheap ptr / stream_ptr): old package copy in .ab_old vs the tree
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/hostab_tests.log 2>&1; rc=$?; tail -3 gpurun_out/hostab_tests.log; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  for which in old new; do
    root=.; [ $which = old ] && root=.ab_old
    timeout -k 10 300 python $root/bench.py --config 350m-moe --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timers > gpurun_out/hostab_350m_${which}_$rep.json 2> gpurun_out/hostab_350m_${which}_$rep.err || { tail -5 gpurun_out/hostab_350m_${which}_$rep.err; exit 1; }
    python -c "import json,sys; d=json.loads(open('gpurun_out/hostab_350m_${which}_$rep.json').read().strip().splitlines()[-1]); print('350m-moe $which $rep', round(d['value']), round(d['ms_per_step'],2))"
  done
done
for which in old new; do
  root=.; [ $which = old ] && root=.ab_old
  timeout -k 10 400 python $root/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-timers > gpurun_out/hostab_1.5b_${which}.json 2> gpurun_out/hostab_1.5b_${which}.err || { tail -5 gpurun_out/hostab_1.5b_${which}.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/hostab_1.5b_${which}.json').read().strip().splitlines()[-1]); print('1.5b-moe $which', round(d['value']), round(d['ms_per_step'],2))"
done
timeout -k 10 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/hostab_1.5b_timers.json 2> gpurun_out/hostab_1.5b_timers.err || { tail -5 gpurun_out/hostab_1.5b_timers.err; exit 1; }
python tools/show_bench.py gpurun_out/hostab_1.5b_timers.json | head -40
