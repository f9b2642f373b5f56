#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/dffn_tests.log 2>&1; rc=$?; tail -3 gpurun_out/dffn_tests.log; [ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
  timeout -k 10 300 python bench.py --config 125m --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/dffn_125m_$rep.json 2> gpurun_out/dffn_125m_$rep.err || { tail -3 gpurun_out/dffn_125m_$rep.err; exit 1; }
  python tools/show_bench.py gpurun_out/dffn_125m_$rep.json | head -12
done
