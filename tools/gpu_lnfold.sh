#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_moe_kernels_gpu.py tests/test_model_gpu.py tests/test_configs_gpu.py tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/lnfold_tests.log 2>&1; rc=$?; tail -2 gpurun_out/lnfold_tests.log; [ $rc -eq 0 ] || exit $rc
for v in prev new prev new; do
  if [ $v = new ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_prev.so; fi
  rm -rf gpurun_out/ln_trace
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ln_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > gpurun_out/lnf_$v.json 2> gpurun_out/lnf_$v.err || { tail -3 gpurun_out/lnf_$v.err; exit 1; }
  f=$(ls gpurun_out/ln_trace/*/*kernel_stats.csv | head -1)
  echo "== $v"; python - "$f" gpurun_out/lnf_$v.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / 4
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"   kernel time {tot:.1f} ms/step, step under the profiler {d['ms_per_step']:.1f} ms")
for r in rows:
    if "layernorm_bwd_k" in r["Name"] or ("ln_fold_k" in r["Name"] and "gather" not in r["Name"]):
        print(f"   {r['Name'][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:6.1f} us total/step {float(r['TotalDurationNs'])/4e6:5.2f} ms")
PY
done
rm -rf gpurun_out/ln_trace
