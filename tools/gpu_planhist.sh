#!/bin/bash
# dispatch plan: histogram without global atomics, slot state from LDS - tests, then kernel times on both MoE configs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_moe_kernels_gpu.py tests/test_configs_gpu.py tests/test_model_gpu.py -m gpu -q -x > gpurun_out/planhist_tests.log 2>&1; rc=$?; tail -2 gpurun_out/planhist_tests.log; [ $rc -eq 0 ] || exit $rc
for c in 350m-moe 1.5b-moe; do
  rm -rf gpurun_out/ph_trace
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ph_trace -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > gpurun_out/ph_$c.json 2> gpurun_out/ph_$c.err || { tail -3 gpurun_out/ph_$c.err; exit 1; }
  f=$(ls gpurun_out/ph_trace/*/*kernel_stats.csv | head -1)
  echo "== $c"; python - "$f" gpurun_out/ph_$c.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / 4
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"   kernel time {tot:.1f} ms/step, step under the profiler {d['ms_per_step']:.1f} ms")
for r in rows:
    if "plan_" in r["Name"]:
        print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:6.1f} us total/step {float(r['TotalDurationNs'])/4e6:5.2f} ms")
PY
done
rm -rf gpurun_out/ph_trace
