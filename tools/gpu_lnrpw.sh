#!/bin/bash
# LayerNorm backward: rows per wave 8 / 16 / 32 (partial rows per call 5632 / 2816 / 1408): kernel + fold times inside the step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in 8 16 32 8 16 32; do
  if [ $v = 8 ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_rpw$v.so; fi
  rm -rf gpurun_out/ln_trace
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ln_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > gpurun_out/ln_$v.json 2> gpurun_out/ln_$v.err || { tail -3 gpurun_out/ln_$v.err; exit 1; }
  f=$(ls gpurun_out/ln_trace/*/*kernel_stats.csv | head -1)
  echo "== rows per wave $v"; python - "$f" gpurun_out/ln_$v.json <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / 4
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"   kernel time {tot:.1f} ms/step, step under the profiler {d['ms_per_step']:.1f} ms")
for r in rows:
    if "layernorm_bwd_k" in r["Name"] or "ln_fold_k" in r["Name"] and "gather" not in r["Name"]:
        print(f"   {r['Name'][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:6.1f} us total/step {float(r['TotalDurationNs'])/4e6:5.2f} ms")
PY
done
rm -rf gpurun_out/ln_trace
