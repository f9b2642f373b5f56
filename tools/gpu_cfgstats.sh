#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE configurations (top kernels by time)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for c in 125m 1.5b-moe-mm; do
  rm -rf gpurun_out/cs_trace
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cs_trace -- python3 bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/cs_$c.json 2> gpurun_out/cs_$c.err || { tail -3 gpurun_out/cs_$c.err; exit 1; }
  f=$(ls gpurun_out/cs_trace/*/*kernel_stats.csv | head -1); cp $f gpurun_out/cs_${c}_kernel_stats.csv
  echo "== $c"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = 6
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps
print(f"   kernel time {tot:.1f} ms/step")
for r in rows[:22]:
    print(f"   {r['Name'][:64]:64s} {int(r['Calls'])//steps:5d} {float(r['TotalDurationNs'])/1e6/steps:7.2f} ms {float(r['AverageNs'])/1e3:8.1f} us")
PY
done
rm -rf gpurun_out/cs_trace
