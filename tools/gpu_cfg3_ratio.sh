#!/bin/bash
# config 3 (350m-moe): kernel time per step (rocprofv3 kernel trace) against the step time of an unprofiled run on the same box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=${1:-r3_cfg3}
timeout -k 10 300 python bench.py --config 350m-moe --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timers > gpurun_out/${TAG}_plain.json 2> gpurun_out/${TAG}_plain.err || { tail -5 gpurun_out/${TAG}_plain.err; exit 1; }
timeout -k 10 300 python bench.py --config 350m-moe --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/${TAG}_timers.json 2> gpurun_out/${TAG}_timers.err || { tail -5 gpurun_out/${TAG}_timers.err; exit 1; }
rm -rf gpurun_out/${TAG}_trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --config 350m-moe --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timers > gpurun_out/${TAG}_trace.json 2> gpurun_out/${TAG}_trace.err || { tail -5 gpurun_out/${TAG}_trace.err; exit 1; }
f=$(ls gpurun_out/${TAG}_trace/*/*kernel_stats.csv | head -1); cp $f gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_trace
python - <<PY
import csv, json
rows = list(csv.DictReader(open("gpurun_out/${TAG}_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
calls = sum(int(r["Calls"]) for r in rows)
steps = 8   # 6 timed + 2 warm-up, all traced
plain = json.loads(open("gpurun_out/${TAG}_plain.json").read().strip().splitlines()[-1])
timers = json.loads(open("gpurun_out/${TAG}_timers.json").read().strip().splitlines()[-1])
prof = json.loads(open("gpurun_out/${TAG}_trace.json").read().strip().splitlines()[-1])
print(f"kernel time {tot / steps:.1f} ms/step in {calls / steps:.0f} launches/step (traced run: {prof['ms_per_step']:.1f} ms/step)")
print(f"unprofiled: {plain['ms_per_step']:.1f} ms/step = {plain['value']:.0f} tokens/s -> kernel/step = {tot / steps / plain['ms_per_step']:.3f};  with event timers: {timers['ms_per_step']:.1f} ms/step = {timers['value']:.0f} tokens/s")
PY
