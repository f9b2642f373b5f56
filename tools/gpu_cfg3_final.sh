#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do
  timeout -k 10 300 python bench.py --config 350m-moe --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timers > gpurun_out/c3f_plain_$rep.json 2> gpurun_out/c3f_plain_$rep.err || { tail -3 gpurun_out/c3f_plain_$rep.err; exit 1; }
  python tools/show_bench.py gpurun_out/c3f_plain_$rep.json | head -1
done
timeout -k 10 300 python bench.py --config 350m-moe --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/c3f_timers.json 2> gpurun_out/c3f_timers.err || exit 1
python tools/show_bench.py gpurun_out/c3f_timers.json | head -8
