#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench workload (3 timed steps + 1 warm-up)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=${1:-r3}; shift
rm -rf gpurun_out/${TAG}_trace
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers "$@" > gpurun_out/${TAG}_trace.json 2> gpurun_out/${TAG}_trace.err || { tail -5 gpurun_out/${TAG}_trace.err; exit 1; }
f=$(ls gpurun_out/${TAG}_trace/*/*kernel_stats.csv | head -1); cp $f gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_trace
tail -1 gpurun_out/${TAG}_trace.json | cut -c1-200
head -30 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150
