#!/bin/bash
# bench line + kernel stats of the default workload
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=${1:-r3}
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
python tools/show_bench.py gpurun_out/${TAG}_bench.json
