#!/bin/bash
# tools/gpu_env_ab.sh <tag> <ENVVAR> <valueA> <valueB> [bench args...]: the default bench line under ENVVAR=valueA / valueB,
# alternating, twice each (same box, same call); one summary line per run
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=$1; VAR=$2; VA=$3; VB=$4; shift 4
for rep in 1 2; do for v in "$VA" "$VB"; do
  export $VAR=$v
  timeout -k 10 500 python bench.py --steps 16 --warmup 4 --no-cpu-baseline "$@" > gpurun_out/${TAG}_${v}_${rep}.json 2> gpurun_out/${TAG}.err || { tail -5 gpurun_out/${TAG}.err; exit 1; }
  echo "== $VAR=$v (pass $rep)"; python tools/show_bench.py gpurun_out/${TAG}_${v}_${rep}.json 2>/dev/null | head -${SHOW:-12}
done; done
