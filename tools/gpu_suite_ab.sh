#!/bin/bash
# tools/gpu_suite_ab.sh <tag> <pytest -k expr | all> <old-lib suffix | -> [bench args...]: the GPU tests (or a -k subset), then the
# default bench line with the in-tree library and with apertis_llm_amd/libapertis_hip_<suffix>.so, alternating, twice each
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=$1; K="$2"; OLD=$3; shift 3
if [ "$K" = "all" ]; then KARGS=""; else KARGS="-k"; fi
timeout -k 10 1000 python -m pytest tests -m gpu -x -q ${KARGS:+-k "$K"} > gpurun_out/${TAG}_tests.log 2>&1; rc=$?
tail -6 gpurun_out/${TAG}_tests.log
if [ $rc -ne 0 ]; then grep -n "Error\|assert\|FAILED" gpurun_out/${TAG}_tests.log | head -30; exit $rc; fi
[ "$OLD" = "-" ] && exit 0
for rep in 1 2; do for v in new old; do
  if [ $v = old ]; then export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$OLD.so; else unset APERTIS_HIP_LIB; fi
  timeout -k 10 500 python bench.py --steps 16 --warmup 4 --no-cpu-baseline "$@" > gpurun_out/${TAG}_${v}_${rep}.json 2> gpurun_out/${TAG}.err || { tail -5 gpurun_out/${TAG}.err; exit 1; }
  echo "== $v (pass $rep)"; python tools/show_bench.py gpurun_out/${TAG}_${v}_${rep}.json | grep "tok/s\|grouped_gemm_nt \|grouped_gemm_tn \|scan_gate"
done; done
