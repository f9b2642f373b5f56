#!/bin/bash
# attribution: the same python, three builds of the library (APERTIS_HIP_LIB)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for which in preplan planonly tree; do
  if [ $which = tree ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$which.so; fi
  timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abc_${which}_$rep.json 2> gpurun_out/abc_${which}_$rep.err || { tail -5 gpurun_out/abc_${which}_$rep.err; exit 1; }
  echo "== $which $rep"; python tools/show_bench.py gpurun_out/abc_${which}_$rep.json | grep -E "tok/s|gemm_nt |gemm_tn " | grep -v dense
done; done
