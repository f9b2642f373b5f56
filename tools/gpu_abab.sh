#!/bin/bash
# old / new / old / new of the default bench line (old = package copy in .ab_old)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for rep in 1 2; do for which in old new; do
  root=.; [ $which = old ] && root=.ab_old
  timeout -k 10 400 python $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/abab_${which}_$rep.json 2> gpurun_out/abab_${which}_$rep.err || { tail -5 gpurun_out/abab_${which}_$rep.err; exit 1; }
  echo "== $which $rep"; python tools/show_bench.py gpurun_out/abab_${which}_$rep.json | grep -E "tok/s|gemm_nt  |gemm_tn  |gemm_nt |gemm_tn " | grep -v dense
done; done
