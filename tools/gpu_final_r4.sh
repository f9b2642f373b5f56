#!/bin/bash
# End-of-round artefacts in two calls (argument 1 or 2), each on one box: 1 = GPU suite + smoke, default bench line (50 steps,
# cpu_baseline), kernel trace of the same workload; 2 = PMC passes, the other BASELINE configurations, decode.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ "$1" = "1" ]; then
  bash tools/gpu_suite.sh || exit 1
  timeout -k 10 700 python bench.py > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err || { tail -5 gpurun_out/r4_final_bench.err; exit 1; }
  python tools/show_bench.py gpurun_out/r4_final_bench.json | head -9
  bash tools/gpu_profile.sh r4_final > gpurun_out/r4_final_profile.log 2>&1 || { tail -5 gpurun_out/r4_final_profile.log; exit 1; }
  head -2 gpurun_out/r4_final_profile.log | cut -c1-200
else
  bash tools/run_pmc.sh gpurun_out/pmc_r4 benchmix 44 > gpurun_out/r4_final_pmc.log 2>&1 || { tail -5 gpurun_out/r4_final_pmc.log; exit 1; }
  tail -3 gpurun_out/r4_final_pmc.log
  for c in 125m 350m-moe 1.5b-moe-mm; do
    timeout -k 10 500 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4_final_$c.json 2> gpurun_out/r4_final_$c.err || { tail -5 gpurun_out/r4_final_$c.err; exit 1; }
    python tools/show_bench.py gpurun_out/r4_final_$c.json | head -1
  done
  timeout -k 10 500 python bench.py --decode > gpurun_out/r4_final_decode.json 2> gpurun_out/r4_final_decode.err || { tail -5 gpurun_out/r4_final_decode.err; exit 1; }
  tail -2 gpurun_out/r4_final_decode.json | cut -c1-300
fi
