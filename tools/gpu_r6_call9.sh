#!/bin/bash
# Round 6, ninth GPU call: did the queue code cost the ring kernel's STATIC walk anything?  The library with the pre-queue
# grouped_gemm.hip (commit 7a891b5) against the in-tree one: expert MLP in isolation, then the bench step, alternating.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c9; mkdir -p $O
KFILTER=gemm bash tools/gpu_prof_libs.sh r6c9_oldgg "tools/prof_expert_mlp.py 10 44" - oldgg > $O/iso.log 2>&1 || { tail -5 $O/iso.log; exit 1; }
cat gpurun_out/r6c9_oldgg.log
run() { local tag=$1; shift
  timeout -k 10 420 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -4 $O/$tag.err; return 1; }
  python tools/show_bench.py $O/$tag.json 2>/dev/null | grep -E 'tok/s|apertis_grouped_gemm_nt  |apertis_grouped_gemm_tn  '
}
for i in 1 2; do
  run new_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
  APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_oldgg.so run old_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
done
echo call9 done
