#!/bin/bash
# Round 6, first GPU call: the new tests, the baseline line, the forced-DP A/B (N > 1 step at N = 1) and the batch sweeps of
# configs 3 and 5.  Every stage under its own timeout; a failed stage ends the call (no GPU step behind a killed one).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c1; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_model_gpu.py tests/test_moe_kernels_gpu.py -q -k "long_capture or generate_matches or predicate or entrance" > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log; [ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" $O/tests.log | head -20; exit $rc; }
run() { # tag, args...
  local tag=$1; shift
  timeout -k 10 420 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -4 $O/$tag.err; return 1; }
  python tools/show_bench.py $O/$tag.json 2>/dev/null | head -20
}
run base --steps 12 --warmup 4 || exit 1
APERTIS_FORCE_DP=1 run forced_dp --steps 12 --warmup 4 --no-cpu-baseline || exit 1
run base2 --steps 12 --warmup 4 --no-cpu-baseline || exit 1
for b in 32 48 64; do run 350m_b$b --config 350m-moe --batch $b --steps 12 --warmup 4 --no-cpu-baseline || break; done
for b in 32 48; do run mm_b$b --config 1.5b-moe-mm --batch $b --steps 12 --warmup 4 --no-cpu-baseline || break; done
echo call1 done
