#!/bin/bash
# Round 6, fifth GPU call: the ring kernel's queue with ONE atomic per work-group: tests, forced-DP A/B, kernel trace
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_moe_kernels_gpu.py tests/test_dp_gpu.py -q -x > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log; [ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" $O/tests.log | head -20; exit $rc; }
run() { local tag=$1; shift
  timeout -k 10 420 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -4 $O/$tag.err; return 1; }
  python tools/show_bench.py $O/$tag.json 2>/dev/null | head -1
}
for i in 1 2; do
  run base_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
  APERTIS_FORCE_DP=1 run forced_dp_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
APERTIS_FORCE_DP=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/forced_dp_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > $O/forced_dp_trace.log 2>&1 || { tail -5 $O/forced_dp_trace.log; exit 1; }
f=$(ls $O/forced_dp_trace/*/*kernel_stats.csv | head -1); cp $f $O/forced_dp_kernel_stats.csv; rm -rf $O/forced_dp_trace
head -5 $O/forced_dp_kernel_stats.csv | cut -c1-140
echo call5 done
