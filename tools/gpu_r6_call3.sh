#!/bin/bash
# Round 6, third GPU call: the whole GPU suite on the split ops package, the forced-DP A/B with the queue-driven kernels (the N > 1
# step at N = 1) + its kernel trace, config 3's scan forms at the new batch, PMC traffic passes for configs 3 and 5.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c3; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/suite.log 2>&1; rc=$?
tail -3 $O/suite.log; [ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" $O/suite.log | head -20; exit $rc; }
run() { local tag=$1; shift
  timeout -k 10 420 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -4 $O/$tag.err; return 1; }
  python tools/show_bench.py $O/$tag.json 2>/dev/null | head -1
}
for i in 1 2; do
  run base_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
  APERTIS_FORCE_DP=1 run forced_dp_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
done
run 350m_default --config 350m-moe --steps 12 --warmup 4 --no-cpu-baseline || exit 1
APERTIS_SCAN_LOOKBACK=all run 350m_lookback --config 350m-moe --steps 12 --warmup 4 --no-cpu-baseline || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
APERTIS_FORCE_DP=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/forced_dp_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > $O/forced_dp_trace.log 2>&1 || { tail -5 $O/forced_dp_trace.log; exit 1; }
f=$(ls $O/forced_dp_trace/*/*kernel_stats.csv | head -1); cp $f $O/forced_dp_kernel_stats.csv; rm -rf $O/forced_dp_trace
head -6 $O/forced_dp_kernel_stats.csv | cut -c1-140
bash tools/run_pmc.sh $O/pmc_350m benchmix 72 350m-moe > $O/pmc_350m.log 2>&1 || { tail -5 $O/pmc_350m.log; exit 1; }
tail -30 $O/pmc_350m.log | grep -A3 'traffic_bytes_per_call' | head -30
bash tools/run_pmc.sh $O/pmc_mm benchmix 72 1.5b-moe-mm > $O/pmc_mm.log 2>&1 || { tail -5 $O/pmc_mm.log; exit 1; }
for d in pmc_350m pmc_mm; do rm -rf $O/$d/sq1 $O/$d/sq2 $O/$d/tcc1 $O/$d/fetch $O/$d/write; done
echo call3 done
