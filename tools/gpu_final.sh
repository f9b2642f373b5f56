#!/bin/bash
# End-of-round artefacts (gpurun_out/<tag>_*), each stage on the box of its call:  tools/gpu_final.sh <tag> <stage>
#   1  the whole GPU suite + smoke(), the default bench line (50 timed steps, cpu_baseline), the kernel trace of the same workload
#   2  PMC passes on the per-layer launches of the bench workload (traffic file for bench.py), the other BASELINE configurations, decode
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; TAG=$1
if [ "$2" = "1" ]; then
  rm -f gpurun_out/parity_report.jsonl
  timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_suite.log 2>&1; rc=$?
  tail -3 gpurun_out/${TAG}_suite.log; [ $rc -eq 0 ] || { grep -n "Error\|FAILED" gpurun_out/${TAG}_suite.log | head; exit $rc; }
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout -k 10 700 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
  python tools/show_bench.py gpurun_out/${TAG}_bench.json 2>/dev/null | head -12
  bash tools/gpu_profile.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}_profile.log; exit 1; }
  head -2 gpurun_out/${TAG}_profile.log | cut -c1-200
else
  bash tools/run_pmc.sh gpurun_out/${TAG}_pmc benchmix 44 > gpurun_out/${TAG}_pmc.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc.log; exit 1; }
  tail -3 gpurun_out/${TAG}_pmc.log
  for c in 125m 350m-moe 1.5b-moe-mm; do
    timeout -k 10 500 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_$c.json 2> gpurun_out/${TAG}_$c.err || { tail -5 gpurun_out/${TAG}_$c.err; exit 1; }
    python tools/show_bench.py gpurun_out/${TAG}_$c.json 2>/dev/null | head -1
  done
  timeout -k 10 500 python bench.py --decode > gpurun_out/${TAG}_decode.json 2> gpurun_out/${TAG}_decode.err || { tail -5 gpurun_out/${TAG}_decode.err; exit 1; }
  tail -2 gpurun_out/${TAG}_decode.json | cut -c1-300
fi
