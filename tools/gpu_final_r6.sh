#!/bin/bash
# End-of-round-6 artefacts (gpurun_out/<tag>_*), each stage on the box of its call:  tools/gpu_final_r6.sh <tag> <stage>
#   1  the whole GPU suite + smoke(), the default bench line (50 timed steps, cpu_baseline), the kernel trace of the same workload
#   2  PMC passes on the per-layer launches of the three MoE configurations (traffic files for bench.py)
#   3  the other BASELINE configurations (with their traffic files in place), decode
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; TAG=$1
if [ "$2" = "1" ]; then
  rm -f gpurun_out/parity_report.jsonl
  timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_suite.log 2>&1; rc=$?
  tail -3 gpurun_out/${TAG}_suite.log; [ $rc -eq 0 ] || { grep -n "Error\|FAILED" gpurun_out/${TAG}_suite.log | head; exit $rc; }
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
  python tools/show_bench.py gpurun_out/${TAG}_bench.json 2>/dev/null | head -20
  bash tools/gpu_profile.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1 || { tail -5 gpurun_out/${TAG}_profile.log; exit 1; }
  head -2 gpurun_out/${TAG}_profile.log | cut -c1-200
elif [ "$2" = "2" ]; then
  for cfg in "1.5b-moe 44" "350m-moe 72" "1.5b-moe-mm 72"; do set -- $cfg
    bash tools/run_pmc.sh gpurun_out/${TAG}_pmc_$1 benchmix $2 $1 > gpurun_out/${TAG}_pmc_$1.log 2>&1 || { tail -5 gpurun_out/${TAG}_pmc_$1.log; exit 1; }
    cp gpurun_out/${TAG}_pmc_$1/traffic.json gpurun_out/${TAG}_pmc_traffic_$1_b$2.json; cp gpurun_out/${TAG}_pmc_$1/summary.txt gpurun_out/${TAG}_pmc_summary_$1_b$2.txt
    rm -rf gpurun_out/${TAG}_pmc_$1
    python3 -c "import json;d=json.load(open('gpurun_out/${TAG}_pmc_traffic_$1_b$2.json'));print('$1',{k:round(v['traffic_bytes_per_call']/1e6,1) for k,v in d.items() if not k.startswith('_')})"
  done
else
  for c in 125m 350m-moe 1.5b-moe-mm; do
    timeout -k 10 600 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_$c.json 2> gpurun_out/${TAG}_$c.err || { tail -5 gpurun_out/${TAG}_$c.err; exit 1; }
    python tools/show_bench.py gpurun_out/${TAG}_$c.json 2>/dev/null | head -7
  done
  timeout -k 10 500 python bench.py --decode > gpurun_out/${TAG}_decode.json 2> gpurun_out/${TAG}_decode.err || { tail -5 gpurun_out/${TAG}_decode.err; exit 1; }
  tail -1 gpurun_out/${TAG}_decode.json | cut -c1-300
fi
