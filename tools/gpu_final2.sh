#!/bin/bash
# end-of-round records: the whole GPU suite with its parity report, the other BASELINE configurations
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_tests_final.log 2>&1; rc=$?
tail -3 gpurun_out/r3_tests_final.log
if [ $rc -ne 0 ]; then exit $rc; fi
for c in 125m 350m-moe 1.5b-moe-mm; do
  timeout -k 10 300 python bench.py --config $c --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3_bench_$c.json 2> gpurun_out/r3_bench_$c.err || { tail -3 gpurun_out/r3_bench_$c.err; exit 1; }
  python tools/show_bench.py gpurun_out/r3_bench_$c.json | head -8
done
