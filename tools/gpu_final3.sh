#!/bin/bash
# end-of-round records (late round 3): whole GPU suite + parity report, smoke, the default bench line, the other configs, decode
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_tests_final3.log 2>&1; rc=$?
tail -3 gpurun_out/r3_tests_final3.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r3f_bench_1.5b-moe_b44.json 2> gpurun_out/r3f_bench_1.5b-moe_b44.err || { tail -3 gpurun_out/r3f_bench_1.5b-moe_b44.err; exit 1; }
python tools/show_bench.py gpurun_out/r3f_bench_1.5b-moe_b44.json
for c in 125m 350m-moe 1.5b-moe-mm; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3f_bench_$c.json 2> gpurun_out/r3f_bench_$c.err || { tail -3 gpurun_out/r3f_bench_$c.err; exit 1; }
  python tools/show_bench.py gpurun_out/r3f_bench_$c.json | head -8
done
timeout -k 10 400 python bench.py --decode --decode-dtype bf16 > gpurun_out/r3f_decode_bf16.json 2> gpurun_out/r3f_decode_bf16.err || { tail -3 gpurun_out/r3f_decode_bf16.err; exit 1; }
grep -E "decode B" gpurun_out/r3f_decode_bf16.err | head
