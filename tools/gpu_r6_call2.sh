#!/bin/bash
# Round 6, second GPU call: the batch sweeps of configs 3 and 5 continued towards the memory limit, the kernel trace of the
# forced-DP step (the N > 1 step at N = 1), and the default line with the bounded cpu_baseline.
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c2; mkdir -p $O
run() { local tag=$1; shift
  timeout -k 10 420 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -4 $O/$tag.err; return 1; }
  python tools/show_bench.py $O/$tag.json 2>/dev/null | head -8; grep -h 'peak mem' $O/$tag.err | tail -1
}
for b in 72 76; do run 350m_b$b --config 350m-moe --batch $b --steps 12 --warmup 4 --no-cpu-baseline || break; done
for b in 64 72; do run mm_b$b --config 1.5b-moe-mm --batch $b --steps 12 --warmup 4 --no-cpu-baseline || break; done
run base --steps 12 --warmup 4 || exit 1
python -c "import json;d=json.loads(open('$O/base.json').read().strip().splitlines()[-1]);print({k:v for k,v in d['cpu_baseline'].items() if k!='sample'})"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
APERTIS_FORCE_DP=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/forced_dp_trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timers > $O/forced_dp_trace.log 2>&1 || { tail -5 $O/forced_dp_trace.log; exit 1; }
f=$(ls $O/forced_dp_trace/*/*kernel_stats.csv | head -1); cp $f $O/forced_dp_kernel_stats.csv; head -12 $O/forced_dp_kernel_stats.csv | cut -c1-160
rm -rf $O/forced_dp_trace
echo call2 done
