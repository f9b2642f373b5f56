#!/bin/bash
# tools/gpu_prof_script.sh <tag> <python script + args...>: rocprofv3 kernel trace + stats (csv) of a script; top rows printed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=$1; shift
rm -rf gpurun_out/${TAG}_trace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python3 "$@" > gpurun_out/${TAG}.log 2> gpurun_out/${TAG}.err || { tail -5 gpurun_out/${TAG}.err; exit 1; }
f=$(ls gpurun_out/${TAG}_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] || { echo "no kernel_stats.csv"; ls -R gpurun_out/${TAG}_trace | head; exit 1; }
cp "$f" gpurun_out/${TAG}_kernel_stats.csv; rm -rf gpurun_out/${TAG}_trace
grep -v amdgpu.ids gpurun_out/${TAG}.log | tail -12
head -14 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
