#!/bin/bash
# tools/gpu_prof_libs.sh <log> <script + args (quoted)> <lib suffix | -> ...: rocprofv3 --kernel-trace --stats of a script under each
# library build ("-" = in-tree), round-robin, twice; per-kernel average times of the GEMM kernels printed per pass
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; SCRIPT="$2"; shift 2
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = "-" ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so; fi
  rm -rf gpurun_out/pl_trace
  echo "== lib $v (pass $rep)" >> $L
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pl_trace -- python3 $SCRIPT > gpurun_out/pl.out 2> gpurun_out/pl.err || { tail -5 gpurun_out/pl.err; exit 1; }
  f=$(ls gpurun_out/pl_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] || { echo "no kernel_stats.csv" >> $L; cat $L; exit 1; }
  python3 - "$f" >> $L <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
import os
keep = os.environ.get("KFILTER")      # (KFILTER=substring: every kernel whose name holds it instead of the ten heaviest)
for r in ([r for r in rows if keep in r["Name"]] if keep else rows[:10]):
    m = re.search(r"([a-z0-9_]+_k)\b", r["Name"])
    print("   %-34s calls %4s  avg %9.1f us  total %8.2f ms" % (m.group(1) if m else r["Name"][:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done; done
rm -rf gpurun_out/pl_trace
cat $L
