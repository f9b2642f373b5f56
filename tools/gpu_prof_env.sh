#!/bin/bash
# tools/gpu_prof_env.sh <log> "<script + args>" <ENVVAR> <v1> <v2> ...: rocprofv3 --kernel-trace --stats of a script under each value
# of one environment switch, round-robin, twice; the kernels whose names hold $KFILTER (or the ten heaviest) per pass
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; SCRIPT="$2"; VAR=$3; shift 3
for rep in 1 2; do for v in "$@"; do
  export $VAR=$v
  rm -rf gpurun_out/pe_trace
  echo "== $VAR=$v (pass $rep)" >> $L
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pe_trace -- python3 $SCRIPT > gpurun_out/pe.out 2> gpurun_out/pe.err || { tail -5 gpurun_out/pe.err; exit 1; }
  f=$(ls gpurun_out/pe_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] || { echo "no kernel_stats.csv" >> $L; cat $L; exit 1; }
  python3 - "$f" >> $L <<'PY'
import csv, re, sys, os
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
keep = os.environ.get("KFILTER")
for r in ([r for r in rows if any(k in r["Name"] for k in keep.split(","))] if keep else rows[:10]):
    m = re.search(r"([a-z0-9_]+_k)(I\w*?Lb[01])?", r["Name"])
    print("   %-44s calls %4s  avg %9.1f us  total %8.2f ms" % ((m.group(0) if m else r["Name"])[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done; done
rm -rf gpurun_out/pe_trace
cat $L
