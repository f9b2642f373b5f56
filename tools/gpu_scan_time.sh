#!/bin/bash
# tools/gpu_scan_time.sh <log> <variants...>: kernel-level times (rocprofv3) of the scan forms at the bench shape for library variants,
# results NOT checked (probe builds may compute garbage)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so; fi
  rm -rf gpurun_out/sv_trace
  echo "== variant $v" >> $L
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sv_trace -- python3 tools/scan_lean_check.py 44 4096 11 16 > gpurun_out/sv.out 2> gpurun_out/sv.err
  f=$(ls gpurun_out/sv_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] || { tail -5 gpurun_out/sv.err; exit 1; }
  python3 - "$f" >> $L <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "scan_lb" in n or "scan_lean_fwd" in n or "scan_lean_bwd" in n:
        m = re.search(r"((scan_[a-z_]+_k)(<[^>]*>)?)", n)
        print("   %-34s calls %4s  avg %9.1f us" % (m.group(1) if m else n[:34], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
rm -rf gpurun_out/sv_trace
cat $L
