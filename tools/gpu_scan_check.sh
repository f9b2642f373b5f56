#!/bin/bash
# tools/gpu_scan_check.sh <log> [variants...]: in-tree library over all shapes of tools/scan_lean_check.py, then every variant (and the
# in-tree library again, "-") at the bench shapes under rocprofv3 for kernel-level times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; shift
timeout -k 10 120 python3 tools/scan_lean_check.py 3 257 11 16 >> $L 2>&1 || { echo "canary failed" >> $L; tail -20 $L; exit 1; }
timeout -k 10 420 python3 tools/scan_lean_check.py >> $L 2>&1 || { echo "check failed rc=$?" >> $L; grep "!!\|FAIL\|Error" $L | head; tail -5 $L; exit 1; }
grep "FAILURES" $L | tail -1
for v in - "$@"; do
  if [ "$v" = "-" ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so; fi
  rm -rf gpurun_out/sv_trace
  echo "== variant $v" >> $L
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sv_trace -- python3 tools/scan_lean_check.py 44 4096 11 16 > gpurun_out/sv.out 2> gpurun_out/sv.err || { tail -5 gpurun_out/sv.err; exit 1; }
  grep "lookback fwd\|!!\|FAIL" gpurun_out/sv.out >> $L
  f=$(ls gpurun_out/sv_trace/*/*kernel_stats.csv 2>/dev/null | head -1)
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
sed -n '/== variant -/,$p' $L
