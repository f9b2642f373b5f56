#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_scan_gate_gpu.py -m gpu -x -q > gpurun_out/scanbar_tests.log 2>&1; rc=$?; tail -2 gpurun_out/scanbar_tests.log; [ $rc -eq 0 ] || exit $rc
export MB_BATCH=44
for which in old new old new; do
  if [ $which = old ]; then export APERTIS_HIP_LIB=$PWD/.ab_old/apertis_llm_amd/libapertis_hip.so; else unset APERTIS_HIP_LIB; fi
  rm -rf gpurun_out/sp_trace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_trace -- python3 tools/microbench.py scan_gate1 > gpurun_out/scanbar_mb_$which.log 2>&1 || { tail -5 gpurun_out/scanbar_mb_$which.log; exit 1; }
  f=$(ls gpurun_out/sp_trace/*/*kernel_stats.csv | head -1)
  echo "== $which"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "scan_gate_bwd" in r["Name"] and "Li1EEE" in r["Name"]:
        print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us min {float(r['MinNs'])/1e3:7.1f} us")
PY
done
rm -rf gpurun_out/sp_trace
