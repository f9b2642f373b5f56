#!/bin/bash
# pipelined scan backward: parity tests, then kernel times with the switch off / on (rocprofv3 kernel trace of the bench shape)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then timeout -k 10 500 python -m pytest tests/test_scan_gate_gpu.py -m gpu -x -q > gpurun_out/scanpipe_tests.log 2>&1; rc=$?; tail -5 gpurun_out/scanpipe_tests.log; [ $rc -eq 0 ] || exit $rc; fi
export MB_BATCH=44
for sw in 0 1 0 1; do
  export APERTIS_SCAN_BWD_PIPE=$sw
  rm -rf gpurun_out/sp_trace
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_trace -- python3 tools/microbench.py scan_gate1 > gpurun_out/scanpipe_mb_$sw.log 2>&1 || { tail -5 gpurun_out/scanpipe_mb_$sw.log; exit 1; }
  f=$(ls gpurun_out/sp_trace/*/*kernel_stats.csv | head -1)
  echo "== APERTIS_SCAN_BWD_PIPE=$sw"; grep "1-launch" gpurun_out/scanpipe_mb_$sw.log | cut -c1-200
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "scan_gate" in r["Name"]:
        print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} us min {float(r['MinNs'])/1e3:7.1f} us")
PY
done
rm -rf gpurun_out/sp_trace
