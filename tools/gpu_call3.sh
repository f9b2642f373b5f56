#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_moe_kernels_gpu.py -m gpu -x -q > gpurun_out/r3_tests3.log 2>&1; rc=$?
tail -4 gpurun_out/r3_tests3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed (rc $rc)"; exit $rc; fi
L=gpurun_out/r3_probe_tn_epi.log; : > $L
PROBE_R3=1 PROBE_ONLY=tn PROBE_REPS=9 timeout -k 10 120 tools/probes/gemm_probe_base.bin 44 >> $L 2>&1 || exit 1
for v in base nopk nohash nogelu neither; do echo "== $v" >> $L; PROBE_R3=1 PROBE_ONLY=walk PROBE_WALKS="8,4" PROBE_REPS=9 timeout -k 10 120 tools/probes/gemm_probe_$v.bin 44 >> $L 2>&1 || exit 1; done
cat $L
cd /tmp; export TMPDIR=/tmp; rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQC_" | head -40 > $GRAFT_REPO_ROOT/gpurun_out/r3_counters_list.txt; cat $GRAFT_REPO_ROOT/gpurun_out/r3_counters_list.txt | cut -c1-160 | head -30
exit $rc
