#!/bin/bash
# Round 6, seventh GPU call: the interleaved saved-gradient forward (grouped_gemm_nt2i_k): correctness, then its time
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c7; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_moe_kernels_gpu.py -q -x -k "ring_kernel or expert_mlp or saved or full_size" > $O/tests.log 2>&1; rc=$?
tail -3 $O/tests.log; [ $rc -eq 0 ] || { grep -n "Error\|FAILED\|assert" $O/tests.log | head -20; exit $rc; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for q in 0 1; do
  rm -rf $O/tr; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 tools/prof_expert_mlp_queue.py 10 44 $q > $O/q.out 2>&1 || { tail -3 $O/q.out; exit 1; }
  f=$(ls $O/tr/*/*kernel_stats.csv | head -1)
  echo "== queue $q (pass $rep)  [0: fc1 forward on nt2i, fused fc2 dgrad on nt4r; 1: both on nt4r under the queue]" | tee -a $O/nt2i_iso.log
  python3 - $f <<'PY' | tee -a $O/nt2i_iso.log
import csv,sys
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:-float(r["TotalDurationNs"]))[:6]:
    n=r["Name"]; k=[x for x in ("nt2i","nt4r","nt352p","nt256p","nt2x","tn5_k","fillBuffer") if x in n]
    print("   %-10s calls %4s avg %9.1f us" % (k[0] if k else n[:10], r["Calls"], float(r["AverageNs"])/1e3))
PY
done; done; rm -rf $O/tr
echo call7 done
