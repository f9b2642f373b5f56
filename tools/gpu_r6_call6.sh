#!/bin/bash
# Round 6, sixth GPU call: same-wave MFMA / VALU probe; the ring kernel static vs queue in isolation and inside the forced-DP step
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r6c6; mkdir -p $O
timeout -k 10 120 tools/probes/mfma_valu_samewave.bin > $O/samewave.log 2>&1; cat $O/samewave.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for q in 0 1; do
  rm -rf $O/tr; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 tools/prof_expert_mlp_queue.py 10 44 $q > $O/q.out 2>&1 || { tail -3 $O/q.out; exit 1; }
  f=$(ls $O/tr/*/*kernel_stats.csv | head -1)
  echo "== queue $q (pass $rep)" | tee -a $O/queue_iso.log
  python3 - $f <<'PY' | tee -a $O/queue_iso.log
import csv,sys
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:-float(r["TotalDurationNs"]))[:5]:
    n=r["Name"]; k=[x for x in ("nt4r","nt352p","nt256p","nt2x","tn5_k","fillBuffer") if x in n]
    print("   %-10s calls %4s avg %9.1f us" % (k[0] if k else n[:10], r["Calls"], float(r["AverageNs"])/1e3))
PY
done; done; rm -rf $O/tr
run() { local tag=$1; shift
  timeout -k 10 420 python bench.py "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -4 $O/$tag.err; return 1; }
  python tools/show_bench.py $O/$tag.json 2>/dev/null | head -1
}
for i in 1 2; do
  APERTIS_FORCE_DP=1 APERTIS_DP_STATIC_WALK=1 run forced_static_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
  APERTIS_FORCE_DP=1 run forced_queue_$i --steps 12 --warmup 4 --no-cpu-baseline || exit 1
done
echo call6 done
