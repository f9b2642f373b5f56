#!/bin/bash
# tools/gpu_scan_lb.sh <log> [variant suffixes...]: tools/scan_lean_check.py over its shape list with the in-tree library (staged /
# lean / look-back forms against each other), then the two seq-4096 bench shapes for every alternate build
# apertis_llm_amd/libapertis_hip_<suffix>.so (APERTIS_HIP_LIB)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; shift
echo "== in-tree, all shapes" >> $L
timeout -k 10 420 python3 tools/scan_lean_check.py >> $L 2>&1 || { echo "check failed rc=$?" >> $L; tail -40 $L; exit 1; }
for v in "$@"; do
  export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so
  for shp in "44 4096 11 16" "16 4096 4 16"; do
    echo "== variant $v shape $shp" >> $L
    timeout -k 10 200 python3 tools/scan_lean_check.py $shp 2>&1 | grep "fwd best\|!!\|FAIL\|error word" >> $L || { echo "variant failed" >> $L; tail -40 $L; exit 1; }
  done
done
grep -v "^   \(lean\|lookback\) *d\|amdgpu.ids" $L | tail -120
