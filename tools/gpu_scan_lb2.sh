#!/bin/bash
# tools/gpu_scan_lb2.sh <log> <variant for the all-shapes check> [variants for the two bench shapes...]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; FULL=$2; shift 2
export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$FULL.so
echo "== $FULL, canary" >> $L
timeout -k 10 120 python3 tools/scan_lean_check.py 3 257 11 16 >> $L 2>&1 || { echo "canary failed rc=$?" >> $L; grep "!!\|FAIL\|Error\|fault" $L | head; tail -5 $L; exit 1; }
echo "== $FULL, all shapes" >> $L
timeout -k 10 420 python3 tools/scan_lean_check.py >> $L 2>&1 || { echo "check failed rc=$?" >> $L; grep "!!\|FAIL\|Error" $L | head; tail -5 $L; exit 1; }
for v in "$@"; do
  if [ "$v" = "-" ]; then unset APERTIS_HIP_LIB; else export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so; fi
  for shp in "44 4096 11 16" "32 2048 14 16"; do
    echo "== variant $v shape $shp" >> $L
    timeout -k 10 200 python3 tools/scan_lean_check.py $shp 2>&1 | grep "fwd best\|!!\|FAIL\|error word" >> $L || { echo "variant failed" >> $L; tail -40 $L; exit 1; }
  done
done
grep "lookback fwd\|!!\|FAIL\|==\|error word" $L | tail -80
