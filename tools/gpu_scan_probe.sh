#!/bin/bash
# tools/gpu_scan_probe.sh <log> <probe lib suffixes...>: phase stamps of the look-back forward (-DLB_PROBE builds) at the bench shape
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; shift
for v in "$@"; do
  export APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_$v.so
  echo "== $v" >> $L
  for shp in "44 4096 11"; do timeout -k 10 200 python3 tools/scan_lb_probe.py $shp 2>&1 | grep -v amdgpu.ids >> $L || { tail -20 $L; exit 1; }; done
done
cat $L
