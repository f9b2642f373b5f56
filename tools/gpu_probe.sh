#!/bin/bash
# tools/gpu_probe.sh <log> <probe args> <bins...>: run probe binaries (tools/probes/<bin>.bin) round-robin, twice; PROBE_* env passes through
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
L=gpurun_out/$1.log; : > $L; shift
ARGS="$1"; shift
for rep in 1 2; do for b in "$@"; do echo "== $b (pass $rep)" >> $L; timeout -k 10 200 tools/probes/$b.bin $ARGS >> $L 2>&1 || { echo "probe $b failed" >> $L; cat $L; exit 1; }; done; done
cat $L
