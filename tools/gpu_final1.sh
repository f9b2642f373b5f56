#!/bin/bash
# PMC passes of the hot kernels at the bench's shapes, then the bench line that reads the traffic file
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
rm -rf gpurun_out/r3_pmc
bash tools/run_pmc.sh gpurun_out/r3_pmc benchmix 44 > gpurun_out/r3_pmc.log 2>&1 || { tail -5 gpurun_out/r3_pmc.log; exit 1; }
ls gpurun_out/r3_pmc | head; python3 -c "
import json; d=json.load(open('gpurun_out/r3_pmc/traffic.json')); print({k:(round(v['traffic_bytes_per_call']/1e9,3), v['calls']) for k,v in d.items()})"
# keep only the small artefacts
cp gpurun_out/r3_pmc/summary.txt gpurun_out/r3_pmc_benchmix_b44_summary.txt; cp gpurun_out/r3_pmc/traffic.json gpurun_out/r3_pmc_traffic_1.5b-moe_b44.json
rm -rf gpurun_out/r3_pmc/*/
