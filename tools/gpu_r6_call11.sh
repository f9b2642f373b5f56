#!/bin/bash
# round 6, call 11: chunk rows of the fused LM head + CE with the weight gradient on the library's TN kernel (same box)
set -e -o pipefail
mkdir -p gpurun_out
for rows in 16384 8192 32768 4096 16384; do
  APERTIS_LCE_CHUNK_ROWS=$rows timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6c11_$rows.json 2> gpurun_out/r6c11_$rows.err
  python - $rows <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r6c11_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("chunk rows", sys.argv[1], round(d["ms_per_step"],2), "ms", round(d["value"]), "tok/s peak GiB", d["config"].get("peak_mem_gib"))
PY
done
