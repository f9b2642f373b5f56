#!/bin/bash
# tools/gpu_lib_ab.sh <tag> <pytest -k expr | -> [bench args]: the in-tree library against apertis_llm_amd/libapertis_hip_base.so
# (built from another revision of one source: APERTIS_HIP_LIB) - the GPU tests selected by -k on the in-tree library first, then
# bench.py base / new / base / new on the same box
set -e -o pipefail
TAG=$1; K="$2"; shift 2
mkdir -p gpurun_out
if [ "$K" != "-" ]; then
  timeout -k 10 900 python -m pytest tests/test_moe_kernels_gpu.py -m gpu -x -q -k "$K" > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; exit 1; }
  tail -2 gpurun_out/${TAG}_tests.log
fi
for i in 1 2; do
  APERTIS_HIP_LIB=$PWD/apertis_llm_amd/libapertis_hip_base.so timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/${TAG}_base_$i.json 2> gpurun_out/${TAG}_base_$i.err
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/${TAG}_new_$i.json 2> gpurun_out/${TAG}_new_$i.err
done
python - $TAG <<'PY'
import json,sys
for t in ("base_1","new_1","base_2","new_2"):
    d=json.loads(open(f"gpurun_out/{sys.argv[1]}_{t}.json").read().strip().splitlines()[-1])
    ra=d["roofline_all"]
    print(t, round(d["ms_per_step"],2), "ms", round(d["value"]), "tok/s  NT", round(d["roofline"]["frac"],4), "avg us", round(d["roofline"]["avg_ms"]*1e3,1), " loss", d["config"].get("final_loss"))
PY
