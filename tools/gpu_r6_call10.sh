#!/bin/bash
# round 6, call 10: the LM head's weight gradient on the library's TN kernel - step-level A/B on one box + the LCE tests under it
set -e -o pipefail
mkdir -p gpurun_out
APERTIS_LCE_OWN_LOGITS=1 timeout -k 10 300 python -m pytest tests/test_model_gpu.py -q -x -k "linear_cross_entropy or fused_lm_head" > gpurun_out/r6c10_tests.log 2>&1
tail -2 gpurun_out/r6c10_tests.log
for i in 1 2; do
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6c10_base_$i.json 2> gpurun_out/r6c10_base_$i.err
  APERTIS_LCE_OWN_LOGITS=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6c10_own_$i.json 2> gpurun_out/r6c10_own_$i.err
done
python - <<'PY'
import json
for t in ("base_1","own_1","base_2","own_2"):
    d=json.loads(open(f"gpurun_out/r6c10_{t}.json").read().strip().splitlines()[-1])
    print(t, round(d["ms_per_step"],2), round(d["value"]), d.get("final_loss"))
PY
