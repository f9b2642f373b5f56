#!/bin/bash
# kernel trace of bench.py --decode (eager token steps, so that every launch shows): calls per kernel name
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -rf gpurun_out/dec_trace
export APERTIS_DECODE_GRAPH=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dec_trace -- python3 bench.py --decode > gpurun_out/dec_trace.json 2> gpurun_out/dec_trace.err || { tail -5 gpurun_out/dec_trace.err; exit 1; }
f=$(ls gpurun_out/dec_trace/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r4_decode_kernel_stats.csv; rm -rf gpurun_out/dec_trace
python3 - <<'PY'
import csv, re
rows = list(csv.DictReader(open("gpurun_out/r4_decode_kernel_stats.csv")))
tot = sum(int(r["Calls"]) for r in rows)
print("launches in the whole run (2 prefills + 2 x 128 token steps):", tot, "->", round(tot / 256 / 44, 1), "per layer and token step")
for r in sorted(rows, key=lambda r: -int(r["Calls"]))[:45]:
    n = r["Name"]; m = re.search(r"([A-Za-z_0-9]+_k\b|Cijk[A-Za-z0-9_]{0,24}|[A-Za-z_]*kernel[A-Za-z_]*)", n)
    print(f"  {(m.group(1) if m else n)[:52]:52s} calls {int(r['Calls']):7d} = {int(r['Calls']) / 256 / 44:5.2f} /layer/step  avg {float(r['AverageNs']) / 1e3:7.1f} us")
PY
