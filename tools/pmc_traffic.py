"""Fold rocprofv3 --pmc passes (tools/run_pmc.sh) into per-kernel and per-entry-point numbers.
HBM traffic follows MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB and on gfx950
FETCH_SIZE counts 128-byte requests as 64 bytes, so reads are doubled:
    traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
Writes <out>/summary.txt and <out>/traffic.json (bytes per C-ABI call)."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
ENTRY = {  # kernel-name fragment -> C-ABI entry point
    "grouped_gemm_nt256p_k": "apertis_grouped_gemm_nt",
    "grouped_gemm_nt352p_k": "apertis_grouped_gemm_nt",
    "grouped_gemm_nt2x_k": "apertis_grouped_gemm_nt",
    "grouped_gemm_nt4r_k": "apertis_grouped_gemm_nt",
    "grouped_gemm_tn3_k": "apertis_grouped_gemm_tn", "tn3_fold_k": "apertis_grouped_gemm_tn",
    "grouped_gemm_tn5_k": "apertis_grouped_gemm_tn", "tn5_fold_k": "apertis_grouped_gemm_tn",
    "grouped_gemm_tn2_k": "apertis_grouped_gemm_tn",
    "scan_fwd_state": "apertis_selective_scan_fwd", "scan_fwd_replay": "apertis_selective_scan_fwd",
    "scan_bwd_state": "apertis_selective_scan_bwd", "scan_bwd_replay": "apertis_selective_scan_bwd",
    "scan_gate_fwd_k": "apertis_scan_gate_fwd", "scan_gate_bwd_k": "apertis_scan_gate_bwd",
    "scan_lean_state_k": "apertis_scan_gate_fwd", "scan_lean_prefix_k<false>": "apertis_scan_gate_fwd",
    "scan_lean_prefix_k<(bool)0>": "apertis_scan_gate_fwd", "scan_lean_fwd_k": "apertis_scan_gate_fwd",
    "scan_lean_bstate_k": "apertis_scan_gate_bwd", "scan_lean_prefix_k<true>": "apertis_scan_gate_bwd",
    "scan_lean_prefix_k<(bool)1>": "apertis_scan_gate_bwd", "scan_lean_bwd_k": "apertis_scan_gate_bwd",
    "scan_lb_fwd_k": "apertis_scan_gate_fwd", "scan_lb_bwd_k": "apertis_scan_gate_bwd",
    "colsum_kernel": "apertis_scan_gate_bwd",
}
CALLS_PER_REP = {"apertis_grouped_gemm_nt": 4, "apertis_grouped_gemm_tn": 1, "apertis_selective_scan_fwd": 1,
                 "apertis_selective_scan_bwd": 1, "apertis_scan_gate_fwd": 1, "apertis_scan_gate_bwd": 1}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/sq1/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = []
entry = collections.defaultdict(lambda: {"fetch_kib": 0.0, "write_kib": 0.0, "launch_groups": 0})
for k in sorted(agg):
    tag = next((e for frag, e in ENTRY.items() if frag in k), None)
    if tag is None and not any(s in k for s in ("scan", "gemm", "colsum", "ln_", "layernorm", "gather", "combine")):
        continue
    d = sorted(dur.get(k, [0]))
    c = agg[k]
    lines.append(f"{k[:100]}\n   launches {len(d)} median_us {d[len(d) // 2]:.1f}  " + "  ".join(
        f"{n}={sum(v) / len(v):.4g}" for n, v in sorted(c.items())))
    if tag and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        entry[tag]["fetch_kib"] += sum(c["FETCH_SIZE"])
        entry[tag]["write_kib"] += sum(c["WRITE_SIZE"])
        entry[tag]["launch_groups"] = max(entry[tag]["launch_groups"], len(c["FETCH_SIZE"]))
res = {}
for tag, v in entry.items():
    if "colsum" in tag:
        continue
    # reps = launches of the per-call-once kernel / calls per rep
    n_calls = None
    for frag, e in ENTRY.items():
        if e == tag and frag not in ("colsum_kernel", "tn3_fold_k", "tn5_fold_k"):
            ks = [k for k in agg if frag in k and "FETCH_SIZE" in agg[k]]
            if ks:
                n = sum(len(agg[k]["FETCH_SIZE"]) for k in ks)
                if tag == "apertis_grouped_gemm_nt":   # each call launches exactly ONE of the NT kernels
                    n_calls = n + (n_calls or 0)
                else:
                    n_calls = n if n_calls is None else min(n_calls, n)
    if not n_calls:
        continue
    res[tag] = {"calls": n_calls, "fetch_bytes_per_call": 2 * v["fetch_kib"] * 1024 / n_calls,
                "write_bytes_per_call": v["write_kib"] * 1024 / n_calls,
                "traffic_bytes_per_call": (2 * v["fetch_kib"] + v["write_kib"]) * 1024 / n_calls,
                "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                        "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)"}
# what the numbers were taken on: bench.py reports them only while the kernel sources are the ones measured
import hashlib, os, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = {}
for f in ("grouped_gemm.hip", "scan_gate.hip", "scan_lookback.hip", "scan_lean.h"):
    with open(os.path.join(root, "apertis_llm_amd", "csrc", f), "rb") as fh:
        src[f] = hashlib.sha256(fh.read()).hexdigest()[:16]
res["_source"] = {"kernel_source_sha16": src}
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open(out + "/traffic.json", "w"), indent=1)
print("\n".join(lines[:60]))
print(json.dumps(res, indent=1))
