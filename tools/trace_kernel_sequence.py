"""The launch sequence of one training step from a rocprofv3 --kernel-trace CSV: every kernel in launch order with its grid, runs of
the same kernel folded - to see WHERE in the step the stock-torch fills / copies sit.
    python tools/trace_kernel_sequence.py <kernel_trace.csv> [first_line last_line]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"([A-Za-z0-9_]+_k)\b", n)
    if m:
        return m.group(1)
    m = re.search(r"(FillFunctor<[^>]*>|CUDAFunctor_add<[^>]*>|copyBuffer|fillBufferAligned|gather_kernel|Cijk\w{0,12}|reduce_kernel|direct_copy)", n)
    return m.group(1) if m else n[:48]
seq = [(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1)) for r in rows]
# the last step: from the last weight_prep_k on
starts = [i for i, (n, _) in enumerate(seq) if n == "weight_prep_k"]
seq = seq[starts[-1]:] if starts else seq
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, 400)
out, prev, cnt = [], None, 0
for i, (n, g) in enumerate(seq[lo:hi]):
    print(f"{lo + i:5d}  {n:40s} grid {g}")
