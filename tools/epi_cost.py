"""Cost split of the fused NT epilogue (persistent 256x256 kernel) at the up-projection shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apertis_llm_amd import _lib, ops
dev = torch.device("cuda:0")
lib = _lib.load(); P, S = _lib.ptr, _lib.stream_ptr
rows, N, K, E = 163840, 2816, 704, 8
x = torch.randn(rows, K, device=dev).bfloat16()
W = torch.randn(E, N, K, device=dev) / K ** 0.5
b = torch.randn(E, N, device=dev) * 0.1
offs = torch.tensor(np.linspace(0, rows, E + 1).astype(np.int32), device=dev)
wc, _ = ops.cast_transpose(W, torch.bfloat16)
out = torch.empty(rows, N, device=dev, dtype=torch.bfloat16); pre = torch.empty_like(out)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2] * 1e3
def call(bias, prep, act, p):
    return lambda: lib.apertis_grouped_gemm_nt(P(x), P(wc), P(bias), P(offs), P(out), P(prep), None, rows, N, K, wc.shape[-1], E, act, p, 7, 1, 1, S())
for name, args in [("plain", (None, None, 0, 0.0)), ("bias", (b, None, 0, 0.0)), ("bias+pre", (b, pre, 0, 0.0)),
                   ("bias+gelu", (b, None, 1, 0.0)), ("bias+gelu+pre", (b, pre, 1, 0.0)), ("bias+gelu+drop", (b, None, 1, 0.1)),
                   ("bias+gelu+drop+pre", (b, pre, 1, 0.1))]:
    print(f"{name:22s} {t(call(*args)):8.1f} us")
