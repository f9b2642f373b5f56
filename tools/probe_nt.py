"""NT two-per-CU kernel phase probe (needs the PROBE build of grouped_gemm.hip)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apertis_llm_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load(); P, S = _lib.ptr, _lib.stream_ptr
raw = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libapertis_hip.so"))
buf = (ctypes.c_ulonglong * 16)()
for (rows, N, K, E) in [(163840, 2816, 704, 8)]:
    x = torch.randn(rows, K, device=dev).bfloat16()
    W = torch.randn(E, N, K, device=dev) / K ** 0.5
    offs = torch.tensor(np.linspace(0, rows, E + 1).astype(np.int32), device=dev)
    wc, _ = ops.cast_transpose(W, torch.bfloat16)
    out = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)
    pre = torch.randn(rows, N, device=dev).bfloat16()
    for mode, args in [("plain", (None, None, 0, 0.0)), ("gelu+drop+pre", (pre, None, 1, 0.1)), ("actbwd", (None, pre, 1, 0.1))]:
        for _ in range(2):
            torch.cuda.synchronize(); raw.apertis_dbg_prof(buf, 1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), P(args[0]), P(args[1]), rows, N, K, wc.shape[-1], E, args[2], args[3], 7, 1, 1, S())
            e.record(); torch.cuda.synchronize(); raw.apertis_dbg_prof(buf, 0)
        n = max(1, buf[3])
        print(f"{mode:14s} K={K} N={N}: {s.elapsed_time(e)*1e3:.0f} us; per WG-tile us: prologue {buf[0]/n/100:.2f}  kloop {buf[1]/n/100:.2f}  epilogue {buf[2]/n/100:.2f}  tiles {n}")
        print(f"      bias {buf[12]/n/100:.2f} | raw pass: conv+ldsw {buf[4]/n/100:.2f} barrier {buf[5]/n/100:.2f} ldsr+store {buf[6]/n/100:.2f} barrier {buf[7]/n/100:.2f} | act pass: {buf[8]/n/100:.2f} {buf[9]/n/100:.2f} {buf[10]/n/100:.2f} {buf[11]/n/100:.2f}")
