"""NT main-loop probe: plain NT at a few K with the kernel's debug modes (APERTIS_GEMM_DBG high bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from apertis_llm_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load(); P, S = _lib.ptr, _lib.stream_ptr
def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
for (rows, N, K, E) in [(163840, 2816, 704, 8), (163840, 704, 2816, 8), (65536, 4096, 4096, 8)]:
    x = torch.randn(rows, K, device=dev).bfloat16()
    W = torch.randn(E, N, K, device=dev) / K ** 0.5
    offs = torch.tensor(np.linspace(0, rows, E + 1).astype(np.int32), device=dev)
    wc, _ = ops.cast_transpose(W, torch.bfloat16)
    out = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), None, None, rows, N, K, wc.shape[-1], E, 0, 0.0, 0, 1, 1, S()))
    print(f"SOLO={os.environ.get('APERTIS_GEMM_SOLO')} rows={rows} N={N} K={K}: {t*1e3:.1f} us  {2.0*rows*N*K/t/1e9:.0f} TF-equivalent")
    import ctypes
    raw = ctypes.CDLL(os.path.join(os.path.dirname(_lib.__file__), "libapertis_hip.so"))
    buf = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize(); raw.apertis_dbg_prof(buf, 1)
    for mode, kw in [("plain", (None, 0, 0.0)), ("gelu+drop+pre", ("pre", 1, 0.1))]:
        pre = torch.empty_like(out) if kw[0] else None
        lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), P(pre), None, rows, N, K, wc.shape[-1], E, kw[1], kw[2], 7, 1, 1, S())
        torch.cuda.synchronize(); raw.apertis_dbg_prof(buf, 1)
        for o, nm in ((0, "wave0"), (4, "wave4")):
            n = max(1, buf[o + 3]); tiles = (rows // 256) * ((N + 255) // 256) / n
            print(f"   {mode:14s} {nm}: per tile (100MHz ticks->us) find+stage0 {buf[o]/n/tiles/100:.2f}  epilogue {buf[o+1]/n/tiles/100:.2f}  kloop {buf[o+2]/n/tiles/100:.2f}  WGs {n}")
