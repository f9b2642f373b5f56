"""The three GEMMs of one chunk of the fused LM head + cross entropy (ops/loss.py: rows = 16384, H = 704, V = 32000, bf16):
stock torch.matmul (hipBLASLt) against the library's own NT / TN kernels on the same operands, HIP-event timed, cache
flushed between launches.    python tools/prof_lm_head.py [rows] [H] [V]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops, _lib
from apertis_llm_amd._lib import ptr, stream_ptr, check

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 704
V = int(sys.argv[3]) if len(sys.argv) > 3 else 32000
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
x = torch.randn(rows, H, device=dev).bfloat16()
w = (torch.randn(V, H, device=dev) * 0.02).bfloat16()
wt = w.t().contiguous()
dl = (torch.randn(rows, V, device=dev) * 0.01).bfloat16()
offs = ops.dense_offsets(rows, dev)
code = _lib.dtype_code(x)
flush = torch.empty(1 << 27, device=dev, dtype=torch.float32)


def timed(fn, n=12):
    ts = []
    for i in range(n + 3):
        flush.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def nt(xm, wm, out, N, K):
    check(lib.apertis_grouped_gemm_nt(ptr(xm), ptr(wm), None, ptr(offs), ptr(out), None, None, rows, N, K, wm.shape[-1], 1,
                                      _lib.ACT_NONE, 0.0, 0, code, code, stream_ptr()), "nt")


flop = 2.0 * rows * H * V
print(f"rows={rows} H={H} V={V}: {flop / 1e12:.3f} TFLOP per GEMM")
# 1. logits = x @ w.T   [rows, V]
lg_t = torch.empty(rows, V, device=dev, dtype=torch.bfloat16)
lg_m = torch.empty(rows, V, device=dev, dtype=torch.bfloat16)
t_med, t_min = timed(lambda: torch.matmul(x, w.t(), out=lg_t))
print(f"logits   torch.matmul          {t_med:8.1f} us (min {t_min:8.1f})  {flop / t_med / 1e6:7.1f} TFLOP/s")
t_med, t_min = timed(lambda: nt(x, w, lg_m, V, H))
print(f"logits   apertis NT N={V} K={H}  {t_med:8.1f} us (min {t_min:8.1f})  {flop / t_med / 1e6:7.1f} TFLOP/s")
print("         max |diff| vs torch", (lg_t.float() - lg_m.float()).abs().max().item(), " max |logit|", lg_t.float().abs().max().item())
# 2. dx = dl @ w   [rows, H]
dx_t = torch.empty(rows, H, device=dev, dtype=torch.bfloat16)
dx_m = torch.empty(rows, H, device=dev, dtype=torch.bfloat16)
t_med, t_min = timed(lambda: torch.matmul(dl, w, out=dx_t))
print(f"dhidden  torch.matmul          {t_med:8.1f} us (min {t_min:8.1f})  {flop / t_med / 1e6:7.1f} TFLOP/s")
t_med, t_min = timed(lambda: nt(dl, wt, dx_m, H, V))
print(f"dhidden  apertis NT N={H} K={V}  {t_med:8.1f} us (min {t_min:8.1f})  {flop / t_med / 1e6:7.1f} TFLOP/s")
print("         max |diff| vs torch", (dx_t.float() - dx_m.float()).abs().max().item(), " max |dx|", dx_t.float().abs().max().item())
# 3. dW = dl.T @ x   [V, H]
t_med, t_min = timed(lambda: torch.matmul(dl.t(), x))
dw_t = torch.matmul(dl.t(), x)
print(f"dweight  torch.matmul          {t_med:8.1f} us (min {t_min:8.1f})  {flop / t_med / 1e6:7.1f} TFLOP/s")
dw_m = torch.empty(1, V, H, device=dev, dtype=torch.float32)
nbytes = lib.apertis_grouped_gemm_tn_workspace_bytes(1, 1)
ws = torch.empty(max(nbytes, 16), device=dev, dtype=torch.uint8)


def tn():
    check(lib.apertis_grouped_gemm_tn(ptr(dl), ptr(x), ptr(offs), ptr(dw_m), None, rows, V, H, 1, ptr(ws), nbytes, code,
                                      stream_ptr()), "tn")


print("         dense variant", lib.apertis_grouped_gemm_tn_dense_variant(V, H))
t_med, t_min = timed(tn)
print(f"dweight  apertis TN M={V} N={H} {t_med:8.1f} us (min {t_min:8.1f})  {flop / t_med / 1e6:7.1f} TFLOP/s  (fp32 out)")
print("         max |diff| vs torch", (dw_t.float() - dw_m[0]).abs().max().item(), " max |dw|", dw_t.float().abs().max().item())
