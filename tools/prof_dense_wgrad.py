"""Dense projections' weight gradients (deterministic split-K on the 128x128 TN kernel) at the bench's per-layer shapes for a few
row-group depths - target of `rocprofv3 --kernel-trace --stats`.  Usage: python tools/prof_dense_wgrad.py [B] [rows ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 40
depths = [int(a) for a in sys.argv[2:]] or [1024, 2048, 4096, 8192]
dev = torch.device("cuda:0")
T = B * 4096
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
for depth in depths:
    ops._splitk_depth = lambda N, K, _d=depth: _d      # override the shape rule: one fixed depth per sweep
    for (N, K) in [(352, 704), (448, 176), (704, 176)]:
        x = torch.randn(T, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.05).requires_grad_(True)
        dy = torch.randn(T, N, device=dev).bfloat16()
        for _ in range(4):
            flush.sum()
            torch.cuda.nvtx.range_push(f"depth{depth}_N{N}_K{K}") if False else None
            y = ops.linear_mfma(x, w, None, compute_dtype=torch.bfloat16)
            y.backward(dy)
            w.grad = None
        torch.cuda.synchronize()
        # event timing of the backward's wgrad part is not separable here; use the kernel trace (grid sizes identify the depth)
print("ok")
