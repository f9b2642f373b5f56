"""Depthwise conv + SiLU forward / backward at the bench's per-layer shape (x = the first Dn columns of the in_proj output),
cold caches between launches - target of `rocprofv3 --kernel-trace --stats`.  Usage: python tools/prof_conv.py [reps] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
L, Dn, k = 4096, 176, 4
xz = torch.randn(B, L, 2 * Dn, device=dev).bfloat16().requires_grad_(True)
w = (torch.randn(Dn, 1, k, device=dev) * 0.3).requires_grad_(True)
b = (torch.randn(Dn, device=dev) * 0.1).requires_grad_(True)
dy = torch.randn(B, L, Dn, device=dev).bfloat16()
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
for _ in range(reps):
    xp, z = ops.split_cols(xz, (Dn, Dn))
    flush.sum()
    y = ops.dwconv_silu(xp, w, b)
    flush.sum()
    y.backward(dy)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
