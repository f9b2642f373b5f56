"""Launch each hot kernel a few times at BASELINE (1.5B, B=8/16, seq 4096) shapes - target of
rocprofv3 --kernel-trace / --pmc runs (see profiles/README.md)."""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from apertis_llm_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
P, S = _lib.ptr, _lib.stream_ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
what = sys.argv[2] if len(sys.argv) > 2 else "all"

if what == "benchmix":
    # exactly the per-layer hot-kernel launches of bench.py's default workload (1.5b-moe, per-GPU batch 32)
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    # (argv[4]: the BASELINE configuration whose layer shapes are replayed - round 6: configs 3 and 5 get traffic files too)
    L, h, N, H, I, E = {"1.5b-moe": (4096, 11, 16, 704, 2816, 8), "350m-moe": (4096, 4, 16, 256, 1024, 8),
                        "1.5b-moe-mm": (2048 + 197, 11, 16, 704, 2816, 8)}[sys.argv[4] if len(sys.argv) > 4 else "1.5b-moe"]
    Dn, R = h * N, math.ceil(H / 16)
    Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64                 # the model's padded x_param_proj layout [Bt | 0 | C | 0 | dt | 0]
    rows = E * int((B * L / E) * 1.25)      # capacity-limited rows of the train step
    p = torch.randn(B, L, 2 * Wb + Wr, device=dev).bfloat16().requires_grad_(True)
    xz = torch.randn(B, L, 2 * Dn, device=dev).bfloat16().requires_grad_(True)
    xc = torch.randn(B, L, Dn, device=dev).bfloat16().requires_grad_(True)
    dl = (torch.randn(B, L, h, device=dev) - 4).requires_grad_(True)
    A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99)).requires_grad_(True)
    Dsk = torch.ones(Dn, device=dev, requires_grad=True)
    dy = torch.randn(B, L, Dn, device=dev).bfloat16()
    xg = torch.randn(rows, H, device=dev).bfloat16().requires_grad_(True)
    w1 = (torch.randn(E, I, H, device=dev) * 0.02).requires_grad_(True)
    b1 = torch.zeros(E, I, device=dev, requires_grad=True)
    w2 = (torch.randn(E, H, I, device=dev) * 0.02).requires_grad_(True)
    b2 = torch.zeros(E, H, device=dev, requires_grad=True)
    offs = torch.tensor(np.linspace(0, rows, E + 1).astype(np.int32), device=dev)
    dyr = torch.randn(rows, H, device=dev).bfloat16()
    for _ in range(reps):
        Btp, Cp, _dt = ops.split_cols(p, (Wb, Wb, Wr))
        _xp, z = ops.split_cols(xz, (Dn, Dn))
        y = ops.scan_gate(dl, A, Btp, Cp, xc, z, Dsk, delta_softplus=True)
        y.backward(dy)
        yr = ops.expert_mlp(xg, w1, b1, w2, b2, offs, rows, act="gelu", drop_p=0.1, seed=5, compute_dtype=torch.bfloat16)
        yr.backward(dyr)
if what in ("all", "scan"):
    B, L, h, N = 16, 4096, 11, 16
    Dn, R = h * N, 44
    p = torch.randn(B, L, R + 2 * Dn, device=dev).bfloat16().requires_grad_(True)
    dl = (torch.randn(B, L, h, device=dev) - 4).requires_grad_(True)
    A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99)).requires_grad_(True)
    dy = torch.randn(B, L, Dn, device=dev)
    for _ in range(reps):
        y = ops.selective_scan(dl, A, p[..., R:R + Dn], p[..., R + Dn:], delta_softplus=True)
        y.backward(dy)
if what in ("all", "gemm"):
    rows, E = 40960, 8
    offs = torch.tensor(np.linspace(0, rows, E + 1).astype(np.int32), device=dev)
    for (N_, K_) in [(2816, 704), (704, 2816)]:
        x = torch.randn(rows, K_, device=dev).bfloat16()
        W = torch.randn(E, N_, K_, device=dev) / K_ ** 0.5
        wc, wt = ops.cast_transpose(W, torch.bfloat16)
        out = torch.empty(rows, N_, device=dev, dtype=torch.bfloat16)
        pre = torch.empty_like(out)
        dw = torch.empty(E, N_, K_, device=dev)
        ws = torch.empty(max(16, lib.apertis_grouped_gemm_tn_workspace_bytes(E, 1)), device=dev, dtype=torch.uint8)
        for _ in range(reps):
            lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), None, None, rows, N_, K_, wc.shape[-1], E, 0, 0.0, 0, 1, 1, S())
            lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), P(pre), None, rows, N_, K_, wc.shape[-1], E, 1, 0.1, 7, 1, 1, S())
            lib.apertis_grouped_gemm_tn(P(out), P(x), P(offs), P(dw), None, rows, N_, K_, E, P(ws), ws.numel(), 1, S())
torch.cuda.synchronize()
