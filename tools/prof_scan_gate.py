"""Target of `rocprofv3 --kernel-trace --stats`: the fused scan + gate op at the bench's per-layer shape, both forms.
    python tools/prof_scan_gate.py [reps] [B] [single_pass 0|1|both]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
which = sys.argv[3] if len(sys.argv) > 3 else "both"
L, h, N, dt = 4096, 11, 16, torch.bfloat16
Dn, R = h * N, 44
Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
p = torch.randn(B, L, 2 * Wb + Wr, device=dev).to(dt).requires_grad_(True)
PAD = bool(os.environ.get("APERTIS_SCAN_PAD_EXPERIMENT"))
Wz = Wb if PAD else Dn
xz = torch.randn(B, L, 2 * Wz, device=dev).to(dt).requires_grad_(True)
xc_buf = torch.randn(B, L, Wz, device=dev).to(dt).requires_grad_(True)
dl = (torch.randn(B, L, h, device=dev) - 4).requires_grad_(True)
A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99)).requires_grad_(True)
D = torch.ones(Dn, device=dev, requires_grad=True)
dout = torch.randn(B, L, Wz, device=dev).to(dt)[..., :Dn]
for sp in ([False, True] if which == "both" else [which == "1"]):
    ops.SCAN_SINGLE_PASS = sp
    for _ in range(reps):
        Btp, Cp, _dt = ops.split_cols(p, (Wb, Wb, Wr))
        _xp, _p1, z, _p2 = ops.split_cols(xz, (Dn, Wz - Dn, Dn, Wz - Dn))
        xc = xc_buf[..., :Dn]
        out = ops.scan_gate(dl, A, Btp, Cp, xc, z, D, delta_softplus=True)
        out.backward(dout)
        torch.cuda.synchronize()
print("err", ops.scan_gate_error())
