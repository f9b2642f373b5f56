"""Lean scan (APERTIS_SCAN_LEAN / _BWD) against the staged single-pass kernels: outputs, gradients, time (cold caches: 1 GiB is
read between launches).  python tools/scan_lean_check.py [B L h N]"""
import math, sys, torch
sys.path.insert(0, ".")
from apertis_llm_amd import ops
dev = torch.device("cuda:0")
shapes = [(44, 4096, 11, 16), (32, 2048, 14, 16), (3, 257, 11, 16), (2, 2245, 11, 16)]
if len(sys.argv) > 4:
    shapes = [tuple(int(a) for a in sys.argv[1:5])]
flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)
names = ("dl", "A", "p", "xz", "xc", "D")
for (B, L, h, N) in shapes:
    torch.manual_seed(0)
    Dn, R = h * N, math.ceil(h * 64 / 16)
    Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
    p = torch.randn(B, L, 2 * Wb + Wr, device=dev).bfloat16().requires_grad_(True)
    xz = torch.randn(B, L, 2 * Dn, device=dev).bfloat16().requires_grad_(True)
    xc = torch.randn(B, L, Dn, device=dev).bfloat16().requires_grad_(True)
    dl = (torch.randn(B, L, h, device=dev) - 4).requires_grad_(True)
    A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99)).requires_grad_(True)
    D = torch.ones(Dn, device=dev, requires_grad=True)
    dout = torch.randn(B, L, Dn, device=dev).bfloat16()
    fb, bb = B * L * (5 * Dn * 2 + 4 * h), B * L * (9 * Dn * 2 + 8 * h)
    res = {}
    for mode in ("staged", "lean", "staged", "lean"):
        ops.SCAN_LEAN = mode == "lean"
        ops.SCAN_LEAN_BWD = mode == "lean"
        def fwd():
            Btp, Cp, _ = ops.split_cols(p, (Wb, Wb, Wr))
            _, z = ops.split_cols(xz, (Dn, Dn))
            return ops.scan_gate(dl, A, Btp, Cp, xc, z, D, delta_softplus=True)
        tf, tb = [], []
        for _ in range(5):
            out = fwd()
            flush.sum()
            with torch.no_grad():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fwd(); e1.record(); e1.synchronize()
            tf.append(e0.elapsed_time(e1) * 1e3)
            timer = ops.KernelTimer(["apertis_scan_gate_bwd"]); ops.set_timer(timer) if hasattr(ops, "set_timer") else None
            flush.sum()
            g = torch.autograd.grad(out, (dl, A, p, xz, xc, D), dout)
            torch.cuda.synchronize()
        res[mode] = (out.detach().float(), [t.detach().float() for t in g])
        print(f"B={B} L={L} Dn={Dn} {mode:6s} fwd best {min(tf[1:]):7.1f} us = {fb/min(tf[1:])/8e6*100:4.1f} % of 8 TB/s")
    a, b = res["staged"], res["lean"]
    d = (a[0] - b[0]).abs()
    print(f"   out: max abs diff {float(d.max()):.3e} (absmax {float(a[0].abs().max()):.3e}), differing {float((d > 0).float().mean())*100:.2f} %")
    for n, ga, gb in zip(names, a[1], b[1]):
        dd = (ga - gb).abs()
        print(f"   d{n}: max abs diff {float(dd.max()):.3e}  absmax {float(ga.abs().max()):.3e}  rel rms {float(dd.pow(2).mean().sqrt() / (ga.pow(2).mean().sqrt() + 1e-30)):.2e}"
              f"  pad cols zero: {bool((gb[..., Dn:Wb] == 0).all()) if n == 'p' else '-'}")
    print("   scan error word", ops.scan_gate_error())
