"""Lean scan forward (APERTIS_SCAN_LEAN) against the single-pass kernel: outputs, saved chunk states, time (cold caches: 1 GiB is
read between launches).  python tools/scan_lean_check.py [B L h N]"""
import math, sys, torch
sys.path.insert(0, ".")
from apertis_llm_amd import ops
dev = torch.device("cuda:0")
shapes = [(44, 4096, 11, 16), (32, 2048, 14, 16), (16, 4096, 4, 16), (3, 257, 11, 16)]
if len(sys.argv) > 4:
    shapes = [tuple(int(a) for a in sys.argv[1:5])]
flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)
for (B, L, h, N) in shapes:
    torch.manual_seed(0)
    Dn, R = h * N, math.ceil(h * 64 / 16)
    Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
    p = torch.randn(B, L, 2 * Wb + Wr, device=dev).bfloat16()
    xz = torch.randn(B, L, 2 * Dn, device=dev).bfloat16()
    xc = torch.randn(B, L, Dn, device=dev).bfloat16()
    dl = torch.randn(B, L, h, device=dev) - 4
    A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99))
    D = torch.ones(Dn, device=dev)
    Btp, Cp, _ = ops.split_cols(p, (Wb, Wb, Wr))
    _, z = ops.split_cols(xz, (Dn, Dn))
    fb = B * L * (5 * Dn * 2 + 4 * h)
    res = {}
    for lean in (False, True, False, True):
        ops.SCAN_LEAN = lean
        with torch.no_grad():
            out, hl = ops.scan_gate(dl, A, Btp, Cp, xc, z, D, delta_softplus=True, return_last=True)
            ts = []
            for _ in range(6):
                flush.sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.scan_gate(dl, A, Btp, Cp, xc, z, D, delta_softplus=True); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        res[lean] = (out.float(), hl)
        t = min(ts[1:])
        print(f"B={B} L={L} Dn={Dn} {'lean  ' if lean else 'single'} fwd best {t:7.1f} us  {fb/t/1e6:6.0f} GB/s = {fb/t/8e6*100:4.1f} % of 8 TB/s   (all {[round(x) for x in ts]})")
    ops.SCAN_LEAN = False
    a, b = res[False], res[True]
    d = (a[0] - b[0]).abs()
    print(f"   out: max abs diff {float(d.max()):.3e} (ref absmax {float(a[0].abs().max()):.3e}), mean {float(d.mean()):.3e}, differing {float((d > 0).float().mean())*100:.2f} %;"
          f" h_last max rel {float(((a[1]-b[1]).abs() / (a[1].abs() + 1e-6)).max()):.2e}; scan error word {ops.scan_gate_error()}")
