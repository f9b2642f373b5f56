"""The scan + gate op's three forms against each other: staged single-pass kernels, lean (three launches), look-back (one
launch) - outputs, gradients, run-to-run bits, time (cold caches: 1 GiB is read between launches).
    python tools/scan_lean_check.py [B L h N]        (SCAN_CHECK_TIMES_ONLY=1: exit 0 whatever the comparison says - probe
    builds with arithmetic or waits compiled out)"""
import math, os, sys, torch
sys.path.insert(0, ".")
from apertis_llm_amd import ops
dev = torch.device("cuda:0")
shapes = [(44, 4096, 11, 16), (16, 4096, 4, 16), (32, 2048, 14, 16), (3, 257, 11, 16), (2, 2245, 11, 16), (5, 100, 4, 16),
          (1, 64, 16, 16), (7, 1000, 2, 16), (3, 130, 1, 16), (9, 333, 8, 16), (2, 4096 * 4, 6, 16)]
if len(sys.argv) > 4:
    shapes = [tuple(int(a) for a in sys.argv[1:5])]
flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)
names = ("dl", "A", "p", "xz", "xc", "D")
MODES = {"staged": (False, False), "lean": (True, False), "lookback": (False, "all")}


def set_mode(mode):
    lean, lb = MODES[mode]
    ops.SCAN_LEAN = ops.SCAN_LEAN_BWD = lean
    ops.SCAN_LOOKBACK = lb


bad = 0
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
    big = B * L >= 65536
    for mode in ("staged", "lean", "lookback") * (2 if big else 1):
        set_mode(mode)

        def fwd():
            Btp, Cp, _ = ops.split_cols(p, (Wb, Wb, Wr))
            _, z = ops.split_cols(xz, (Dn, Dn))
            return ops.scan_gate(dl, A, Btp, Cp, xc, z, D, delta_softplus=True)
        tf, tb, prev = [], [], None
        for it in range(4 if big else 2):
            out = fwd()
            flush.sum()
            with torch.no_grad():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fwd(); e1.record(); e1.synchronize()
            tf.append(e0.elapsed_time(e1) * 1e3)
            flush.sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g = torch.autograd.grad(out, (dl, A, p, xz, xc, D), dout)
            e1.record(); e1.synchronize()
            tb.append(e0.elapsed_time(e1) * 1e3)
            cur = (out.detach(), [t.detach() for t in g])
            if prev is not None:      # run-to-run identical bits
                same = torch.equal(prev[0], cur[0]) and all(torch.equal(a, b) for a, b in zip(prev[1], cur[1]))
                if not same:
                    bad += 1
                    print(f"   !! {mode}: two runs differ")
            prev = cur
        res[mode] = (out.detach().float(), [t.detach().float() for t in g])
        print(f"B={B} L={L} Dn={Dn} {mode:8s} fwd best {min(tf[1:]):7.1f} us = {fb/min(tf[1:])/8e6*100:4.1f} %   "
              f"bwd (op, host-timed) best {min(tb[1:]):7.1f} us = {bb/min(tb[1:])/8e6*100:4.1f} % of 8 TB/s")
    a = res["staged"]
    for mode in ("lean", "lookback"):
        b = res[mode]
        d = (a[0] - b[0]).abs()
        tol = 2.0 ** -7 * float(a[0].abs().max())
        flag = "" if float(d.max()) <= tol and torch.isfinite(b[0]).all() else "  !! OUT"
        bad += bool(flag)
        print(f"   {mode:8s} out: max abs diff {float(d.max()):.3e} (absmax {float(a[0].abs().max()):.3e}), differing {float((d > 0).float().mean())*100:.2f} %{flag}")
        for n, ga, gb in zip(names, a[1], b[1]):
            dd = (ga - gb).abs()
            rel = float(dd.pow(2).mean().sqrt() / (ga.pow(2).mean().sqrt() + 1e-30))
            pad_ok = bool((gb[..., Dn:Wb] == 0).all() and (gb[..., Wb + Dn:2 * Wb] == 0).all()) if n == "p" else True
            flag = "" if rel < 4e-3 and pad_ok and torch.isfinite(gb).all() else "  !! GRAD"
            bad += bool(flag)
            print(f"   {mode:8s} d{n}: max abs diff {float(dd.max()):.3e}  absmax {float(ga.abs().max()):.3e}  rel rms {rel:.2e}"
                  f"  pad cols zero: {pad_ok if n == 'p' else '-'}{flag}")
    print("   scan error word", ops.scan_gate_error())
print("FAILURES", bad)
sys.exit(1 if bad and os.environ.get("SCAN_CHECK_TIMES_ONLY") != "1" else 0)
