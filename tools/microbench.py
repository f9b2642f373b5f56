"""Kernel micro-benchmarks at BASELINE shapes (runs on the GPU box).
    python tools/microbench.py scan|gemm|all
Reports algorithmic GB/s or TFLOP/s per C-ABI call (HIP events, median of reps)."""
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def scan(only_bench_shape=False):
    shapes = [(32, 4096, 11, 16, torch.bfloat16)] if only_bench_shape else None
    for (B, L, h, N, dt) in shapes or [(8, 4096, 11, 16, torch.bfloat16), (16, 4096, 11, 16, torch.bfloat16),
                             (32, 4096, 11, 16, torch.bfloat16), (32, 4096, 11, 16, torch.float32),
                             (8, 4096, 4, 16, torch.bfloat16), (32, 2048, 14, 16, torch.bfloat16),
                             (64, 4096, 11, 16, torch.bfloat16)]:
        Dn, R = h * N, math.ceil(h * 64 / 16)
        p = torch.randn(B, L, R + 2 * Dn, device=dev).to(dt)
        dl = torch.randn(B, L, h, device=dev) - 4
        A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99))
        dy = torch.randn(B, L, Dn, device=dev)
        pg, dlg, Ag = p.clone().requires_grad_(True), dl.clone().requires_grad_(True), A.clone().requires_grad_(True)
        f = lambda: ops.selective_scan(dl, A, p[..., R:R + Dn], p[..., R + Dn:], delta_softplus=True)
        t_f = timeit(f)
        y = ops.selective_scan(dlg, Ag, pg[..., R:R + Dn], pg[..., R + Dn:], delta_softplus=True)
        def b():
            torch.autograd.grad(y, (dlg, Ag, pg), dy, retain_graph=True)
        t_b = timeit(b)
        e = p.element_size()
        T = B * L
        fb = T * (Dn * (2 * e + 4) + 4 * h)
        bb = T * (Dn * (4 * e + 4) + 8 * h)
        print(f"scan B={B} L={L} Dn={Dn} {str(dt)[6:]}: fwd {t_f*1e3:7.1f} us {fb/t_f/1e6:7.0f} GB/s ({fb/t_f/8e9*100:4.1f}%)  "
              f"bwd(+autograd overhead) {t_b*1e3:7.1f} us {bb/t_b/1e6:7.0f} GB/s ({bb/t_b/8e9*100:4.1f}%)  [{fb/1e6:.0f}/{bb/1e6:.0f} MB]")


def scan_gate(only_bench_shape=False):
    """The fused scan + gate op (what the model runs), both forms, on the model's padded layout; GB/s on the fused
    variant's own byte count (5*Dn*e + 4h forward, 9*Dn*e + 8h backward)."""
    import os
    shapes = [(int(os.environ.get("MB_BATCH", "32")), 4096, 11, 16, torch.bfloat16)] if only_bench_shape else [
        (8, 4096, 11, 16, torch.bfloat16), (32, 4096, 11, 16, torch.bfloat16), (16, 4096, 4, 16, torch.bfloat16),
        (32, 2048, 14, 16, torch.bfloat16), (16, 2245, 11, 16, torch.bfloat16), (8, 4096, 11, 16, torch.float32)]
    for (B, L, h, N, dt) in shapes:
        Dn, R = h * N, math.ceil(h * 64 / 16)
        Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
        p = torch.randn(B, L, 2 * Wb + Wr, device=dev).to(dt).requires_grad_(True)
        xz = torch.randn(B, L, 2 * Dn, device=dev).to(dt).requires_grad_(True)
        xc = torch.randn(B, L, Dn, device=dev).to(dt).requires_grad_(True)
        dl = (torch.randn(B, L, h, device=dev) - 4).requires_grad_(True)
        A = torch.empty(h, N, device=dev).uniform_(math.log(.5), math.log(.99)).requires_grad_(True)
        D = torch.ones(Dn, device=dev, requires_grad=True)
        dout = torch.randn(B, L, Dn, device=dev).to(dt)
        e, T = p.element_size(), B * L
        fb, bb = T * (5 * Dn * e + 4 * h), T * (9 * Dn * e + 8 * h)
        for sp in (False, True):
            ops.SCAN_SINGLE_PASS = sp

            def mk():
                Btp, Cp, _ = ops.split_cols(p, (Wb, Wb, Wr))
                _, z = ops.split_cols(xz, (Dn, Dn))
                return ops.scan_gate(dl, A, Btp, Cp, xc, z, D, delta_softplus=True)
            with torch.no_grad():
                t_f = timeit(mk)
            out = mk()

            def b():
                torch.autograd.grad(out, (dl, A, p, xz, xc, D), dout, retain_graph=True)
            t_b = timeit(b)
            print(f"scan_gate {'1-launch' if sp else '2-launch'} B={B} L={L} Dn={Dn} {str(dt)[6:]}: fwd {t_f*1e3:7.1f} us "
                  f"{fb/t_f/1e6:7.0f} GB/s ({fb/t_f/8e9*100:4.1f}%)  bwd(+autograd/split overhead) {t_b*1e3:7.1f} us "
                  f"{bb/t_b/1e6:7.0f} GB/s ({bb/t_b/8e9*100:4.1f}%)  [{fb/1e6:.0f}/{bb/1e6:.0f} MB]  err={ops.scan_gate_error()}")
        ops.SCAN_SINGLE_PASS = True


def gemm():
    import numpy as np
    for (rows, N, K, E) in [(163840, 2816, 704, 8), (163840, 704, 2816, 8), (40960, 2816, 704, 8), (40960, 704, 2816, 8), (81920, 2816, 704, 8), (81920, 704, 2816, 8),
                            (10240, 1024, 256, 8), (65536, 4096, 4096, 8), (32768, 352, 704, 1), (32768, 704, 176, 1),
                            # the SSM block's dense projections and their data gradients at the bench shape
                            (131072, 352, 704, 1), (131072, 400, 176, 1), (131072, 704, 176, 1), (131072, 176, 704, 1),
                            (131072, 176, 400, 1), (131072, 704, 352, 1)]:
        x = torch.randn(rows, K, device=dev).bfloat16()
        W = torch.randn(E, N, K, device=dev) / K ** 0.5
        offs = torch.tensor(np.linspace(0, rows, E + 1).astype(np.int32), device=dev)
        wc, wt = ops.cast_transpose(W, torch.bfloat16)
        out = torch.empty(rows, N, device=dev, dtype=torch.bfloat16)
        pre = torch.empty_like(out)
        from apertis_llm_amd import _lib
        lib = _lib.load()
        P, S = _lib.ptr, _lib.stream_ptr
        def nt():
            lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), None, None, rows, N, K, wc.shape[-1], E, 0, 0.0, 0, 1, 1, S())
        def nt_epi():
            lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), P(pre), None, rows, N, K, wc.shape[-1], E, 1, 0.1, 7, 1, 1, S())
        def nt_actbwd():
            lib.apertis_grouped_gemm_nt(P(x), P(wc), None, P(offs), P(out), None, P(pre), rows, N, K, wc.shape[-1], E, 1, 0.1, 7, 1, 1, S())
        dw = torch.empty(E, N, K, device=dev)
        ws = torch.empty(max(16, _lib.load().apertis_grouped_gemm_tn_workspace_bytes(E, 1)), device=dev, dtype=torch.uint8)
        def tn():
            lib.apertis_grouped_gemm_tn(P(out), P(x), P(offs), P(dw), None, rows, N, K, E, P(ws), ws.numel(), 1, S())
        fl = 2.0 * rows * N * K
        t1, t2, t3 = timeit(nt), timeit(nt_epi), timeit(tn)
        pre.normal_()
        t4 = timeit(nt_actbwd)
        print(f"gemm rows={rows} N={N} K={K} E={E}: NT {t1*1e3:7.1f} us {fl/t1/1e9:6.0f} TF  NT+gelu+drop+pre {t2*1e3:7.1f} us "
              f"{fl/t2/1e9:6.0f} TF  NT*act'(pre)*mask {t4*1e3:7.1f} us {fl/t4/1e9:6.0f} TF  TN {t3*1e3:7.1f} us {fl/t3/1e9:6.0f} TF")


def rows():
    """Row kernels of the MoE / SSM glue at the bench shapes: time and algorithmic GB/s (fwd+bwd separately)."""
    T, H, E = 24 * 4096, 704, 8
    x = torch.randn(T, H, device=dev).bfloat16().requires_grad_(True)
    lw, lb = torch.ones(H, device=dev, requires_grad=True), torch.zeros(H, device=dev, requires_grad=True)
    W, b = (torch.randn(E, H, device=dev) * 0.02).requires_grad_(True), torch.zeros(E, device=dev, requires_grad=True)
    gl, gp = torch.randn(T, E, device=dev), torch.randn(T, H, device=dev).bfloat16()

    def fwd_bwd(make, grads, nbytes_f, nbytes_b, name):
        outs = make()
        tf = timeit(make)
        def bw():
            torch.autograd.backward(outs, grads, retain_graph=True)
        tb = timeit(bw)
        print(f"{name:28s} fwd {tf*1e3:7.1f} us {nbytes_f/tf/1e6:7.0f} GB/s   bwd {tb*1e3:7.1f} us {nbytes_b/tb/1e6:7.0f} GB/s")

    fwd_bwd(lambda: ops.router_ln_linear(x, lw, lb, 1e-5, W, b), [gl, gp], T * H * 2, T * H * 2 * 3, "router_ln_linear")
    xf = torch.randn(T, H, device=dev, requires_grad=True)
    gy = torch.randn(T, H, device=dev).bfloat16()
    gpf = torch.randn(T, H, device=dev)
    fwd_bwd(lambda: ops.layer_norm_pass(xf, lw, lb, 1e-5, out_dtype=torch.bfloat16), [gy, gpf], T * H * 6, T * H * 14,
            "layer_norm_pass f32->bf16")
    R, h, ld = 44, 11, 400
    p = torch.randn(24, 4096, ld, device=dev).bfloat16().requires_grad_(True)
    Wd, bd = (torch.randn(h, R, device=dev) * 0.1).requires_grad_(True), torch.zeros(h, device=dev, requires_grad=True)
    gd = torch.randn(24, 4096, h, device=dev)
    fwd_bwd(lambda: (ops.tiny_linear(p[..., :R], Wd, bd),), [gd], T * (R * 2 + h * 4), T * (R * 4 + h * 4), "tiny_linear 44->11")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("scan", "all"):
        scan()
    if what in ("scan_gate", "all"):
        scan_gate()
    if what == "scan_gate1":
        scan_gate(True)
    if what == "scan1":          # the bench shape only (for rocprofv3 --kernel-trace --stats)
        scan(True)
    if what in ("gemm", "all"):
        gemm()
    if what in ("rows", "all"):
        rows()
