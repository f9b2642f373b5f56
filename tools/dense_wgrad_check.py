"""Dense (one-group) weight gradients: the wide-tile kernel with its own row split (apertis_grouped_gemm_tn, E = 1, workspace)
against the 128 x 128 kernel over pseudo-groups + apertis_colsum_f32 - values (against fp64 on the same bf16 operands) and
time (HIP events, cold caches).  python tools/dense_wgrad_check.py [B=44]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops, _lib
from apertis_llm_amd.ops import ptr, stream_ptr, check

B = int(sys.argv[1]) if len(sys.argv) > 1 else 44
dev = torch.device("cuda:0")
lib = _lib.load()
T = B * 4096
flush = torch.empty(1 << 28, device=dev, dtype=torch.float32)
code = _lib.BF16 if hasattr(_lib, "BF16") else 1
shapes = [(352, 704), (448, 176), (704, 176), (704, 2816), (768, 768), (448, 224), (896, 224), (448, 896), (896, 448)]


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        flush.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


for (N, K) in shapes:
    torch.manual_seed(0)
    x = torch.randn(T, K, device=dev).bfloat16()
    dy = torch.randn(T, N, device=dev).bfloat16()
    code = ops.dtype_code(x)
    offs = ops._dense_offsets(T, dev)
    depth = ops._splitk_depth(N, K)
    G = -(-T // depth)
    soffs = ops._splitk_offsets(T, G, depth, dev)
    part = torch.empty(G, N, K, device=dev, dtype=torch.float32)
    dw_a = torch.empty(1, N, K, device=dev, dtype=torch.float32)
    dw_b = torch.full((1, N, K), float("nan"), device=dev, dtype=torch.float32)

    def narrow():
        check(lib.apertis_grouped_gemm_tn(ptr(dy), ptr(x), ptr(soffs), ptr(part), None, T, N, K, G, None, 0, code, stream_ptr()), "tn")
        check(lib.apertis_colsum_f32(ptr(part), ptr(dw_a), G, N * K, stream_ptr()), "colsum")

    ws, wsb = ops._tn_workspace(1, 1, dev, T)
    var = lib.apertis_grouped_gemm_tn_dense_variant(N, K)

    def wide():
        check(lib.apertis_grouped_gemm_tn(ptr(dy), ptr(x), ptr(offs), ptr(dw_b), None, T, N, K, 1, ptr(ws), wsb, code, stream_ptr()), "tn wide")

    ta = timed(narrow)
    tb = timed(wide)
    # reference on a row sample large enough to be meaningful but cheap: the first 16384 rows in fp64, both paths rerun on them
    Ts = 16384
    ref = dy[:Ts].double().t() @ x[:Ts].double()
    offs_s = ops._dense_offsets(Ts, dev)
    Gs = -(-Ts // depth)
    soffs_s = ops._splitk_offsets(Ts, Gs, depth, dev)
    check(lib.apertis_grouped_gemm_tn(ptr(dy), ptr(x), ptr(soffs_s), ptr(part), None, Ts, N, K, Gs, None, 0, code, stream_ptr()), "tn")
    check(lib.apertis_colsum_f32(ptr(part), ptr(dw_a), Gs, N * K, stream_ptr()), "colsum")
    dw_b.fill_(float("nan"))
    check(lib.apertis_grouped_gemm_tn(ptr(dy), ptr(x), ptr(offs_s), ptr(dw_b), None, Ts, N, K, 1, ptr(ws), wsb, code, stream_ptr()), "tn wide")
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    ea, eb = float((dw_a[0].double() - ref).abs().max()) / sc, float((dw_b[0].double() - ref).abs().max()) / sc
    fl = 2.0 * T * N * K
    print(f"dW [{N:4d},{K:4d}] rows {T}: 128x128 + colsum {ta:7.1f} us ({fl / ta / 1e6:6.0f} TF) | E=1 call (variant {var:2d}) {tb:7.1f} us "
          f"({fl / tb / 1e6:6.0f} TF) | max err / max: {ea:.2e} vs {eb:.2e}", flush=True)
