import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_moe_kernels_gpu import _tn_call
dev = torch.device("cuda:0")
sizes, M, N = [700, 100, 0, 513], 256, 128
E, R = len(sizes), sum(sizes)
offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32), device=dev)
torch.manual_seed(0)
A = torch.randn(R, M, device=dev).bfloat16(); Bm = torch.randn(R, N, device=dev).bfloat16()
dW, db = _tn_call(dev, A, Bm, offs, E, True, True)
for e in range(E):
    r0, r1 = int(offs[e]), int(offs[e+1])
    refb = A[r0:r1].double().sum(0).cpu(); refw = (A[r0:r1].double().T @ Bm[r0:r1].double()).cpu()
    eb = (db[e].double().cpu() - refb).abs(); ew = (dW[e].double().cpu() - refw).abs()
    bad = (eb > 1e-2).nonzero().flatten().tolist()
    print(e, "w err", ew.max().item(), "b err", eb.max().item(), "nbad", len(bad), bad[:40])
    if bad: print(" got", db[e, bad[:8]].tolist(), "ref", refb[bad[:8]].tolist())
