"""Target of `rocprofv3 --kernel-trace --stats`: the expert MLP (fwd + bwd, bench layer size) with the GEMM tile queues off / on.
    python tools/prof_expert_mlp_queue.py [reps] [batch] [0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 44
ops.GEMM_DYNAMIC_QUEUE = len(sys.argv) > 3 and sys.argv[3] == "1"
torch.manual_seed(1)
H, I, E, p = 704, 2816, 8, 0.1
per = int((batch * 4096 / E) * 1.25)
rows = per * E
offs = torch.arange(E + 1, device=dev, dtype=torch.int32) * per
xg = torch.randn(rows, H, device=dev).bfloat16().requires_grad_(True)
w1 = (torch.randn(E, I, H, device=dev) * 0.03).requires_grad_(True)
b1 = (torch.randn(E, I, device=dev) * 0.1).requires_grad_(True)
w2 = (torch.randn(E, H, I, device=dev) * 0.03).requires_grad_(True)
b2 = (torch.randn(E, H, device=dev) * 0.1).requires_grad_(True)
dy = torch.randn(rows, H, device=dev).bfloat16()
for r in range(reps):
    y = ops.expert_mlp(xg, w1, b1, w2, b2, offs, rows, act="gelu", drop_p=p, seed=12345 + r, compute_dtype=torch.bfloat16)
    y.backward(dy)
    for t in (xg, w1, b1, w2, b2):
        t.grad = None
torch.cuda.synchronize()
print("done", ops.GEMM_DYNAMIC_QUEUE)
