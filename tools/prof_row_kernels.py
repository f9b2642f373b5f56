"""Target of `rocprofv3 --kernel-trace --stats`: the ROW kernels of one MoE layer boundary pair at the bench's full per-GPU size
(T = 44 x 4096 tokens, H = 704, 8 experts top-2 with the training capacity) called one by one on valid synthetic inputs - each
op's outputs are consumed by nothing else, so a probe build of the library whose row kernels compile their data stores out
(-DROW_PROBE_NOSTORE) can be timed next to the real one.    python tools/prof_row_kernels.py [reps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from apertis_llm_amd import ops
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B = int(sys.argv[2]) if len(sys.argv) > 2 else 44
torch.manual_seed(0)
L, H, E, K, p = 4096, 704, 8, 2, 0.1
T = B * L
cap = max(1, int(T / E * 1.25))
bf = torch.bfloat16
res = torch.randn(T, H, device=dev).requires_grad_(True)                 # the fp32 residual stream
blk = torch.randn(T, H, device=dev).to(bf).requires_grad_(True)          # a sub-block's output
g, b_ = (torch.randn(H, device=dev).requires_grad_(True) for _ in range(2))
rg, rb = (torch.randn(H, device=dev).requires_grad_(True) for _ in range(2))
rw = (torch.randn(E, H, device=dev) * 0.02).requires_grad_(True)
rbias = torch.zeros(E, device=dev, requires_grad=True)
eg, eb = (torch.randn(E, H, device=dev).requires_grad_(True) for _ in range(2))
dy = torch.randn(T, H, device=dev)
dxn = torch.randn(T, H, device=dev).to(bf)
for r in range(reps):
    # boundary in front of the MoE block: y = res + dropout(blk), xn = LN(y), router logits
    y, xn, logits = ops.dropout_add_layer_norm_router(blk, res, g, b_, 1e-12, p, True, rg, rb, 1e-12, rw, rbias, out_dtype=bf)
    idx, w, lb, rz = ops.moe_gate_topk_aux(logits, K, 0.01, 0.001)
    plan = ops.moe_plan(idx, w, E, cap)
    xg = ops.moe_gather_ln(xn, eg, eb, plan, 1e-12, out_dtype=bf)
    yr = torch.randn(plan.max_rows, H, device=dev).to(bf).requires_grad_(True)       # stands for the expert MLP's output rows
    # boundary behind the MoE block: the combine of the expert rows inside the next boundary kernel
    y2, xn2 = ops.dropout_add_layer_norm(yr, y, g, b_, 1e-12, p, True, out_dtype=bf, combine=(w, plan))
    # plain pre-norm (pass-through form) as the SSM block's entry uses it
    xn3, y3 = ops.layer_norm_pass(y2, g, b_, 1e-12, out_dtype=bf)
    loss_terms = [(xn3.float() * 1e-3).sum(), (xn2.float() * 1e-3).sum(), (y3 * 1e-3).sum(), lb, rz,
                  (xg.float() * 1e-3).sum()]
    torch.autograd.backward(loss_terms)
    for t in (res, blk, g, b_, rg, rb, rw, rbias, eg, eb):
        t.grad = None
torch.cuda.synchronize()
print("done", ops.scan_gate_error())
