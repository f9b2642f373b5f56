"""MoE operators: gate / top-k (+ aux losses), dispatch plan, gather-LayerNorm, combine, the small-batch entrance, the expert MLP.

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import os as _os

import torch

from .. import _lib
from .._lib import check, dtype_code, ptr, stream_ptr
from ._base import _apply, _f32, _grad_wanted, _launch, _require_gpu, _zero_placeholder
from .prep import cast_transpose
from . import gemm as _gemm
from .gemm import _ACTS, _RowsWork, _launch_nt, _tn_workspace


# ----------------------------------------------------------------------------------------------
# MoE: gate, plan, gather+LayerNorm, grouped linear, combine
# ----------------------------------------------------------------------------------------------
class _GateTopK(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, K):
        _require_gpu(logits)
        lib = _lib.load()
        logits = logits.float().contiguous()
        S, E = logits.shape
        dev = logits.device
        gates = torch.empty(S, E, device=dev, dtype=torch.float32)
        idx = torch.empty(S, K, device=dev, dtype=torch.int32)
        w = torch.empty(S, K, device=dev, dtype=torch.float32)
        check(lib.apertis_moe_gate_topk_fwd(ptr(logits), ptr(gates), ptr(idx), ptr(w), S, E, K, stream_ptr()),
              "apertis_moe_gate_topk_fwd")
        ctx.save_for_backward(gates, idx)
        ctx.K = K
        ctx.mark_non_differentiable(idx)
        return gates, idx, w

    @staticmethod
    def backward(ctx, dgates, _didx, dw):
        lib = _lib.load()
        gates, idx = ctx.saved_tensors
        S, E = gates.shape
        dgates = None if dgates is None else dgates.float().contiguous()
        dw = None if dw is None else dw.float().contiguous()
        dlogits = torch.empty_like(gates)
        check(lib.apertis_moe_gate_topk_bwd(ptr(gates), ptr(idx), ptr(dw), ptr(dgates), ptr(dlogits), S, E, ctx.K,
                                            stream_ptr()), "apertis_moe_gate_topk_bwd")
        return dlogits, None


class _GateTopKAux(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, K, lb_coef, rz_coef, w_noise, alpha, seed):
        _require_gpu(logits, w_noise)
        lib = _lib.load()
        logits = logits.float().contiguous()
        wn = None if w_noise is None else _f32(w_noise)
        S, E = logits.shape
        dev = logits.device
        gates = torch.empty(S, E, device=dev, dtype=torch.float32)
        idx = torch.empty(S, K, device=dev, dtype=torch.int32)
        w = torch.empty(S, K, device=dev, dtype=torch.float32)
        lse = torch.empty(S, device=dev, dtype=torch.float32)
        part = torch.empty(lib.apertis_moe_gate_aux_blocks(S), 2 * E + 1, device=dev, dtype=torch.float32)
        stats = torch.empty(2 + E, device=dev, dtype=torch.float32)
        check(lib.apertis_moe_gate_topk_noisy_aux_fwd(ptr(logits), ptr(wn), float(alpha), int(seed), ptr(gates), ptr(idx), ptr(w),
                                                      ptr(lse), ptr(part), ptr(stats), S, E, K, float(lb_coef), float(rz_coef),
                                                      stream_ptr()), "apertis_moe_gate_topk_noisy_aux_fwd")
        ctx.save_for_backward(gates, idx, lse, stats, wn)
        ctx.cfg = (K, float(lb_coef), float(rz_coef), float(alpha), int(seed), None if w_noise is None else w_noise.dtype)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)      # (autograd otherwise zero-fills an [S, K] gradient for idx: one launch per layer)
        return idx, w, stats[0], stats[1]

    @staticmethod
    def backward(ctx, _didx, dw, dlb, drz):
        lib = _lib.load()
        gates, idx, lse, stats, wn = ctx.saved_tensors
        if dw is None and dlb is None and drz is None:
            return None, None, None, None, None, None, None
        K, lb_coef, rz_coef, alpha, seed, wdt = ctx.cfg
        S, E = gates.shape
        dw = None if dw is None else dw.float().contiguous()
        dlb = None if dlb is None else dlb.float().reshape(1).contiguous()
        drz = None if drz is None else drz.float().reshape(1).contiguous()
        dlogits = torch.empty_like(gates)
        npart = dwn = None
        if wn is not None:
            npart = torch.empty(lib.apertis_moe_gate_aux_blocks(S), E, device=gates.device, dtype=torch.float32)
            dwn = torch.empty(E, device=gates.device, dtype=torch.float32)
        check(lib.apertis_moe_gate_topk_noisy_aux_bwd(ptr(gates), ptr(idx), ptr(dw), ptr(lse), ptr(stats), ptr(dlb), ptr(drz),
                                                      lb_coef, rz_coef, ptr(wn), alpha, seed, ptr(dlogits), ptr(npart), ptr(dwn),
                                                      S, E, K, stream_ptr()), "apertis_moe_gate_topk_noisy_aux_bwd")
        return dlogits, None, None, None, (None if dwn is None else dwn.to(wdt)), None, None


def moe_gate_topk_aux(logits, K, lb_coef, rz_coef, w_noise=None, alpha=0.0, seed=0):
    """moe_gate_topk plus the router's two auxiliary losses in the same pass (reference core.py:491-505,
    524-529): returns idx [S,K] int32, w [S,K] fp32, lb_loss and rz_loss (fp32 scalars on the device; a
    coefficient of 0 switches a loss off).  With `w_noise` [E] the reference's noisy top-k routing
    (core.py:485-488: logits += randn * softplus(w_noise) * alpha) happens inside the kernels, the normals drawn from a
    counter hash of `seed`; the gradient of w_noise comes back from the backward kernel."""
    return _GateTopKAux.apply(logits, K, lb_coef, rz_coef, w_noise, alpha, seed)


def moe_gate_topk(logits, K):
    """softmax -> top-K -> renormalised weights (reference core.py:491-492,529).
    Returns gates [S,E] fp32, idx [S,K] int32 (descending probability, ties lowest index),
    w [S,K] fp32."""
    return _GateTopK.apply(logits, K)


class MoePlan:
    """Device-side dispatch plan (reference core.py:547-591), canonical expert-major order."""
    __slots__ = ("offsets", "row_token", "row_k", "slot_of", "S", "E", "K", "max_rows")


def moe_plan(idx, w, E, capacity=None, active=None):
    """idx [S,K] int32, w [S,K] fp32.  capacity None/<=0 = unlimited (eval).  active: optional
    [E] bool mask of experts that are not dropped.  No host sync: row counts stay on the device;
    max_rows is the static bound min(S*K, E*capacity) used to size buffers and grids."""
    _require_gpu(idx, w)
    lib = _lib.load()
    S, K = idx.shape
    dev = idx.device
    idx = idx.to(torch.int32).contiguous()
    w = _f32(w)
    cap = int(capacity) if capacity is not None and capacity > 0 else 0
    p = MoePlan()
    p.S, p.E, p.K = S, E, K
    p.max_rows = min(S * K, E * cap) if cap > 0 else S * K
    p.offsets = torch.empty(E + 1, device=dev, dtype=torch.int32)
    p.row_token = torch.empty(max(S * K, 1), device=dev, dtype=torch.int32)
    p.row_k = torch.empty(max(S * K, 1), device=dev, dtype=torch.int32)
    p.slot_of = torch.empty(S, K, device=dev, dtype=torch.int32)
    ws = torch.empty(lib.apertis_moe_plan_workspace_bytes(S, E, K) // 4 + 1, device=dev, dtype=torch.int32)
    act = None if active is None else active.to(torch.uint8).contiguous()
    check(lib.apertis_moe_plan(ptr(idx), ptr(w), ptr(act), cap, ptr(p.offsets), ptr(p.row_token), ptr(p.row_k),
                               ptr(p.slot_of), ptr(ws), S, E, K, stream_ptr()), "apertis_moe_plan")
    return p


def moe_route_small_supported(logits, x, K):
    """Shapes apertis_moe_route_small takes: a handful of rows (the decode step), inference only."""
    S, E = logits.shape
    # (the entry point takes S <= 64; past 16 rows its one work-group walks the gather-LN rows slower than the row kernel's many)
    return (x.is_cuda and not torch.is_grad_enabled() and 1 <= S <= 16 and E in (4, 8, 16) and E * K <= 16 and K <= E
            and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1024 and x.dtype in (torch.float32, torch.bfloat16))


def moe_route_small(logits, x, gamma, beta, eps, K, out_dtype=None):
    """moe_gate_topk + moe_plan (no capacity, every expert active) + moe_gather_ln for S <= 64 rows as ONE launch (reference
    core.py:491-492,529,547-593 for a single-token step): returns (gates, idx, w, plan, xg) - the same values as the three ops."""
    _require_gpu(logits, x, gamma, beta)
    lib = _lib.load()
    S, E = logits.shape
    H = x.shape[-1]
    dev = x.device
    out_dtype = out_dtype or x.dtype
    lg = logits.float().contiguous()
    x = x.contiguous()
    gates = torch.empty(S, E, device=dev, dtype=torch.float32)
    idx = torch.empty(S, K, device=dev, dtype=torch.int32)
    w = torch.empty(S, K, device=dev, dtype=torch.float32)
    p = MoePlan()
    p.S, p.E, p.K, p.max_rows = S, E, K, S * K
    p.offsets = torch.empty(E + 1, device=dev, dtype=torch.int32)
    p.row_token = torch.empty(S * K, device=dev, dtype=torch.int32)
    p.row_k = torch.empty(S * K, device=dev, dtype=torch.int32)
    p.slot_of = torch.empty(S, K, device=dev, dtype=torch.int32)
    xg = torch.empty(S * K, H, device=dev, dtype=out_dtype)
    mean = torch.empty(S * K, device=dev, dtype=torch.float32)
    rstd = torch.empty(S * K, device=dev, dtype=torch.float32)
    check(lib.apertis_moe_route_small(ptr(lg), ptr(gates), ptr(idx), ptr(w), ptr(p.offsets), ptr(p.row_token), ptr(p.row_k),
                                      ptr(p.slot_of), ptr(x), ptr(_f32(gamma)), ptr(_f32(beta)), float(eps), ptr(xg), ptr(mean),
                                      ptr(rstd), S, H, E, K, dtype_code(x), dtype_code(xg), stream_ptr()), "apertis_moe_route_small")
    return gates, idx, w, p, xg


def moe_enter_small_supported(blk, res, E, K):
    """Shapes apertis_moe_enter_small takes: <= 16 rows of an fp32 residual stream under no_grad (the decode step)."""
    S = res.numel() // res.shape[-1]
    H = res.shape[-1]
    # the kernel's own LDS bound (csrc/moe_routing.hip, apertis_moe_enter_small: the boundary / router / expert affine vectors
    # in fp32 plus the S block rows in their own dtype, plus 4 KiB for its static tables, <= 160 KiB): a shape past it must
    # take the general path HERE - by the time the launch declined it, _decode_prepass has already advanced every layer's SSM state
    lds = (3 * E + 4) * H * 4 + S * H * blk.element_size() + 4096
    return (res.is_cuda and not torch.is_grad_enabled() and 1 <= S <= 16 and E in (4, 8) and E * K <= 16 and K <= E
            and H % 4 == 0 and H <= 1024 and lds <= 160 * 1024 and res.dtype == torch.float32
            and blk.dtype in (torch.float32, torch.bfloat16) and tuple(blk.shape) == tuple(res.shape))


def moe_enter_small(blk, res, weight, bias, eps, r_ln_w, r_ln_b, r_eps, r_w, r_b, e_ln_w, e_ln_b, e_eps, K):
    """dropout_add_layer_norm_router (inference: no dropout) + moe_route_small as ONE launch for <= 16 rows: the residual
    stream y = res + blk, the router's logits on LayerNorm(y), gate, plan and the per-expert LayerNorm of the routed rows
    (reference core.py:888,847,481-482,491-492,529,547-593 for a single-token step).  Returns (y, logits, w, plan, xg)."""
    _require_gpu(blk, res, weight, bias, r_ln_w, r_ln_b, r_w, e_ln_w, e_ln_b)
    lib = _lib.load()
    shape = res.shape
    H = shape[-1]
    E = r_w.shape[0]
    blk2 = blk.reshape(-1, H).contiguous()
    res2 = res.reshape(-1, H).contiguous()
    S = res2.shape[0]
    dev = res.device
    y = torch.empty_like(res2)
    logits = torch.empty(S, E, device=dev, dtype=torch.float32)
    gates = torch.empty(S, E, device=dev, dtype=torch.float32)
    idx = torch.empty(S, K, device=dev, dtype=torch.int32)
    w = torch.empty(S, K, device=dev, dtype=torch.float32)
    p = MoePlan()
    p.S, p.E, p.K, p.max_rows = S, E, K, S * K
    p.offsets = torch.empty(E + 1, device=dev, dtype=torch.int32)
    p.row_token = torch.empty(S * K, device=dev, dtype=torch.int32)
    p.row_k = torch.empty(S * K, device=dev, dtype=torch.int32)
    p.slot_of = torch.empty(S, K, device=dev, dtype=torch.int32)
    xg = torch.empty(S * K, H, device=dev, dtype=blk2.dtype)
    mean = torch.empty(S * K, device=dev, dtype=torch.float32)
    rstd = torch.empty(S * K, device=dev, dtype=torch.float32)
    check(lib.apertis_moe_enter_small(ptr(blk2), ptr(res2), ptr(_f32(weight)), ptr(_f32(bias)), float(eps), ptr(y), None,
                                      ptr(_f32(r_ln_w)), ptr(_f32(r_ln_b)), float(r_eps), ptr(_f32(r_w)),
                                      ptr(None if r_b is None else _f32(r_b)), ptr(logits), ptr(gates), ptr(idx), ptr(w),
                                      ptr(p.offsets), ptr(p.row_token), ptr(p.row_k), ptr(p.slot_of), ptr(_f32(e_ln_w)),
                                      ptr(_f32(e_ln_b)), float(e_eps), ptr(xg), ptr(mean), ptr(rstd), S, H, E, K,
                                      dtype_code(res2), dtype_code(blk2), stream_ptr()), "apertis_moe_enter_small")
    return y.reshape(shape), logits, w, p, xg


class _GatherLN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, plan, eps, out_dtype, link):
        _require_gpu(x, gamma, beta)
        lib = _lib.load()
        x = x.contiguous()
        S, H = x.shape
        ctx.link = link
        g = _f32(gamma)
        b = _f32(beta)
        dev = x.device
        R = max(plan.max_rows, 1)
        xg = torch.empty(R, H, device=dev, dtype=out_dtype)
        mean = torch.empty(R, device=dev, dtype=torch.float32)
        rstd = torch.empty(R, device=dev, dtype=torch.float32)
        check(lib.apertis_moe_gather_ln_fwd(ptr(x), ptr(plan.row_token), ptr(plan.offsets), ptr(g), ptr(b), float(eps),
                                            ptr(xg), ptr(mean), ptr(rstd), plan.max_rows, H, plan.E, dtype_code(x),
                                            dtype_code(xg), stream_ptr()), "apertis_moe_gather_ln_fwd")
        ctx.save_for_backward(x, g, mean, rstd)
        ctx.plan = plan
        return xg

    @staticmethod
    def backward(ctx, dxg):
        lib = _lib.load()
        x, g, mean, rstd = ctx.saved_tensors
        plan = ctx.plan
        S, H = x.shape
        dev = x.device
        dxg = dxg.contiguous()
        dxr = torch.empty_like(dxg)
        dgb = torch.zeros(2, plan.E, H, device=dev, dtype=torch.float32)      # one fill for both accumulators
        dgamma, dbeta = dgb[0], dgb[1]
        nblk = lib.apertis_moe_gather_ln_bwd_blocks(plan.max_rows)
        part = torch.empty(nblk, 2 * H, device=dev, dtype=torch.float32)
        blk_e = torch.empty(nblk, device=dev, dtype=torch.int32)
        check(lib.apertis_moe_gather_ln_bwd(ptr(x), ptr(plan.row_token), ptr(plan.offsets), ptr(g), ptr(mean), ptr(rstd),
                                            ptr(dxg), ptr(dxr), ptr(dgamma), ptr(dbeta), ptr(part), ptr(blk_e), plan.max_rows,
                                            H, plan.E, dtype_code(x), dtype_code(dxg), stream_ptr()),
              "apertis_moe_gather_ln_bwd")
        if ctx.link is not None and plan.K <= 2 and dxr.dtype == x.dtype and ctx.link.rows is None:
            # the consumer of this gradient is the router op that handed x through: it gathers the rows itself
            ctx.link.rows, ctx.link.slot_of, ctx.link.K = dxr, plan.slot_of, plan.K
            return _zero_placeholder((S, H), dev, x.dtype), dgamma, dbeta, None, None, None, None
        dx = torch.empty(S, H, device=dev, dtype=x.dtype)
        check(lib.apertis_moe_combine_fwd(ptr(dxr), ptr(plan.slot_of), None, ptr(dx), S, H, plan.K, 0, dtype_code(dxr),
                                          dtype_code(dx), stream_ptr()), "apertis_moe_combine_fwd(scatter)")
        return dx, dgamma, dbeta, None, None, None, None


ROWS_GRADIENT = True   # tests switch it off to compare with the dense hand-over


def moe_gather_ln(x, gamma, beta, plan, eps, out_dtype=None):
    """xg[r] = LayerNorm_e(x[token(r)]) for every kept row r, expert-sorted (reference core.py:593
    gather + :436 per-expert LayerNorm).  x [S,H]; gamma/beta [E,H]."""
    link = getattr(x, "_apertis_rows_link", None) if ROWS_GRADIENT else None
    return _GatherLN.apply(x, gamma, beta, plan, eps, out_dtype or x.dtype, link)


class _Combine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, yr, w, plan, out_dtype):
        _require_gpu(yr, w)
        lib = _lib.load()
        yr = yr.contiguous()
        wf = w.float().contiguous()
        H = yr.shape[1]
        out = torch.empty(plan.S, H, device=yr.device, dtype=out_dtype)
        check(lib.apertis_moe_combine_fwd(ptr(yr), ptr(plan.slot_of), ptr(wf), ptr(out), plan.S, H, plan.K, 1,
                                          dtype_code(yr), dtype_code(out), stream_ptr()), "apertis_moe_combine_fwd")
        ctx.save_for_backward(yr, wf)
        ctx.plan = plan
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        yr, wf = ctx.saved_tensors
        plan = ctx.plan
        H = yr.shape[1]
        dout = dout.contiguous()
        dyr = torch.empty_like(yr)
        dw = torch.zeros(plan.S, plan.K, device=yr.device, dtype=torch.float32)
        check(lib.apertis_moe_combine_bwd(ptr(dout), ptr(yr), ptr(plan.row_token), ptr(plan.row_k), ptr(plan.offsets),
                                          ptr(wf), ptr(dyr), ptr(dw), plan.max_rows, plan.S, H, plan.K, plan.E,
                                          dtype_code(dout), dtype_code(yr), stream_ptr()), "apertis_moe_combine_bwd")
        return dyr, dw, None, None


def moe_combine(yr, w, plan, out_dtype=None):
    """out[s] = sum_k w[s,k] * yr[slot(s,k)] over kept assignments, k ascending (reference
    core.py:594,605); dropped tokens give exact zeros."""
    return _Combine.apply(yr, w, plan, out_dtype or yr.dtype)


FUSE_ACT_BWD = not _os.environ.get("APERTIS_NO_FUSE_ACT_BWD")


# APERTIS_NT2I=1 (round 6, default OFF): the saved-gradient forward on grouped_gemm_nt2i_k - one wave per SIMD, the epilogue of
# tile i between the MFMA groups of tile i + 1.  Bit-identical outputs; 15 % slower than the ring kernel at the bench shape
# (profiles/r6_probe_nt2i_vs_nt4r.log): kept for the record and under test, not used.
NT2I = _os.environ.get("APERTIS_NT2I") == "1"
SAVE_ACT_GRAD = True    # expert MLP: the forward leaves act'(pre) * mask / (1-p) instead of pre (tests switch it off to compare)


def grad_destination(param, shape, device):
    """Where a backward kernel should write the fp32 gradient of `param`: a fresh alias of the slice the data-parallel
    wrapper reserved for it in its bucket (`param._apertis_grad_view`, parallel.BucketedDataParallel) when the parameter
    has no gradient yet - autograd then adopts that tensor as param.grad and nothing is copied into the bucket - else a
    new tensor (accumulation micro-steps add into the bucket in place)."""
    view = getattr(param, "_apertis_grad_view", None)
    if view is not None and param.grad is None and view.dtype == torch.float32 and tuple(view.shape) == tuple(shape):
        return view.view_as(view)
    return torch.empty(shape, device=device, dtype=torch.float32)


class _ExpertMLP(torch.autograd.Function):
    """yr = (dropout(act(xg @ W1[e].T + b1[e]))) @ W2[e].T + b2[e] per group, as ONE autograd node so the
    backward can fuse act'/dropout into the epilogue of the second layer's data-gradient GEMM."""

    @staticmethod
    def forward(ctx, xg, w1, b1, w2, b2, offsets, max_rows, act, drop_p, seed, cd):
        _require_gpu(xg, w1, w2, offsets)
        lib = _lib.load()
        E, I, H = w1.shape
        xg = xg.to(cd).contiguous()
        need = _grad_wanted(ctx, 5)
        w1c, w1t = cast_transpose(w1, cd, want_transposed=need, cache=not need)
        w2c, w2t = cast_transpose(w2, cd, want_transposed=need, cache=not need)
        b1f, b2f = _f32(b1), _f32(b2)
        code, act_code = dtype_code(xg), _ACTS[act]
        R = xg.shape[0]
        h = torch.empty(R, I, device=xg.device, dtype=cd)
        pre = torch.empty_like(h) if need else None
        # second output: the pre-activation, or - where the kernel offers it - g' = act'(pre) * keep / (1-p) itself, which
        # the data-gradient epilogue of the backward then only multiplies by (no activation derivative, no mask hash there;
        # one evaluation per element yields both outputs in the forward)
        saved_grad = bool(need and FUSE_ACT_BWD and SAVE_ACT_GRAD and
                          lib.apertis_grouped_gemm_nt_saves_grad(max_rows, I, H, w1c.shape[-1], E, act_code, code, code))
        _launch_nt("apertis_grouped_gemm_nt", lib,
                (ptr(xg), ptr(w1c), ptr(b1f), ptr(offsets), ptr(h), ptr(pre), None, max_rows, I, H, w1c.shape[-1], E,
                 act_code | ((_lib.ACT_SAVE_GRAD | (_lib.ACT_INTERLEAVED if NT2I else 0)) if saved_grad else 0), float(drop_p),
                 int(seed), code, code, stream_ptr()),
                _RowsWork(offsets, E, 2.0 * I * H), xg.device)
        yr = torch.empty(R, H, device=xg.device, dtype=cd)
        _launch_nt("apertis_grouped_gemm_nt", lib,
                (ptr(h), ptr(w2c), ptr(b2f), ptr(offsets), ptr(yr), None, None, max_rows, H, I, w2c.shape[-1], E, _lib.ACT_NONE,
                 0.0, 0, code, code, stream_ptr()), _RowsWork(offsets, E, 2.0 * I * H), h.device)
        ctx.save_for_backward(xg, pre, h, w1t, w2t, offsets)
        ctx.cfg = (E, I, H, max_rows, act_code, float(drop_p), int(seed), w1.dtype, w2.dtype)
        ctx.saved_grad = saved_grad
        ctx.wparams = (w1, w2)     # for grad_destination() in the backward
        return yr

    @staticmethod
    def backward(ctx, dyr):
        lib = _lib.load()
        xg, pre, h, w1t, w2t, offsets = ctx.saved_tensors
        E, I, H, max_rows, act_code, drop_p, seed, w1dt, w2dt = ctx.cfg
        code = dtype_code(xg)
        dev = xg.device
        dyr = dyr.to(xg.dtype).contiguous()
        work = _RowsWork(offsets, E, 2.0 * I * H)
        dpre = torch.empty_like(h)
        if ctx.saved_grad:
            # dpre = (dyr @ W2) * g' with the g' the forward left in `pre`
            _launch_nt("apertis_grouped_gemm_nt", lib,
                    (ptr(dyr), ptr(w2t), None, ptr(offsets), ptr(dpre), None, ptr(pre), max_rows, I, H, w2t.shape[-1], E,
                     _lib.ACT_MUL_SAVED, 0.0, 0, code, code, stream_ptr()), work, dyr.device)
        elif FUSE_ACT_BWD:
            # dpre = (dyr @ W2) * keep/(1-p) * act'(pre): layer 1's activation backward in the dgrad epilogue
            _launch_nt("apertis_grouped_gemm_nt", lib,
                    (ptr(dyr), ptr(w2t), None, ptr(offsets), ptr(dpre), None, ptr(pre), max_rows, I, H, w2t.shape[-1], E, act_code,
                     drop_p, seed, code, code, stream_ptr()), work, dyr.device)
        else:
            # APERTIS_NO_FUSE_ACT_BWD=1 (A/B switch): plain data gradient, then the separate bandwidth-bound pass in
            # place on dpre.  On the 256x256 persistent kernel the fused form was slower (nothing overlaps its
            # epilogue); on the two-per-CU kernel it is +3 % of the whole step
            _launch_nt("apertis_grouped_gemm_nt", lib,
                    (ptr(dyr), ptr(w2t), None, ptr(offsets), ptr(dpre), None, None, max_rows, I, H, w2t.shape[-1], E, _lib.ACT_NONE,
                     0.0, 0, code, code, stream_ptr()), work, dyr.device)
            if act_code != _lib.ACT_NONE or drop_p > 0:
                check(lib.apertis_act_dropout_bwd(ptr(dpre), ptr(pre), ptr(dpre), ptr(offsets), max_rows, I, E, act_code,
                                                  drop_p, seed, code, stream_ptr()), "apertis_act_dropout_bwd")
        dxg = None
        if ctx.needs_input_grad[0]:
            dxg = torch.empty_like(xg)
            _launch_nt("apertis_grouped_gemm_nt", lib,
                    (ptr(dpre), ptr(w1t), None, ptr(offsets), ptr(dxg), None, None, max_rows, H, I, w1t.shape[-1], E, _lib.ACT_NONE,
                     0.0, 0, code, code, stream_ptr()), work, dpre.device)
        # both weight gradients in ONE launch: dW2 = dyr^T h, dW1 = dpre^T xg
        dw2 = grad_destination(ctx.wparams[1], (E, H, I), dev)
        db2 = torch.empty(E, H, device=dev, dtype=torch.float32)
        dw1 = grad_destination(ctx.wparams[0], (E, I, H), dev)
        db1 = torch.empty(E, I, device=dev, dtype=torch.float32)
        ws, ws_bytes = _tn_workspace(E, 2, dev, max_rows)
        # (item queue of the weight-gradient kernels: only with TN_DYNAMIC_QUEUE on top of GEMM_DYNAMIC_QUEUE - measured slower
        # than static shares under a CU hog with the 352-wide tiles)
        _launch("apertis_grouped_gemm_tn", lib.apertis_grouped_gemm_tn_pair_q,
                (ptr(dyr), ptr(h), ptr(dw2), ptr(db2), H, I, ptr(dpre), ptr(xg), ptr(dw1), ptr(db1), I, H, ptr(offsets),
                 max_rows, E, ptr(ws), ws_bytes, code, int(_gemm.GEMM_DYNAMIC_QUEUE and _gemm.TN_DYNAMIC_QUEUE), stream_ptr()), _RowsWork(offsets, E, 4.0 * I * H))
        return dxg, dw1.to(w1dt), db1, dw2.to(w2dt), db2, None, None, None, None, None, None


def expert_mlp(xg, w1, b1, w2, b2, offsets, max_rows, act="gelu", drop_p=0.0, seed=0, compute_dtype=None):
    """Grouped expert MLP (reference core.py:437-440): Linear(H->I) -> act -> Dropout -> Linear(I->H) for
    expert-sorted rows xg [R,H]; w1 [E,I,H], b1 [E,I], w2 [E,H,I], b2 [E,H] (fp32 masters)."""
    return _apply(_ExpertMLP, xg, w1, b1, w2, b2, offsets, max_rows, act, drop_p, seed, compute_dtype or xg.dtype)
