"""LayerNorm family: plain, residual + dropout + LayerNorm block boundaries, the router norm + projection, their fused backward forms.

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import os as _os

import torch

from .. import _lib
from .._lib import check, dtype_code, ptr, stream_ptr
from ._base import _RowsGrad, _f32, _require_gpu


class _RouterLN(torch.autograd.Function):
    """(Linear(LayerNorm(x)), x): the router projection with its norm fused in, handing x through so the
    gradient of x's other consumers (the expert path) is added inside the backward kernel."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, eps, weight, bias):
        _require_gpu(x, ln_w, ln_b, weight, bias)
        lib = _lib.load()
        x = x.contiguous()
        T, H = x.shape
        N = weight.shape[0]
        g, be = _f32(ln_w), _f32(ln_b)
        w = _f32(weight)
        b = None if bias is None else _f32(bias)
        logits = torch.empty(T, N, device=x.device, dtype=torch.float32)
        mean = torch.empty(T, device=x.device, dtype=torch.float32)
        rstd = torch.empty(T, device=x.device, dtype=torch.float32)
        check(lib.apertis_router_fwd(ptr(x), ptr(g), ptr(be), float(eps), ptr(w), ptr(b), ptr(logits), ptr(mean), ptr(rstd),
                                     T, H, N, dtype_code(x), stream_ptr()), "apertis_router_fwd")
        ctx.save_for_backward(x, g, be, mean, rstd, w)
        ctx.cfg = (ln_w.dtype, ln_b.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.link = _RowsGrad()
        return logits, x.view_as(x), ctx.link

    @staticmethod
    def backward(ctx, dlogits, dpass, _dlink=None):
        lib = _lib.load()
        x, g, be, mean, rstd, w = ctx.saved_tensors
        T, H = x.shape
        N = w.shape[0]
        rows, slot_of, KS = ctx.link.take()
        if dpass is not None and rows is not None and dpass.stride() == (0,) * dpass.dim():
            dpass = None                                        # the gather op's placeholder: its gradient is `rows`
        if dlogits is None:
            if rows is not None:                                # (router output unused: form the dense gradient after all)
                dense = torch.empty_like(x)
                check(lib.apertis_moe_combine_fwd(ptr(rows), ptr(slot_of), None, ptr(dense), T, H, KS, 0, dtype_code(rows),
                                                  dtype_code(dense), stream_ptr()), "apertis_moe_combine_fwd(scatter)")
                dpass = dense if dpass is None else dpass + dense
            return dpass, None, None, None, None, None
        dlogits = dlogits.float().contiguous()
        if dpass is not None:
            dpass = dpass.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        nblk = lib.apertis_router_bwd_blocks(T)
        cols = N * H + N + 2 * H
        part = torch.empty(nblk, cols, device=x.device, dtype=torch.float32)
        out = torch.empty(cols, device=x.device, dtype=torch.float32)
        check(lib.apertis_router_bwd_rows(ptr(x), ptr(g), ptr(be), ptr(mean), ptr(rstd), ptr(w), ptr(dlogits), ptr(dpass),
                                          ptr(rows), ptr(slot_of), KS, ptr(dx), ptr(part), ptr(out), T, H, N, dtype_code(x),
                                          stream_ptr()), "apertis_router_bwd_rows")
        gdt, bedt, wdt, bdt = ctx.cfg
        dW, db = out[:N * H].reshape(N, H), out[N * H:N * H + N]
        dg, dbe = out[N * H + N:N * H + N + H], out[N * H + N + H:]
        return dx, dg.to(gdt), dbe.to(bedt), None, dW.to(wdt), (db.to(bdt) if bdt is not None else None)


def router_ln_linear_supported(x, H, N):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and N in (2, 4, 8) and H % 4 == 0 and H <= 1024


def router_ln_linear(x, ln_weight, ln_bias, eps, weight, bias=None):
    """(fp32 logits [T,N], x) with logits = Linear(LayerNorm(x)) (reference core.py:481-482) in one pass over
    x; route x's other uses through the returned x so their gradient is folded into this op's backward."""
    logits, xp, link = _RouterLN.apply(x, ln_weight, ln_bias, eps, weight, bias)
    xp._apertis_rows_link = link          # moe_gather_ln(xp, ...) hands its gradient over as rows (see _RowsGrad)
    return logits, xp


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        _require_gpu(x, weight, bias)
        lib = _lib.load()
        shape = x.shape
        H = shape[-1]
        x2 = x.reshape(-1, H).contiguous()
        T = x2.shape[0]
        g = _f32(weight)
        b = _f32(bias)
        y = torch.empty(T, H, device=x.device, dtype=out_dtype)
        mean = torch.empty(T, device=x.device, dtype=torch.float32)
        rstd = torch.empty(T, device=x.device, dtype=torch.float32)
        check(lib.apertis_layernorm_fwd(ptr(x2), ptr(g), ptr(b), float(eps), ptr(y), ptr(mean), ptr(rstd), T, H,
                                        dtype_code(x2), dtype_code(y), stream_ptr()), "apertis_layernorm_fwd")
        ctx.save_for_backward(x2, g, mean, rstd)
        ctx.shape = shape
        ctx.pdtypes = (weight.dtype, bias.dtype)
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy, dres=None):
        lib = _lib.load()
        x2, g, mean, rstd = ctx.saved_tensors
        T, H = x2.shape
        if dy is None:      # the normalised output was not used: only the pass-through carries gradient
            return dres, None, None, None, None
        dy2 = dy.reshape(T, H).contiguous()
        if dres is not None:
            dres = dres.reshape(T, H).to(x2.dtype).contiguous()
        dx = torch.empty(T, H, device=x2.device, dtype=x2.dtype)
        nw = lib.apertis_layernorm_bwd_blocks(T, H)
        part = torch.empty(nw, 2, H, device=x2.device, dtype=torch.float32)
        dg = torch.empty(H, device=x2.device, dtype=torch.float32)
        db = torch.empty(H, device=x2.device, dtype=torch.float32)
        check(lib.apertis_layernorm_bwd(ptr(x2), ptr(g), ptr(mean), ptr(rstd), ptr(dy2), ptr(dres), ptr(dx), None, 0.0, 0,
                                        ptr(part), ptr(dg), ptr(db), T, H, dtype_code(x2), dtype_code(dy2), stream_ptr()),
              "apertis_layernorm_bwd")
        return dx.reshape(ctx.shape), dg.to(ctx.pdtypes[0]), db.to(ctx.pdtypes[1]), None, None


class _LayerNormPass(_LayerNorm):
    """LayerNorm that also hands its input through: (LN(x), x).  In a pre-norm residual block
    y = x + f(LN(x)) the residual add reads the pass-through, so both gradients reach this node together
    and the backward kernel adds them (dx = LN backward + dres) instead of autograd running a separate
    full-width add."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        y = _LayerNorm.forward(ctx, x, weight, bias, eps, out_dtype)
        return y, x.view_as(x)


class _DropoutAddLN(torch.autograd.Function):
    """(y, xn) = (res + dropout(blk), LayerNorm(y)): the boundary between two pre-norm sub-blocks as one
    node.  Forward: y is written once and normalised in the same pass; backward: the gradients of y (the
    residual path) and xn arrive together, one kernel writes d_res = LN backward + dy and d_blk = its masked
    copy."""

    @staticmethod
    def forward(ctx, blk, res, weight, bias, eps, p, seed, out_dtype, wk=None, plan=None):
        """plan/wk given: blk is the MoE expert output [rows,H] and the block output is its combine."""
        _require_gpu(blk, res, weight, bias)
        lib = _lib.load()
        shape = res.shape
        H = shape[-1]
        blk2 = blk.reshape(-1, H).to(out_dtype).contiguous()
        res2 = res.reshape(-1, H).contiguous()
        T = res2.shape[0]
        wf = None if plan is None else wk.float().contiguous()
        g = _f32(weight)
        b = _f32(bias)
        y = torch.empty_like(res2)
        xn = torch.empty(T, H, device=res.device, dtype=out_dtype)
        mean = torch.empty(T, device=res.device, dtype=torch.float32)
        rstd = torch.empty(T, device=res.device, dtype=torch.float32)
        check(lib.apertis_dropout_add_layernorm_fwd(ptr(blk2), None if plan is None else ptr(plan.slot_of), ptr(wf),
                                                    0 if plan is None else plan.K, ptr(res2), ptr(g), ptr(b), float(eps), ptr(y),
                                                    ptr(xn), ptr(mean), ptr(rstd), T, H, float(p), int(seed), dtype_code(res2),
                                                    dtype_code(xn), stream_ptr()), "apertis_dropout_add_layernorm_fwd")
        if plan is None:
            ctx.save_for_backward(y, g, mean, rstd)
        else:
            ctx.save_for_backward(y, g, mean, rstd, blk2, wf)
        ctx.plan = plan
        ctx.cfg = (shape, float(p), int(seed), weight.dtype, bias.dtype, blk.dtype, out_dtype, tuple(blk.shape))
        return y.reshape(shape), xn.reshape(shape)

    @staticmethod
    def _to_inputs(ctx, dblk):
        """Gradient of the block-output argument(s) from the token-major dblk [T,H]."""
        lib = _lib.load()
        shape, _p, _seed, _wdt, _bdt, blkdt, _odt, blkshape = ctx.cfg
        plan = ctx.plan
        if plan is None:
            return dblk.reshape(shape).to(blkdt), None
        yr, wf = ctx.saved_tensors[4], ctx.saved_tensors[5]
        H = yr.shape[1]
        dyr = torch.empty_like(yr)
        dw = torch.zeros(plan.S, plan.K, device=yr.device, dtype=torch.float32)
        check(lib.apertis_moe_combine_bwd(ptr(dblk), ptr(yr), ptr(plan.row_token), ptr(plan.row_k), ptr(plan.offsets),
                                          ptr(wf), ptr(dyr), ptr(dw), plan.max_rows, plan.S, H, plan.K, plan.E,
                                          dtype_code(dblk), dtype_code(yr), stream_ptr()), "apertis_moe_combine_bwd")
        return dyr.reshape(blkshape).to(blkdt), dw

    @staticmethod
    def backward(ctx, dy, dxn):
        lib = _lib.load()
        y, g, mean, rstd = ctx.saved_tensors[:4]
        shape, p, seed, wdt, bdt, blkdt, odt, _ = ctx.cfg
        T, H = y.shape
        if dxn is None:      # the normalised output was not used: only the residual path carries gradient
            dy2 = dy.reshape(T, H).contiguous()
            dblk = torch.empty(T, H, device=y.device, dtype=odt)
            check(lib.apertis_dropout_bwd(ptr(dy2), ptr(dblk), dy2.numel(), p, seed, dtype_code(dy2), dtype_code(dblk), stream_ptr()),
                  "apertis_dropout_bwd")
            dblk_in, dwk = _DropoutAddLN._to_inputs(ctx, dblk)
            return dblk_in, dy, None, None, None, None, None, None, dwk, None
        dxn2 = dxn.reshape(T, H).to(odt).contiguous()
        dres = None if dy is None else dy.reshape(T, H).to(y.dtype).contiguous()
        dx = torch.empty_like(y)
        nw = lib.apertis_layernorm_bwd_blocks(T, H)
        part = torch.empty(nw, 2, H, device=y.device, dtype=torch.float32)
        dg = torch.empty(H, device=y.device, dtype=torch.float32)
        db = torch.empty(H, device=y.device, dtype=torch.float32)
        plan = ctx.plan
        if plan is not None and FUSE_COMBINE_BWD and plan.K <= 2:
            # the block output was the MoE combine: its backward inside the LayerNorm backward's pass over the row - the masked
            # gradient rows [T, H] are neither written nor read back (round 6; bit-identical to the two launches below)
            yr, wf = ctx.saved_tensors[4], ctx.saved_tensors[5]
            dyr = torch.empty_like(yr)
            dw = torch.zeros(plan.S, plan.K, device=yr.device, dtype=torch.float32)
            rc = lib.apertis_layernorm_combine_bwd(ptr(y), ptr(g), ptr(mean), ptr(rstd), ptr(dxn2), ptr(dres), ptr(dx), p, seed,
                                                   ptr(part), ptr(dg), ptr(db), ptr(plan.slot_of), ptr(wf), ptr(yr), ptr(dyr), ptr(dw),
                                                   T, H, plan.K, dtype_code(y), dtype_code(dxn2), stream_ptr())
            if rc != -2:
                check(rc, "apertis_layernorm_combine_bwd")
                blkdt, blkshape = ctx.cfg[5], ctx.cfg[7]
                return dyr.reshape(blkshape).to(blkdt), dx.reshape(shape), dg.to(wdt), db.to(bdt), None, None, None, None, dw, None
        dblk = torch.empty(T, H, device=y.device, dtype=odt)
        check(lib.apertis_layernorm_bwd(ptr(y), ptr(g), ptr(mean), ptr(rstd), ptr(dxn2), ptr(dres), ptr(dx), ptr(dblk), p, seed,
                                        ptr(part), ptr(dg), ptr(db), T, H, dtype_code(y), dtype_code(dxn2), stream_ptr()),
              "apertis_layernorm_bwd")
        dblk_in, dwk = _DropoutAddLN._to_inputs(ctx, dblk)
        return dblk_in, dx.reshape(shape), dg.to(wdt), db.to(bdt), None, None, None, None, dwk, None


class _DropoutAddLNRouter(torch.autograd.Function):
    """_DropoutAddLN (dense block output) with the MoE router's projection of its normalised output in the same forward
    pass: (y, xn, logits) = (res + dropout(blk), LayerNorm(y), Linear(router_norm(xn))).  The backward is the two existing
    kernels in sequence: the router backward turns dlogits (+ whatever reached xn: the expert path's gradient rows, see
    _RowsGrad, and any dense term) into the total gradient of xn, the boundary backward takes it from there."""

    @staticmethod
    def forward(ctx, blk, res, weight, bias, eps, p, seed, out_dtype, r_ln_w, r_ln_b, r_eps, r_w, r_b):
        _require_gpu(blk, res, weight, bias, r_ln_w, r_ln_b, r_w, r_b)
        lib = _lib.load()
        shape = res.shape
        H = shape[-1]
        N = r_w.shape[0]
        blk2 = blk.reshape(-1, H).to(out_dtype).contiguous()
        res2 = res.reshape(-1, H).contiguous()
        T = res2.shape[0]
        dev = res.device
        g, b = _f32(weight), _f32(bias)
        rg, rbe = _f32(r_ln_w), _f32(r_ln_b)
        rw = _f32(r_w)
        rbias = None if r_b is None else _f32(r_b)
        y = torch.empty_like(res2)
        xn = torch.empty(T, H, device=dev, dtype=out_dtype)
        mean, rstd = torch.empty(T, device=dev, dtype=torch.float32), torch.empty(T, device=dev, dtype=torch.float32)
        rmean, rrstd = torch.empty(T, device=dev, dtype=torch.float32), torch.empty(T, device=dev, dtype=torch.float32)
        logits = torch.empty(T, N, device=dev, dtype=torch.float32)
        check(lib.apertis_dropout_add_layernorm_router_fwd(ptr(blk2), ptr(res2), ptr(g), ptr(b), float(eps), ptr(y), ptr(xn),
                                                           ptr(mean), ptr(rstd), ptr(rg), ptr(rbe), float(r_eps), ptr(rw),
                                                           ptr(rbias), ptr(logits), ptr(rmean), ptr(rrstd), T, H, N, float(p),
                                                           int(seed), dtype_code(res2), dtype_code(xn), stream_ptr()),
              "apertis_dropout_add_layernorm_router_fwd")
        ctx.save_for_backward(y, g, mean, rstd, xn, rg, rbe, rmean, rrstd, rw)
        ctx.plan = None
        ctx.cfg = (shape, float(p), int(seed), weight.dtype, bias.dtype, blk.dtype, out_dtype, tuple(blk.shape))
        ctx.rcfg = (r_ln_w.dtype, r_ln_b.dtype, r_w.dtype, None if r_b is None else r_b.dtype)
        ctx.link = _RowsGrad()
        ctx.set_materialize_grads(False)      # (an unused output's gradient arrives as None, not as a [T, H] tensor of zeros)
        return y.reshape(shape), xn.reshape(shape), logits, ctx.link

    @staticmethod
    def backward(ctx, dy, dxn, dlogits, _dlink=None):
        lib = _lib.load()
        y, g, mean, rstd, xn, rg, rbe, rmean, rrstd, rw = ctx.saved_tensors
        T, H = y.shape
        N = rw.shape[0]
        rows, slot_of, KS = ctx.link.take()
        if dxn is not None and rows is not None and dxn.stride() == (0,) * dxn.dim():
            dxn = None                                            # the gather op's placeholder: its gradient is `rows`
        r_grads = (None, None, None, None, None)
        if (dlogits is not None and dxn is None and FUSE_ROUTER_BOUNDARY_BWD and y.dtype == torch.float32 and
                (rows is None or KS <= 2)):
            # one pass: the router's dx half + the boundary norm's backward (xn's gradient stays in registers)
            shape, p, seed, wdt, bdt, blkdt, odt, _ = ctx.cfg
            dres = None if dy is None else dy.reshape(T, H).to(y.dtype).contiguous()
            dx = torch.empty_like(y)
            dblk = torch.empty(T, H, device=y.device, dtype=odt)
            nblk = lib.apertis_router_bwd_blocks(T)
            cols = N * H + N + 2 * H
            part = torch.empty(nblk * (cols + 2 * H), device=y.device, dtype=torch.float32)
            out = torch.empty(cols, device=y.device, dtype=torch.float32)
            dg = torch.empty(H, device=y.device, dtype=torch.float32)
            db = torch.empty(H, device=y.device, dtype=torch.float32)
            rc = lib.apertis_boundary_router_bwd(ptr(y), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dblk), p, seed,
                                                 ptr(xn), ptr(rg), ptr(rbe), ptr(rmean), ptr(rrstd), ptr(rw),
                                                 ptr(dlogits.float().contiguous()), ptr(rows), ptr(slot_of), KS or 0, ptr(part),
                                                 ptr(out), ptr(dg), ptr(db), T, H, N, dtype_code(y), dtype_code(xn), stream_ptr())
            if rc != -2:                 # (-2 = APERTIS_ERR_UNSUPPORTED: the two calls below)
                check(rc, "apertis_boundary_router_bwd")
                global FUSED_ROUTER_BWD_CALLS
                FUSED_ROUTER_BWD_CALLS += 1
                gdt, bedt, rwdt, rbdt = ctx.rcfg
                r_grads = (out[N * H + N:N * H + N + H].to(gdt), out[N * H + N + H:].to(bedt), None,
                           out[:N * H].reshape(N, H).to(rwdt), (out[N * H:N * H + N].to(rbdt) if rbdt is not None else None))
                dblk_in, _ = _DropoutAddLN._to_inputs(ctx, dblk)
                return (dblk_in, dx.reshape(shape), dg.to(wdt), db.to(bdt), None, None, None, None) + r_grads
        if dlogits is not None:
            dres = None if dxn is None else dxn.reshape(T, H).to(xn.dtype).contiguous()
            dxn_t = torch.empty_like(xn)
            nblk = lib.apertis_router_bwd_blocks(T)
            cols = N * H + N + 2 * H
            part = torch.empty(nblk, cols, device=y.device, dtype=torch.float32)
            out = torch.empty(cols, device=y.device, dtype=torch.float32)
            check(lib.apertis_router_bwd_rows(ptr(xn), ptr(rg), ptr(rbe), ptr(rmean), ptr(rrstd), ptr(rw),
                                              ptr(dlogits.float().contiguous()), ptr(dres), ptr(rows), ptr(slot_of), KS,
                                              ptr(dxn_t), ptr(part), ptr(out), T, H, N, dtype_code(xn), stream_ptr()),
                  "apertis_router_bwd_rows")
            gdt, bedt, wdt, bdt = ctx.rcfg
            r_grads = (out[N * H + N:N * H + N + H].to(gdt), out[N * H + N + H:].to(bedt), None,
                       out[:N * H].reshape(N, H).to(wdt), (out[N * H:N * H + N].to(bdt) if bdt is not None else None))
            dxn = dxn_t
        elif rows is not None:                                    # router output unused: form the dense gradient after all
            dense = torch.empty_like(xn)
            check(lib.apertis_moe_combine_fwd(ptr(rows), ptr(slot_of), None, ptr(dense), T, H, KS, 0, dtype_code(rows),
                                              dtype_code(dense), stream_ptr()), "apertis_moe_combine_fwd(scatter)")
            dxn = dense if dxn is None else dxn.reshape(T, H) + dense
        base = _DropoutAddLN.backward(ctx, dy, dxn)               # (dblk, dres, dgamma, dbeta, None x 4, dwk, None)
        return base[:8] + r_grads


def dropout_add_layer_norm_router(blk, residual, weight, bias, eps, p, training, r_ln_w, r_ln_b, r_eps, r_w, r_b, out_dtype=None):
    """dropout_add_layer_norm for the boundary in front of an MoE feed-forward, with the router's logits
    (reference core.py:481-482 on the normalised output) formed in the same pass.  Returns (y, xn, logits); xn carries
    the hand-over for moe_gather_ln's gradient rows like router_ln_linear's pass-through does."""
    p = float(p) if training else 0.0
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p > 0 else 0
    y, xn, logits, link = _DropoutAddLNRouter.apply(blk, residual, weight, bias, eps, p, seed, out_dtype or residual.dtype,
                                                    r_ln_w, r_ln_b, r_eps, r_w, r_b)
    xn._apertis_rows_link = link
    return y, xn, logits


def dropout_add_layer_norm(blk, residual, weight, bias, eps, p, training, out_dtype=None, combine=None):
    """(residual + dropout(blk), LayerNorm(of that)) in one pass each way (reference core.py:698 + :847, :888 +
    :667 of the next layer, :1294).  combine=(w, plan): blk is the MoE expert output [rows,H] and the block output
    its weighted combine (core.py:594,605), formed inside the same forward pass."""
    p = float(p) if training else 0.0
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p > 0 else 0
    wk, plan = combine if combine is not None else (None, None)
    return _DropoutAddLN.apply(blk, residual, weight, bias, eps, p, seed, out_dtype or residual.dtype, wk, plan)


def layer_norm(x, weight, bias, eps, out_dtype=None):
    """LayerNorm over the last dimension; x fp32/bf16, statistics in fp32, output in out_dtype
    (bf16 under autocast: the following GEMM reads it directly)."""
    return _LayerNorm.apply(x, weight, bias, eps, out_dtype or x.dtype)


def layer_norm_pass(x, weight, bias, eps, out_dtype=None):
    """(LayerNorm(x), x): see _LayerNormPass."""
    return _LayerNormPass.apply(x, weight, bias, eps, out_dtype or x.dtype)


# APERTIS_NO_FUSE_ROUTER_BWD=1: the router backward and the boundary's LayerNorm backward as two calls (xn's gradient through HBM)
FUSE_ROUTER_BOUNDARY_BWD = not _os.environ.get("APERTIS_NO_FUSE_ROUTER_BWD")
# the MoE combine's backward inside the LayerNorm backward of the boundary behind it (apertis_layernorm_combine_bwd, round 6)
FUSE_COMBINE_BWD = _os.environ.get("APERTIS_FUSE_COMBINE_BWD", "1") == "1"


FUSED_ROUTER_BWD_CALLS = 0        # (times the one-pass form ran: the tests look at it)
