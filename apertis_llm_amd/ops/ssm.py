"""SSM companions: depthwise causal conv + SiLU, residual dropout-add, post-scan gate, column splits of a projection output.

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import os as _os

import torch

from .. import _lib
from .._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr
from ._base import _ColSlot, _f32, _grad_out, _require_gpu, _rows, _slot_of


# ----------------------------------------------------------------------------------------------
# SSM companions: depthwise causal conv + SiLU, post-scan gate
# ----------------------------------------------------------------------------------------------
class _DwConvSilu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        _require_gpu(x, w, b)
        lib = _lib.load()
        B, L, Dn = x.shape
        ctx.slot = _slot_of(x)
        x, x_rs = _rows(x, Dn)
        w2 = _f32(w).reshape(Dn, -1)
        b2 = _f32(b)
        k = w2.shape[1]
        out = torch.empty(B, L, Dn, device=x.device, dtype=x.dtype)
        check(lib.apertis_dwconv_silu_fwd(ptr(x), x_rs, ptr(w2), ptr(b2), ptr(out), Dn, B, L, Dn, k, dtype_code(x),
                                          stream_ptr()), "apertis_dwconv_silu_fwd")
        ctx.save_for_backward(x, w2, b2)
        ctx.wshape = w.shape
        return out

    @staticmethod
    def backward(ctx, dout, dout2=None):
        lib = _lib.load()
        x, w2, b2 = ctx.saved_tensors
        B, L, Dn = x.shape
        k = w2.shape[1]
        if dout is None:
            dout, dout2 = dout2, None
        if dout is None:
            return None, None, None
        # (the kernel reads both gradients in the io dtype: the "same bits as autograd's add" claim of the pair form holds only
        # then - a consumer that hands back another dtype is cast here, as autograd's own accumulation would cast it)
        dout = dout.to(x.dtype).contiguous()
        if dout2 is not None:
            dout2 = dout2.to(x.dtype).contiguous()
        nblk = lib.apertis_dwconv_bwd_blocks(B, L, Dn)
        dev = x.device
        dx, dx_rs = _grad_out(ctx.slot, (B, L), Dn, x.dtype, dev)
        dw_part = torch.empty(nblk, Dn, k, device=dev, dtype=torch.float32)
        db_part = torch.empty(nblk, Dn, device=dev, dtype=torch.float32)
        dw = torch.empty(Dn, k, device=dev, dtype=torch.float32)
        db = torch.empty(Dn, device=dev, dtype=torch.float32)
        check(lib.apertis_dwconv_silu_bwd2(ptr(x), x.stride(-2), ptr(w2), ptr(b2), ptr(dout), Dn, ptr(dout2), Dn, ptr(dx), dx_rs,
                                           ptr(dw_part), ptr(db_part), ptr(dw), ptr(db), B, L, Dn, k, dtype_code(x),
                                           stream_ptr()), "apertis_dwconv_silu_bwd")
        return dx, dw.reshape(ctx.wshape), db


# APERTIS_NO_DWCONV_PAIR=1: the conv output as ONE tensor for both consumers (autograd adds their gradients in a pass of its own)
DWCONV_PAIR = not _os.environ.get("APERTIS_NO_DWCONV_PAIR")


class _DwConvSiluPair(_DwConvSilu):
    """The conv output handed out TWICE (two views of one tensor) for its two consumers - x_param_proj and the scan
    (reference core.py:376 and :388-396): their gradients then reach this node separately and the backward kernel adds them
    where it reads the rows, instead of autograd running a [B, L, Dn] add in front of it (30 us per layer at the bench shape;
    the sum is rounded to the io dtype as that add rounds it: the same bits).
    CONTRACT: the two outputs alias one storage, so neither may be written in place by a consumer (autograd marks the second a
    view made inside a custom Function and raises on an in-place op under grad mode; the model's two consumers only read).
    For the grad-enabled path only - without a gradient to merge there is nothing to gain from the alias (dwconv_silu_pair)."""

    @staticmethod
    def forward(ctx, x, w, b):
        out = _DwConvSilu.forward(ctx, x, w, b)
        ctx.set_materialize_grads(False)
        return out, out.view_as(out)


def dwconv_silu(x, weight, bias):
    """silu(causal depthwise conv1d(x)) on token-major x [B,L,Dn] (reference core.py:368-375).
    weight [Dn,1,k] (nn.Conv1d layout), bias [Dn]."""
    return _DwConvSilu.apply(x, weight, bias)


def dwconv_silu_pair(x, weight, bias):
    """dwconv_silu as two views of the one output, one per consumer: see _DwConvSiluPair (under no_grad: one tensor, twice)."""
    if not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad)):
        out = _DwConvSilu.apply(x, weight, bias)
        return out, out
    return _DwConvSiluPair.apply(x, weight, bias)


class _DropoutAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, p, seed):
        _require_gpu(x, res)
        lib = _lib.load()
        x = x.contiguous()
        res = res.contiguous()
        y = torch.empty_like(res)
        check(lib.apertis_dropout_add_fwd(ptr(x), ptr(res), ptr(y), x.numel(), float(p), int(seed), dtype_code(x),
                                          dtype_code(res), stream_ptr()), "apertis_dropout_add_fwd")
        ctx.cfg = (float(p), int(seed), x.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        p, seed, xdt = ctx.cfg
        g = g.contiguous()
        dx = torch.empty(g.shape, device=g.device, dtype=xdt)
        check(lib.apertis_dropout_bwd(ptr(g), ptr(dx), g.numel(), p, seed, dtype_code(g), dtype_code(dx), stream_ptr()),
              "apertis_dropout_bwd")
        return dx, g, None, None


def dropout_add(x, residual, p, training):
    """residual + dropout(x) (reference core.py:836-837, 918-919) in one kernel; the backward regenerates
    the mask from the seed.  x: block output (compute dtype), residual: the fp32 stream."""
    p = float(p) if training else 0.0
    seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p > 0 else 0
    return _DropoutAdd.apply(x, residual, p, seed)


class _SsmGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, xc, z, D):
        _require_gpu(y, xc, z, D)
        lib = _lib.load()
        B, L, Dn = y.shape
        ctx.zslot = _slot_of(z)
        y, y_rs = _rows(y, Dn)
        xc, xc_rs = _rows(xc, Dn)
        z, z_rs = _rows(z, Dn)
        if xc.dtype != z.dtype:
            raise ApertisHipError("xc and z must share a dtype")
        Df = _f32(D)
        out = torch.empty(B, L, Dn, device=y.device, dtype=xc.dtype)
        check(lib.apertis_ssm_gate_fwd(ptr(y), y_rs, ptr(xc), xc_rs, ptr(z), z_rs, ptr(Df), ptr(out), Dn, B * L, Dn,
                                       dtype_code(y), dtype_code(xc), stream_ptr()), "apertis_ssm_gate_fwd")
        ctx.save_for_backward(y, xc, z, Df)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        y, xc, z, Df = ctx.saved_tensors
        B, L, Dn = y.shape
        dout = dout.contiguous()
        dev = y.device
        nblk = lib.apertis_ssm_gate_bwd_blocks(B * L, Dn)
        dy = torch.empty(B, L, Dn, device=dev, dtype=y.dtype)
        dxc = torch.empty(B, L, Dn, device=dev, dtype=xc.dtype)
        dz, dz_rs = _grad_out(ctx.zslot, (B, L), Dn, z.dtype, dev)
        part = torch.empty(nblk, Dn, device=dev, dtype=torch.float32)
        dD = torch.empty(Dn, device=dev, dtype=torch.float32)
        check(lib.apertis_ssm_gate_bwd(ptr(dout), Dn, ptr(y), y.stride(-2), ptr(xc), xc.stride(-2), ptr(z), z.stride(-2),
                                       ptr(Df), ptr(dy), Dn, ptr(dxc), Dn, ptr(dz), dz_rs, ptr(part), ptr(dD), B * L, Dn,
                                       dtype_code(y), dtype_code(xc), stream_ptr()), "apertis_ssm_gate_bwd")
        return dy, dxc, dz, dD


def ssm_gate(y, xc, z, D):
    """(y + D*xc) * silu(z)  (reference core.py:395-396); y fp32 or bf16, xc/z/out share a dtype."""
    return _SsmGate.apply(y, xc, z, D)


class _SplitCols(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)     # an unused piece (the pad columns) arrives as None: zeroed in place below, not
        return tuple(x[..., a:a + n] for a, n in zip(slot.offsets, slot.widths))   # as a materialised zero tensor + a copy

    @staticmethod
    def backward(ctx, *grads):
        slot = ctx.slot
        buf, slot.buf = slot.buf, None
        zeroed, slot.zeroed = slot.zeroed, set()
        if buf is None:
            ref = next((g for g in grads if g is not None), None)
            if ref is None:
                return None, None
            parts = [g if g is not None else ref.new_zeros(*ref.shape[:-1], n) for g, n in zip(grads, slot.widths)]
            return torch.cat(parts, dim=-1), None
        es = buf.element_size()
        for j, (g, off, n) in enumerate(zip(grads, slot.offsets, slot.widths)):
            if n == 0:
                continue
            dst = buf[..., off:off + n]
            if g is None:
                if j not in zeroed:          # (else: the neighbouring view's kernel has written the zeros - _ColSlot.zero_next)
                    dst.zero_()
            elif not (g.data_ptr() == buf.data_ptr() + off * es and g.stride() == dst.stride()):
                dst.copy_(g)      # produced by an op that does not know the protocol
        return buf, None


def split_cols(x, sizes):
    """Column views x[..., a:b] of consecutive widths `sizes` (summing to x.shape[-1]).  The backward assembles the
    input gradient without a pass of its own when the consumers are ops of this module (they write into one shared
    buffer, see _ColSlot), with one concatenation otherwise; plain slicing leaves autograd to zero-fill, copy and
    add a full-width tensor per slice."""
    assert sum(sizes) == x.shape[-1]
    slot = _ColSlot(sizes)
    outs = _SplitCols.apply(x, slot)
    for i, o in enumerate(outs):
        o._apertis_slot = (slot, i)
    return outs
