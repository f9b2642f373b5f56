"""Single-token decode step operators (generate()): conv window / state update, the stacked cache-only pre-pass, in_proj with gate epilogue, skinny GEMV.

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import torch

from .. import _lib
from .._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr
from ._base import _f32, _require_gpu
from .prep import cast_transpose


def ssm_decode_step(xp, conv_state, conv_w, conv_b, inplace=False):
    """First half of the single-token SSM step (reference core.py:368-375 with a cached window): returns
    (xc [B,Dn], new conv_state [B,Dn,k-1]).  xp [B,Dn] (a row-strided view is fine), conv_state [B,Dn,k-1]."""
    _require_gpu(xp, conv_state, conv_w, conv_b)
    lib = _lib.load()
    B, Dn = xp.shape
    k = conv_w.shape[-1]
    if xp.stride(-1) != 1:
        xp = xp.contiguous()
    cs = conv_state.to(xp.dtype).contiguous()
    w2 = _f32(conv_w).reshape(Dn, k)
    b2 = _f32(conv_b)
    xc = torch.empty(B, Dn, device=xp.device, dtype=xp.dtype)
    cs_out = cs if (inplace and cs.data_ptr() == conv_state.data_ptr()) else torch.empty_like(cs)   # (in place: the decode graph's cache)
    check(lib.apertis_ssm_decode_conv(ptr(xp), xp.stride(0), ptr(cs), ptr(cs_out), ptr(w2), ptr(b2), ptr(xc), B, Dn, k,
                                      dtype_code(xp), stream_ptr()), "apertis_ssm_decode_conv")
    return xc, cs_out


def ssm_decode_state(dt_logits, A_log, Bt, C, xc, z, D, state, delta_softplus=True):
    """Second half: state <- exp(delta*A)*state + Bt (in place, fp32 [B,Dn]); returns (C*state + D*xc)*silu(z) [B,Dn]
    (reference core.py:347-349 for one token, then :395-396).  Bt / C / z may be row-strided views."""
    _require_gpu(dt_logits, A_log, Bt, C, xc, z, D, state)
    lib = _lib.load()
    B, h = dt_logits.shape
    N = A_log.shape[1]
    Dn = h * N
    fix = lambda t: t if t.stride(-1) == 1 else t.contiguous()
    Bt, C, z = fix(Bt), fix(C), fix(z)
    xc = xc.contiguous()
    if not (Bt.dtype == C.dtype == xc.dtype == z.dtype) or state.dtype != torch.float32 or not state.is_contiguous():
        raise ApertisHipError("ssm_decode_state: Bt, C, xc, z share a dtype; state is contiguous fp32")
    out = torch.empty(B, Dn, device=xc.device, dtype=xc.dtype)
    check(lib.apertis_ssm_decode_state(ptr(dt_logits.float().contiguous()), ptr(_f32(A_log)), ptr(Bt),
                                       Bt.stride(0), ptr(C), C.stride(0), ptr(xc), ptr(z), z.stride(0),
                                       ptr(_f32(D)), ptr(state), ptr(out), B, h, N, dtype_code(xc),
                                       int(delta_softplus), stream_ptr()), "apertis_ssm_decode_state")
    return out


def ssm_decode_state_dt(dt_in, W_dt, b_dt, A_log, Bt, C, xc, z, D, state, delta_softplus=True):
    """ssm_decode_state(tiny_linear(dt_in, W_dt, b_dt), ...) as one launch (dt_proj_head inside the state kernel; the same
    bits).  dt_in [B, R]: the dt columns of the x_param_proj output, read in place."""
    _require_gpu(dt_in, W_dt, b_dt, A_log, Bt, C, xc, z, D, state)
    lib = _lib.load()
    B, R = dt_in.shape
    h, N = A_log.shape
    Dn = h * N
    fix = lambda t: t if t.stride(-1) == 1 else t.contiguous()
    dt_in, Bt, C, z = fix(dt_in), fix(Bt), fix(C), fix(z)
    xc = xc.contiguous()
    if not (dt_in.dtype == Bt.dtype == C.dtype == xc.dtype == z.dtype) or state.dtype != torch.float32 or not state.is_contiguous():
        raise ApertisHipError("ssm_decode_state_dt: dt_in, Bt, C, xc, z share a dtype; state is contiguous fp32")
    out = torch.empty(B, Dn, device=xc.device, dtype=xc.dtype)
    check(lib.apertis_ssm_decode_state_dt(ptr(dt_in), dt_in.stride(0), ptr(_f32(W_dt)),
                                          ptr(None if b_dt is None else _f32(b_dt)), R,
                                          ptr(_f32(A_log)), ptr(Bt), Bt.stride(0), ptr(C), C.stride(0),
                                          ptr(xc), ptr(z), z.stride(0), ptr(_f32(D)), ptr(state), ptr(out),
                                          B, h, N, dtype_code(xc), int(delta_softplus), stream_ptr()), "apertis_ssm_decode_state_dt")
    return out


def decode_pre_conv(conv_all, conv_w, conv_b):
    """The conv output of a single-token step for ALL layers at once (csrc/decode_step.hip): conv_all [NL,B,Dn,k-1] (the cached
    windows), conv_w [NL,Dn,k], conv_b [NL,Dn] fp32 -> xc [NL,B,Dn].  The values ssm_decode_step returns per layer."""
    _require_gpu(conv_all, conv_w, conv_b)
    lib = _lib.load()
    NL, B, Dn, km1 = conv_all.shape
    xc = torch.empty(NL, B, Dn, device=conv_all.device, dtype=conv_all.dtype)
    check(lib.apertis_decode_pre_conv(ptr(conv_all), ptr(conv_w), ptr(conv_b), ptr(xc), NL, B, Dn, km1 + 1, dtype_code(conv_all),
                                      stream_ptr()), "apertis_decode_pre_conv")
    return xc


def decode_pre_state(p_all, off_bt, off_c, off_dt, W_dt, b_dt, A_log, D, xc_all, state_all, delta_softplus=True):
    """dt_proj_head + state update of ALL layers at once: p_all [NL*B, P] (the x_param_proj outputs: Bt / C / dt columns at the
    given offsets), W_dt [NL,h,R], b_dt [NL,h] or None, A_log [NL,h,N], D [NL,Dn], xc_all [NL,B,Dn]; state_all [NL,B,Dn] fp32 is
    updated in place.  Returns pre [NL,B,Dn] fp32 = C s + D xc (what ssm_decode_state multiplies by silu(z))."""
    _require_gpu(p_all, W_dt, A_log, D, xc_all, state_all)
    lib = _lib.load()
    NL, B, Dn = xc_all.shape
    h, N = A_log.shape[1], A_log.shape[2]
    R = W_dt.shape[2]
    if state_all.dtype != torch.float32 or not state_all.is_contiguous() or not p_all.is_contiguous() or p_all.dtype != xc_all.dtype:
        raise ApertisHipError("decode_pre_state: contiguous fp32 states, p and xc of one dtype")
    pre = torch.empty(NL, B, Dn, device=xc_all.device, dtype=torch.float32)
    check(lib.apertis_decode_pre_state(ptr(p_all), p_all.shape[1], off_bt, off_c, off_dt, ptr(W_dt), ptr(b_dt), R, ptr(A_log), ptr(D),
                                       ptr(xc_all), ptr(state_all), ptr(pre), NL, B, h, N, int(delta_softplus), dtype_code(xc_all),
                                       stream_ptr()), "apertis_decode_pre_state")
    return pre


def decode_post(pre, xz, conv_state):
    """Between in_proj and out_proj of a single-token step whose first half ran ahead (decode_pre_*): gated [B,Dn] =
    pre * silu(z) with xz [B, 2 Dn] = (xp | z), and xp is pushed into conv_state [B,Dn,k-1] IN PLACE."""
    _require_gpu(pre, xz, conv_state)
    lib = _lib.load()
    B, Dn = pre.shape
    if xz.stride(-1) != 1 or conv_state.dtype != xz.dtype or not conv_state.is_contiguous() or not pre.is_contiguous():
        raise ApertisHipError("decode_post: xz rows contiguous, the window contiguous and of xz's dtype")
    gated = torch.empty(B, Dn, device=xz.device, dtype=xz.dtype)
    check(lib.apertis_decode_post(ptr(pre), ptr(xz), xz.stride(0), ptr(conv_state), ptr(gated), B, Dn, conv_state.shape[-1] + 1,
                                  dtype_code(xz), stream_ptr()), "apertis_decode_post")
    return gated


def decode_dense_gemv(x, weight, bias=None):
    """x [B, K] @ weight.T (+ bias) for <= 16 bf16 rows and K < 512 under no_grad: linear_mfma's skinny kernel with the row count
    by value (csrc/decode_step.hip: one dependent round trip less; the same bits).  None when the shapes are not the kernel's."""
    B, K = x.shape
    N = weight.shape[0]
    if not (x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.bfloat16 and 1 <= B <= 16 and K % 8 == 0 and 8 <= K < 512
            and N % 4 == 0 and x.is_contiguous() and weight.shape[1] == K):
        return None
    lib = _lib.load()
    wc = cast_transpose(weight.unsqueeze(0), torch.bfloat16, want_transposed=False, cache=True)[0]
    out = torch.empty(B, N, device=x.device, dtype=torch.bfloat16)
    check(lib.apertis_decode_dense_gemv(ptr(x), ptr(wc), wc.shape[-1], ptr(None if bias is None else _f32(bias)), ptr(out), B, K, N,
                                        stream_ptr()), "apertis_decode_dense_gemv")
    return out


def decode_inproj(w_in, pre, conv_state, xn=None, boundary=None):
    """The in_proj product of a single-token step whose cache-only half ran ahead (bf16, <= 16 rows, 512 <= H <= 1024) with the gate
    and the window push in its epilogue: returns gated [S, Dn] = pre * silu(z) - conv_state [S,Dn,k-1] is pushed in place, xz is
    never written.  Either xn [S, H] is given, or boundary = (blk, res, weight, bias, eps, combine) and the block boundary
    (dropout_add_layer_norm without dropout; combine = (w, plan) or None) runs as the product's prologue in every work-group
    (taken for S <= 4: a row per wave): then (y, gated) comes back.  None when the shapes are not the kernel's."""
    lib = _lib.load()
    N, H = w_in.shape
    S, Dn = pre.shape
    if not (pre.is_cuda and not torch.is_grad_enabled() and 1 <= S <= 16 and 512 <= H <= 1024 and H % 8 == 0 and N == 2 * Dn
            and conv_state.dtype == torch.bfloat16 and conv_state.is_contiguous() and pre.is_contiguous() and pre.dtype == torch.float32
            and 2 <= conv_state.shape[-1] + 1 <= 16):
        return None
    blk2 = res2 = g = b = y = slot = wk = x2 = None
    KK, eps = 0, 0.0
    if xn is not None:
        if xn.dtype != torch.bfloat16 or xn.numel() != S * H:
            return None
        x2 = xn.reshape(S, H).contiguous()
    else:
        blk, res, weight, bias, eps, combine = boundary
        if S > 4 or res.dtype != torch.float32 or blk.dtype != torch.bfloat16 or res.numel() != S * H:
            return None     # (S <= 4 = a row per wave: every work-group normalises the rows for itself - at S = 16 that costs far more than a launch)
        res2 = res.reshape(S, H).contiguous()
        blk2 = blk.reshape(-1, H).contiguous()
        if combine is not None:
            wv, plan = combine
            wk, slot, KK = _f32(wv), plan.slot_of, plan.K
        elif blk2.shape[0] != S:
            return None
        g, b = _f32(weight), _f32(bias)
        y = torch.empty_like(res2)
    wc = cast_transpose(w_in.unsqueeze(0), torch.bfloat16, want_transposed=False, cache=True)[0]       # [1, N, H padded to 64]
    gated = torch.empty(S, Dn, device=pre.device, dtype=torch.bfloat16)
    rc = lib.apertis_decode_inproj(ptr(blk2), ptr(slot), ptr(wk), KK, ptr(res2), ptr(g), ptr(b), float(eps), ptr(y), ptr(x2), ptr(wc),
                                   wc.shape[-1], None, ptr(pre), ptr(conv_state), conv_state.shape[-1] + 1, ptr(gated), S, H, N, Dn,
                                   stream_ptr())
    if rc == -2:
        return None
    check(rc, "apertis_decode_inproj")
    return gated if xn is not None else (y.reshape(boundary[1].shape), gated)
