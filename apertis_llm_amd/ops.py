"""Host-side operators: torch.autograd.Functions over the C ABI (libapertis_hip.so).

torch is used for device memory, streams and autograd bookkeeping only; every computation
below is a HIP kernel launch through apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import torch

from . import _lib
from ._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr


def _require_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ApertisHipError("apertis_llm_amd ops run on a ROCm device only (tensor on %s); "
                                  "there is no CPU fallback" % t.device)


def _row_stride(t, width):
    """Row stride (elements) of a [..., rows, width] view whose rows are `width` contiguous
    elements apart by a constant stride and whose leading dims are packed on top of it."""
    if t.stride(-1) != 1:
        raise ApertisHipError("innermost dimension must be contiguous")
    rs = t.stride(-2)
    if t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * rs:
        raise ApertisHipError("batch stride must equal L * row_stride")
    return rs


# ----------------------------------------------------------------------------------------------
# selective scan
# ----------------------------------------------------------------------------------------------
class _SelectiveScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dlt, A_log, Bt, C, h0, delta_softplus, y_dtype, return_last):
        _require_gpu(dlt, A_log, Bt, C, h0)
        lib = _lib.load()
        B, L, h = dlt.shape
        N = A_log.shape[1]
        Dn = h * N
        if Bt.shape != (B, L, Dn) or C.shape != (B, L, Dn) or A_log.shape[0] != h:
            raise ApertisHipError(f"scan shapes: dlt {tuple(dlt.shape)} A_log {tuple(A_log.shape)} "
                                  f"Bt {tuple(Bt.shape)} C {tuple(C.shape)}")
        if Bt.dtype != C.dtype:
            raise ApertisHipError("Bt and C must share a dtype")
        dlt = dlt.float().contiguous()
        A_log = A_log.float().contiguous()
        if h0 is not None:
            h0 = h0.float().reshape(B, Dn).contiguous()
        bt_rs, c_rs = _row_stride(Bt, Dn), _row_stride(C, Dn)
        nch = lib.apertis_scan_num_chunks(B, L, Dn)
        dev = dlt.device
        y = torch.empty(B, L, Dn, device=dev, dtype=y_dtype)
        agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
        h_in = torch.empty(B, nch, Dn, device=dev, dtype=torch.float32)
        h_last = torch.empty(B, Dn, device=dev, dtype=torch.float32) if return_last else None
        check(lib.apertis_selective_scan_fwd(ptr(dlt), ptr(A_log), ptr(Bt), bt_rs, ptr(C), c_rs, ptr(h0), ptr(y), Dn,
                                             ptr(h_last), ptr(agg), ptr(h_in), B, L, h, N, dtype_code(Bt),
                                             dtype_code(y), int(delta_softplus), stream_ptr()),
              "apertis_selective_scan_fwd")
        ctx.save_for_backward(dlt, A_log, Bt, C, h_in)
        ctx.cfg = (B, L, h, N, bool(delta_softplus))
        ctx.mark_non_differentiable(*([h_last] if return_last else []))
        return (y, h_last) if return_last else y

    @staticmethod
    def backward(ctx, dy, *_unused):
        lib = _lib.load()
        dlt, A_log, Bt, C, h_in = ctx.saved_tensors
        B, L, h, N, sp = ctx.cfg
        Dn = h * N
        dy = dy.contiguous()
        dev = dlt.device
        nch = h_in.shape[1]
        # dBt/dC keep the layout of the forward views when those are slices of one projection
        # output, so autograd's slice-backward sees dense tensors of the expected shape
        dBt = torch.empty(B, L, Dn, device=dev, dtype=Bt.dtype)
        dC = torch.empty(B, L, Dn, device=dev, dtype=C.dtype)
        d_dlt = torch.empty(B, L, h, device=dev, dtype=torch.float32)
        dA_log = torch.empty(h, N, device=dev, dtype=torch.float32)
        agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
        mu_in = torch.empty(B, nch, Dn, device=dev, dtype=torch.float32)
        dA_part = torch.empty(B * nch, Dn, device=dev, dtype=torch.float32)
        check(lib.apertis_selective_scan_bwd(ptr(dlt), ptr(A_log), ptr(Bt), _row_stride(Bt, Dn), ptr(C),
                                             _row_stride(C, Dn), ptr(dy), Dn, ptr(h_in), ptr(dBt), Dn, ptr(dC), Dn,
                                             ptr(d_dlt), ptr(dA_log), ptr(agg), ptr(mu_in), ptr(dA_part),
                                             B, L, h, N, dtype_code(Bt), dtype_code(dy), int(sp), stream_ptr()),
              "apertis_selective_scan_bwd")
        return d_dlt, dA_log, dBt, dC, None, None, None, None


def selective_scan(dlt, A_log, Bt, C, h0=None, delta_softplus=False, y_dtype=torch.float32, return_last=False):
    """y[b,t,c] = C*s,  s_t = exp(delta_t*A)*s_{t-1} + Bt_t  (reference core.py:337-353).

    dlt [B,L,h] fp32 (delta, or its pre-softplus logits when delta_softplus), A_log [h,N],
    Bt/C [B,L,h*N] fp32 or bf16 (strided column slices are taken as they are), h0 [B,h*N] or
    None.  Returns y [B,L,h*N] (and the final state [B,h*N] when return_last)."""
    return _SelectiveScan.apply(dlt, A_log, Bt, C, h0, delta_softplus, y_dtype, return_last)
