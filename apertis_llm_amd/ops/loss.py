"""Loss operators: shifted cross-entropy and the fused LM-head + cross-entropy (no [B, L, V] logits tensor).

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import os as _os

import torch

from .. import _lib
from .._lib import check, dtype_code, ptr, stream_ptr
from ._base import _apply, _grad_wanted, _require_gpu
from . import gemm as _gemm


class _ShiftedCrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, n_pos, ignore_index):
        _require_gpu(logits, labels)
        lib = _lib.load()
        B, L, V = logits.shape
        logits = logits.contiguous()
        labels = labels.contiguous()
        lse = torch.empty(B * L, device=logits.device, dtype=torch.float32)
        row_loss = torch.empty(B * L, device=logits.device, dtype=torch.float32)
        check(lib.apertis_cross_entropy_fwd(ptr(logits), ptr(labels), ptr(lse), ptr(row_loss), B, L, V, labels.shape[1], n_pos,
                                            ignore_index, dtype_code(logits), stream_ptr()), "apertis_cross_entropy_fwd")
        # the same predicate as the kernel (a label outside [0, V) is skipped there): 0 targets -> nan, like
        # F.cross_entropy; out-of-range labels are rejected by shifted_cross_entropy() unless it was told not to look
        tgt = labels[:, 1:n_pos + 1]
        count = ((tgt != ignore_index) & (tgt >= 0) & (tgt < V)).sum().to(torch.float32)
        ctx.save_for_backward(logits, labels, lse, count)
        ctx.cfg = (n_pos, ignore_index)
        return row_loss.sum() / count

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        logits, labels, lse, count = ctx.saved_tensors
        n_pos, ignore_index = ctx.cfg
        B, L, V = logits.shape
        gscale = (dloss.to(torch.float32) / count).reshape(1).contiguous()
        dlogits = torch.empty_like(logits)
        check(lib.apertis_cross_entropy_bwd(ptr(logits), ptr(labels), ptr(lse), ptr(gscale), ptr(dlogits), B, L, V,
                                            labels.shape[1], n_pos, ignore_index, dtype_code(logits), stream_ptr()),
              "apertis_cross_entropy_bwd")
        return dlogits, None, None, None


def shifted_cross_entropy_supported(logits, labels):
    V = logits.shape[-1]
    return (logits.is_cuda and labels.is_cuda and logits.dim() == 3 and labels.dim() == 2 and labels.dtype == torch.int64 and
            logits.shape[0] == labels.shape[0] and logits.dtype in (torch.float32, torch.bfloat16) and
            V % (8 if logits.dtype == torch.bfloat16 else 4) == 0 and logits.shape[0] * logits.shape[1] < 2 ** 31)


def shifted_cross_entropy(logits, labels, ignore_index=-100):
    """mean_{valid (b,l)} CE(logits[b, l, :], labels[b, l+1]) for l < min(L, L_labels) - 1, fp32 math on the
    logits as stored: the reference's shift + CrossEntropyLoss(ignore_index) (core.py:1407-1416) without the
    shifted / fp32 copies of the [B, L, V] tensor.  Labels must be in [0, V) or ignore_index."""
    n_pos = min(logits.shape[1], labels.shape[1]) - 1
    return _ShiftedCrossEntropy.apply(logits, labels, n_pos, ignore_index)


# rows of logits held at a time by linear_cross_entropy (16384 x 32000 bf16 = 1 GiB): with one 4096-token sequence per
# chunk the 32 weight-gradient partial GEMMs and their fp32 accumulation cost 4 ms of the 440 ms step, with four per chunk
# the step time equals the logits path's and the peak is still 14 GiB lower at batch 32
_LCE_CHUNK_ROWS = int(_os.environ.get("APERTIS_LCE_CHUNK_ROWS", "16384"))
# the chunk's weight gradient dl.T @ x on the library's own wide-tile TN kernel (fp32 partials straight into the fp32 sum)
# instead of a stock bf16 GEMM + add
LCE_OWN_WGRAD = _os.environ.get("APERTIS_LCE_OWN_WGRAD", "1") == "1"
LCE_FUSED_CE = _os.environ.get("APERTIS_LCE_FUSED_CE", "1") == "1"     # apertis_cross_entropy_fwd_bwd on the chunk (tests switch it off)
# (the chunk's logits on the library's NT kernel instead: 761 against 760 us alone, the step within noise - stays on hipBLASLt;
# d hidden = dl @ W, K = 32000: 1 019 against 802 us - stays too.  tools/prof_lm_head.py)


class _LinearCrossEntropy(torch.autograd.Function):
    """loss = shifted CE(hidden @ W.T, labels) without the [B, L, V] logits tensor: the LM head and the loss are walked
    a few sequences at a time - logits of the chunk (hipBLASLt GEMM), apertis_cross_entropy_fwd (log-sum-exp + loss),
    apertis_cross_entropy_bwd IN PLACE on the chunk (softmax - onehot, already scaled by 1 / #targets), and the chunk's
    two gradient GEMMs (d hidden, and d W accumulated in fp32) - so only one chunk of logits ever exists and nothing is
    recomputed in the backward, which just scales the stored gradients by the incoming scalar."""

    @staticmethod
    def forward(ctx, hidden, weight, labels, ignore_index, compute_dtype):
        _require_gpu(hidden, weight, labels)
        lib = _lib.load()
        B, L, H = hidden.shape
        V = weight.shape[0]
        labels = labels.contiguous()
        n_pos = min(L, labels.shape[1]) - 1
        need = _grad_wanted(ctx, 2)
        x = hidden.to(compute_dtype).contiguous()
        w = weight.detach().to(compute_dtype)
        tgt = labels[:, 1:n_pos + 1]
        count = ((tgt != ignore_index) & (tgt >= 0) & (tgt < V)).sum().to(torch.float32)
        gscale = (1.0 / count).reshape(1).contiguous()
        dev = hidden.device
        code = dtype_code(x)
        loss_sum = torch.zeros((), device=dev, dtype=torch.float32)
        dx = torch.empty_like(x) if need else None
        dw = part = None
        nb = max(1, _LCE_CHUNK_ROWS // L)                                     # sequences per chunk
        lse = torch.empty(nb * L, device=dev, dtype=torch.float32)
        row_loss = torch.empty(nb * L, device=dev, dtype=torch.float32)
        own_wgrad = (need and LCE_OWN_WGRAD and compute_dtype == torch.bfloat16 and V % 8 == 0 and H % 8 == 0
                     and lib.apertis_grouped_gemm_tn_dense_variant(V, H) >= 0)
        if need:   # the first chunk's weight gradient is written, the later ones are added
            dw = torch.empty(V, H, device=dev, dtype=torch.float32)
            part = torch.empty(V, H, device=dev, dtype=torch.float32) if own_wgrad and B > nb else None
        for b0 in range(0, B, nb):
            n = min(nb, B - b0)
            xb = x[b0:b0 + n].reshape(n * L, H)
            logits = torch.matmul(xb, w.t()).reshape(n, L, V)                 # n sequences of logits
            lab = labels[b0:b0 + n]
            # (with gradients wanted: log-sum-exp, loss and softmax - onehot of a row in ONE pass over it - the row waits in its
            #  work-group's registers - when the vocabulary fits; else, and without gradients, the two kernels)
            fused = need and LCE_FUSED_CE
            if fused:
                rc = lib.apertis_cross_entropy_fwd_bwd(ptr(logits), ptr(lab), ptr(lse), ptr(row_loss), ptr(gscale), ptr(logits), n,
                                                       L, V, labels.shape[1], n_pos, ignore_index, code, stream_ptr())
                if rc == -2:
                    fused = False
                else:
                    check(rc, "apertis_cross_entropy_fwd_bwd")
            if not fused:
                check(lib.apertis_cross_entropy_fwd(ptr(logits), ptr(lab), ptr(lse), ptr(row_loss), n, L, V, labels.shape[1], n_pos,
                                                    ignore_index, code, stream_ptr()), "apertis_cross_entropy_fwd")
            loss_sum += row_loss[:n * L].sum()
            if need:
                if not fused:
                    check(lib.apertis_cross_entropy_bwd(ptr(logits), ptr(lab), ptr(lse), ptr(gscale), ptr(logits), n, L, V,
                                                        labels.shape[1], n_pos, ignore_index, code, stream_ptr()),
                          "apertis_cross_entropy_bwd")
                dl = logits.reshape(n * L, V)
                torch.matmul(dl, w, out=dx[b0:b0 + n].reshape(n * L, H))
                if own_wgrad:
                    ws, ws_bytes = _gemm._tn_workspace(1, 1, dev, n * L)
                    check(lib.apertis_grouped_gemm_tn(ptr(dl), ptr(xb), ptr(_gemm._dense_offsets(n * L, dev)),
                                                      ptr(part if b0 else dw), None, n * L, V, H, 1, ptr(ws), ws_bytes, code,
                                                      stream_ptr()), "apertis_grouped_gemm_tn")
                    if b0:
                        dw.add_(part)
                elif b0:
                    dw.add_(torch.matmul(dl.t(), xb))
                else:
                    dw.copy_(torch.matmul(dl.t(), xb))
        ctx.save_for_backward(dx, dw)
        ctx.cfg = (hidden.dtype, weight.dtype)
        return loss_sum / count

    @staticmethod
    def backward(ctx, dloss):
        dx, dw = ctx.saved_tensors
        hdt, wdt = ctx.cfg
        g = dloss.to(torch.float32)
        return ((dx * g.to(dx.dtype)).to(hdt) if ctx.needs_input_grad[0] else None,
                (dw * g).to(wdt) if ctx.needs_input_grad[1] else None, None, None, None)


def linear_cross_entropy_supported(hidden, weight, labels):
    V = weight.shape[0]
    cd = torch.bfloat16 if torch.is_autocast_enabled() else hidden.dtype
    return (hidden.is_cuda and weight.is_cuda and labels.is_cuda and hidden.dim() == 3 and labels.dim() == 2 and
            labels.dtype == torch.int64 and hidden.shape[0] == labels.shape[0] and cd in (torch.float32, torch.bfloat16) and
            V % (8 if cd == torch.bfloat16 else 4) == 0 and hidden.shape[1] >= 2)


def linear_cross_entropy(hidden, weight, labels, ignore_index=-100, compute_dtype=None):
    """mean over valid (b, l) of CE((hidden @ weight.T)[b, l], labels[b, l + 1]): the LM head (reference core.py:1412) and
    the shifted cross entropy (core.py:1417-1450) as one op that never holds more than one sequence of logits.
    hidden [B, L, H], weight [V, H] (the tied embedding), labels [B, >= L] int64."""
    return _apply(_LinearCrossEntropy, hidden, weight, labels, ignore_index, compute_dtype or hidden.dtype)
