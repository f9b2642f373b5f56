"""Grouped / dense GEMM operators over apertis_grouped_gemm_nt / _tn (expert and dense linears, skinny and tiny linears, split-K offsets, tile queues).

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import os as _os

import torch

from .. import _lib
from .._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr
from ._base import _apply, _f32, _grad_wanted, _launch, _require_gpu, _rows, _slot_of
from .prep import cast_transpose


class _SkinnyLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, weight, bias)
        lib = _lib.load()
        x = x.contiguous()
        T, K = x.shape
        N = weight.shape[0]
        w = _f32(weight)
        b = None if bias is None else _f32(bias)
        y = torch.empty(T, N, device=x.device, dtype=torch.float32)
        check(lib.apertis_skinny_linear_fwd(ptr(x), ptr(w), ptr(b), ptr(y), T, K, N, dtype_code(x), stream_ptr()),
              "apertis_skinny_linear_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        T, K = x.shape
        N = w.shape[0]
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        nblk = lib.apertis_skinny_linear_bwd_blocks(T)
        part = torch.empty(nblk, N * K + N, device=x.device, dtype=torch.float32)
        out = torch.empty(N * K + N, device=x.device, dtype=torch.float32)
        check(lib.apertis_skinny_linear_bwd(ptr(x), ptr(w), ptr(dy), ptr(dx), ptr(part), ptr(out), T, K, N, dtype_code(x),
                                            stream_ptr()), "apertis_skinny_linear_bwd")
        wdt, bdt = ctx.cfg
        return dx, out[:N * K].reshape(N, K).to(wdt), (out[N * K:].to(bdt) if bdt is not None else None)


class _TinyLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_gpu(x, weight, bias)
        lib = _lib.load()
        K, N = x.shape[-1], weight.shape[0]
        ctx.slot = _slot_of(x)
        x3 = x if x.dim() == 3 else x.reshape(1, -1, K)
        xr, ldx = _rows(x3, K)
        T = x.numel() // K
        w = _f32(weight)
        b = None if bias is None else _f32(bias)
        y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        check(lib.apertis_tiny_linear_fwd(ptr(xr), ldx, ptr(w), ptr(b), ptr(y), T, K, N, dtype_code(xr), stream_ptr()),
              "apertis_tiny_linear_fwd")
        ctx.save_for_backward(xr, w)
        ctx.cfg = (ldx, T, tuple(x.shape), weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        xr, w = ctx.saved_tensors
        ldx, T, xshape, wdt, bdt = ctx.cfg
        N, K = w.shape
        dy = dy.float().contiguous()
        if ctx.slot is not None and ctx.slot[0].widths[ctx.slot[1]] == K:
            dx, Kp = ctx.slot[0].out(ctx.slot[1], xshape[:-1], xr.dtype, xr.device)   # a column range of the shared buffer
            dxp = dx
        else:
            Kp = -(-K // 8) * 8     # 16-byte row pitch: the kernel stores whole 16-byte chunks
            dxp = torch.empty(*xshape[:-1], Kp, device=xr.device, dtype=xr.dtype)
            dx = dxp[..., :K]
        nblk = lib.apertis_tiny_linear_bwd_blocks(T)
        part = torch.empty(nblk, N * K + N, device=xr.device, dtype=torch.float32)
        out = torch.empty(N * K + N, device=xr.device, dtype=torch.float32)
        check(lib.apertis_tiny_linear_bwd(ptr(xr), ldx, ptr(w), ptr(dy), ptr(dxp), Kp, ptr(part), ptr(out), T, K, N,
                                          dtype_code(xr), stream_ptr()), "apertis_tiny_linear_bwd")
        return dx, out[:N * K].reshape(N, K).to(wdt), (out[N * K:].to(bdt) if bdt is not None else None)


def tiny_linear_supported(x, K, N):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and 1 <= K <= 64 and 1 <= N <= 16


def tiny_linear(x, weight, bias=None):
    """fp32 y = x @ W.T + b for K <= 64 inputs and N <= 16 outputs, x read in place when it is a column
    slice (the SSM's dt_proj_head on p[..., :dt_rank], reference core.py:382)."""
    return _TinyLinear.apply(x, weight, bias)


def skinny_linear_supported(K, N):
    return N in (2, 4, 8, 16) and K % 4 == 0 and K <= 1024 and (N <= 8 or K <= 256)


def skinny_linear(x, weight, bias=None):
    """fp32 y = x @ W.T + b for a handful of output columns (the MoE router, reference core.py:482):
    bandwidth-bound row kernel instead of a GEMM-library call."""
    return _SkinnyLinear.apply(x, weight, bias)


_SPLITK_ROWS = 1024


_splitk_cache = {}


# APERTIS_DENSE_WGRAD_WIDE=0: every dense weight gradient on the 128 x 128 kernel over pseudo-groups (the round-1..3 form)
DENSE_WGRAD_WIDE = _os.environ.get("APERTIS_DENSE_WGRAD_WIDE", "1") == "1"


def _splitk_depth(N, K):
    """Rows per pseudo-group of a dense weight gradient: deeper groups halve the partial sums (written and folded: as many
    bytes as the operands at 1024 rows) but leave fewer work-groups; measured at T = 163840 (tools/prof_dense_wgrad.py, us at
    1024 / 2048 / 4096 rows): dW [352, 704] 153 / 138 / 157, [704, 176] 105 / 93 / 90, [448, 176] 73 / 86 / 80."""
    tiles = -(-N // 128) * -(-K // 128)
    return _SPLITK_ROWS if tiles <= 8 else 2 * _SPLITK_ROWS


def _splitk_offsets(rows, G, depth, device):
    key = (rows, G, depth, str(device))
    t = _splitk_cache.get(key)
    if t is None:
        t = torch.tensor([min(i * depth, rows) for i in range(G + 1)], dtype=torch.int32, device=device)
        _splitk_cache[key] = t
    return t


_dense_offsets_cache = {}


def _dense_offsets(rows, device):
    key = (rows, str(device))
    t = _dense_offsets_cache.get(key)
    if t is None:
        t = torch.tensor([0, rows], dtype=torch.int32, device=device)
        _dense_offsets_cache[key] = t
    return t


class _RowsWork:
    """flops of a grouped GEMM = (rows actually routed, read from the device after the run) x
    flops per row."""
    __slots__ = ("offsets", "E", "per_row")

    def __init__(self, offsets, E, per_row):
        self.offsets, self.E, self.per_row = offsets, E, per_row

    def __call__(self):
        return float(self.offsets[self.E].item()) * self.per_row


# Set by parallel.BucketedDataParallel when it wraps a model for world_size > 1: the persistent NT GEMM and the 256x256
# weight-gradient kernel then take their tiles from per-launch counters (apertis_grouped_gemm_nt_q / _tn_q / _tn_pair_q), so
# that a work-group whose CU an RCCL kernel holds does not walk a full static share alone at the end (measured with
# tools/probes/hog_probe.hip: 32 of 256 CUs held -> NT 1109 us static / 762 queue / 702 alone, TN 2400 / 1900 / 1460).
# Off on one GPU: the queues cost 3 % there.
GEMM_DYNAMIC_QUEUE = False


# ... and whether the WEIGHT-GRADIENT kernels follow it.  Off (round 4): with the 352-wide tiles a (problem, expert) group has
# 22 tiles on 16 CUs, and a queue of such coarse items quantises badly when CUs are missing - measured at B = 44 with 32 of 256
# CUs held (tools/probes/hog_probe.hip, profiles/r4_probe_cu_hog_32.log): static shares 2130-2210 us, item queue 2620-2750 us
# (1700-1790 / 1680-1730 alone; 224 CUs' worth of work would be 1940).  The work-groups that start late on a freed CU run
# their static share on an otherwise idle chip, which costs less than the queue's last round.  The NT tile queue keeps its gain
# (903 against 1254 us under the same hog; 867 against 780 alone).
TN_DYNAMIC_QUEUE = _os.environ.get("APERTIS_TN_QUEUE", "0") == "1"


_NT_QUEUE = {}


def _nt_queue(device):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    q = _NT_QUEUE.get(key)
    if q is None:
        q = _NT_QUEUE[key] = torch.zeros(512, device=device, dtype=torch.int32)     # APERTIS_NT_QUEUE_INTS: one counter per XCD
    return q


def _launch_nt(name, lib, args, work, device, detail=None, nbytes=0.0):
    """apertis_grouped_gemm_nt, or its _q form with the stream's tile-queue counter when GEMM_DYNAMIC_QUEUE is on.
    `args` ends with the stream pointer."""
    if GEMM_DYNAMIC_QUEUE:
        _launch(name, lib.apertis_grouped_gemm_nt_q, args[:-1] + (ptr(_nt_queue(device)), args[-1]), work, detail, nbytes)
    else:
        _launch(name, lib.apertis_grouped_gemm_nt, args, work, detail, nbytes)


_ACTS = {None: _lib.ACT_NONE, "none": _lib.ACT_NONE, "gelu": _lib.ACT_GELU, "relu": _lib.ACT_RELU,
         "silu": _lib.ACT_SILU, "swish": _lib.ACT_SILU}


_TN_WS = {}


def _tn_workspace(E, n_problems, device, max_rows=None):
    """Scratch for the split tiles of the weight-gradient GEMM: one buffer per (device, stream),
    sized by the library, contents don't care (apertis_hip.h: TN workspace).  Passing it selects the 256 x 256-tile
    kernel; short groups (under 2048 rows each on average: a 256-row-deep slice per CU does not amortise the tile
    prologue / epilogue and the fold) get none and run on the 128 x 128 kernel."""
    if max_rows is not None and max_rows // max(E, 1) < 2048:
        return None, 0
    nbytes = _lib.load().apertis_grouped_gemm_tn_workspace_bytes(E, n_problems)
    if nbytes <= 0:
        return None, 0
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _TN_WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _TN_WS[key] = torch.empty(nbytes, device=device, dtype=torch.uint8)
    return buf, nbytes


def _dense_tag(E, rows, N, K, esize):
    """(shape tag, algorithmic bytes) of a one-group NT call for the kernel timer: X read once, Y written once, W once."""
    if E != 1:
        return None, 0.0
    return f"rows={rows} N={N} K={K}", float(rows) * (N + K) * esize + float(N) * K * esize


class _GroupedLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, offsets, max_rows, act, drop_p, seed, compute_dtype):
        _require_gpu(x, weight, bias, offsets)
        lib = _lib.load()
        E, N, K = weight.shape
        if x.dtype != compute_dtype:
            x = x.to(compute_dtype)
        x = x.contiguous()
        if x.shape[1] != K:
            raise ApertisHipError(f"grouped_linear: x {tuple(x.shape)} vs weight {tuple(weight.shape)}")
        need_grad = _grad_wanted(ctx, 3)
        if compute_dtype == torch.float32 and weight.dtype == torch.float32 and weight.is_contiguous():
            wc = weight.detach()
            wt = cast_transpose(weight, compute_dtype, want_plain=False)[1] if need_grad else None
        else:
            wc, wt = cast_transpose(weight, compute_dtype, want_transposed=need_grad, cache=not need_grad)
        bf = None if bias is None else _f32(bias)
        code = dtype_code(x)
        act_code = _ACTS[act]
        out = torch.empty(x.shape[0], N, device=x.device, dtype=compute_dtype)
        pre = torch.empty_like(out) if (act_code != _lib.ACT_NONE and need_grad) else None
        _launch_nt("apertis_grouped_gemm_nt" if E > 1 else "apertis_grouped_gemm_nt[dense]", lib,
                (ptr(x), ptr(wc), ptr(bf), ptr(offsets), ptr(out), ptr(pre), None, max_rows, N, K, wc.shape[-1], E, act_code,
                 float(drop_p), int(seed), code, code, stream_ptr()), _RowsWork(offsets, E, 2.0 * N * K), x.device,
                *_dense_tag(E, max_rows, N, K, x.element_size()))
        ctx.save_for_backward(x, wt, pre, offsets)
        ctx.cfg = (E, N, K, max_rows, act_code, float(drop_p), int(seed), bias is not None, weight.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, wt, pre, offsets = ctx.saved_tensors
        E, N, K, max_rows, act_code, drop_p, seed, has_bias, wdtype = ctx.cfg
        code = dtype_code(x)
        dout = dout.to(x.dtype).contiguous()
        if act_code != _lib.ACT_NONE:
            dpre = torch.empty_like(dout)
            check(lib.apertis_act_dropout_bwd(ptr(dout), ptr(pre), ptr(dpre), ptr(offsets), max_rows, N, E, act_code,
                                              drop_p, seed, code, stream_ptr()), "apertis_act_dropout_bwd")
        else:
            dpre = dout
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _launch_nt("apertis_grouped_gemm_nt" if E > 1 else "apertis_grouped_gemm_nt[dense]", lib,
                    (ptr(dpre), ptr(wt), None, ptr(offsets), ptr(dx), None, None, max_rows, K, N, wt.shape[-1], E, _lib.ACT_NONE,
                     0.0, 0, code, code, stream_ptr()), _RowsWork(offsets, E, 2.0 * N * K), dpre.device,
                    *_dense_tag(E, max_rows, K, N, x.element_size()))
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            if (E == 1 and max_rows >= 4 * _SPLITK_ROWS and DENSE_WGRAD_WIDE and x.dtype == torch.bfloat16 and not has_bias
                    and lib.apertis_grouped_gemm_tn_dense_variant(N, K) >= 0):
                # a dense layer with enough output (from ~240 000 elements: dW [352, 704] 191 -> 166 us, [704, 2816] 1105 ->
                # 637 us) on the wide-tile kernel: the library splits the rows of every 352-wide tile over the CUs and folds
                # the slices in order; narrower projections (dW [704, 176]: 133 vs 141 us) stay on the pseudo-groups below
                dw = torch.empty(1, N, K, device=x.device, dtype=torch.float32)
                ws, ws_bytes = _tn_workspace(1, 1, x.device, max_rows)
                _launch("apertis_grouped_gemm_tn[dense]", lib.apertis_grouped_gemm_tn,
                        (ptr(dpre), ptr(x), ptr(offsets), ptr(dw), None, max_rows, N, K, 1, ptr(ws), ws_bytes, code, stream_ptr()),
                        2.0 * max_rows * N * K, f"rows={max_rows} M={N} N={K}", float(max_rows) * (N + K) * x.element_size())
                db = None
            elif E == 1 and max_rows >= 4 * _SPLITK_ROWS:
                # dense layer: the K dimension of the weight gradient is ALL rows; cut it into
                # pseudo-groups of _SPLITK_ROWS rows so the grid fills the chip, then fold the
                # partials in a fixed order (deterministic split-K, no atomics).  The dense layers of
                # this model are narrow (352 / 704 wide): 128x128 tiles waste 8 % of the MFMA work on
                # them where the 256x256 split-K kernel wastes 37 % (measured 322 vs 144 TF)
                depth = _splitk_depth(N, K)
                G = -(-max_rows // depth)
                soffs = _splitk_offsets(max_rows, G, depth, x.device)
                part = torch.empty(G, N, K, device=x.device, dtype=torch.float32)
                bpart = torch.empty(G, N, device=x.device, dtype=torch.float32) if has_bias else None
                _launch("apertis_grouped_gemm_tn[dense]", lib.apertis_grouped_gemm_tn,
                        (ptr(dpre), ptr(x), ptr(soffs), ptr(part), ptr(bpart), max_rows, N, K, G, None, 0, code, stream_ptr()),
                        2.0 * max_rows * N * K, f"rows={max_rows} M={N} N={K}",
                        float(max_rows) * (N + K) * x.element_size() + 4.0 * G * N * K)
                dw = torch.empty(1, N, K, device=x.device, dtype=torch.float32)
                check(lib.apertis_colsum_f32(ptr(part), ptr(dw), G, N * K, stream_ptr()), "apertis_colsum_f32")
                db = None
                if has_bias:
                    db = torch.empty(1, N, device=x.device, dtype=torch.float32)
                    check(lib.apertis_colsum_f32(ptr(bpart), ptr(db), G, N, stream_ptr()), "apertis_colsum_f32")
            else:
                dw = torch.empty(E, N, K, device=x.device, dtype=torch.float32)
                db = torch.empty(E, N, device=x.device, dtype=torch.float32) if has_bias else None
                ws, ws_bytes = _tn_workspace(E, 1, x.device, max_rows)
                _launch("apertis_grouped_gemm_tn" if E > 1 else "apertis_grouped_gemm_tn[dense]", lib.apertis_grouped_gemm_tn_q,
                        (ptr(dpre), ptr(x), ptr(offsets), ptr(dw), ptr(db), max_rows, N, K, E, ptr(ws), ws_bytes, code,
                         int(GEMM_DYNAMIC_QUEUE and TN_DYNAMIC_QUEUE), stream_ptr()), _RowsWork(offsets, E, 2.0 * N * K))
            dw = dw.to(wdtype)
        return dx, dw, db, None, None, None, None, None, None


def grouped_linear(x, weight, bias, offsets, max_rows, act=None, drop_p=0.0, seed=0, compute_dtype=None):
    """Per-group  act(x @ W[e].T + b[e])  with optional fused inverted dropout, on MFMA.
    x [R,K] rows sorted by group, weight [E,N,K] (nn.Linear layout, fp32 master), bias [E,N],
    offsets [E+1] int32 device tensor.  Rows >= offsets[E] are neither read nor written.
    (reference: expert Linear/activation/Dropout/Linear, core.py:437-440)"""
    return _apply(_GroupedLinear, x, weight, bias, offsets, max_rows, act, drop_p, seed, compute_dtype or x.dtype)


class _ScatterRows(torch.autograd.Function):
    """out[dst_idx[i]] = w[i], every other row zero; the backward gathers the same rows back.  (As slices + zeros + cat
    the backward was eight fill / copy / add kernels per call.)"""
    @staticmethod
    def forward(ctx, w, dst_idx, rows_out):
        out = w.new_zeros(rows_out, *w.shape[1:])
        out.index_copy_(0, dst_idx, w.detach())
        ctx.save_for_backward(dst_idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (dst_idx,) = ctx.saved_tensors
        return g.index_select(0, dst_idx), None, None


def scatter_rows(w, dst_idx, rows_out):
    """Rows of `w` placed at `dst_idx` (int64 device tensor, a permutation into `rows_out` >= len(w) rows), the rest zero."""
    return _ScatterRows.apply(w, dst_idx, rows_out)


def dense_offsets(rows, device):
    """Group offsets [0, rows] of a one-group (dense) call of the grouped kernels, cached per (rows, device)."""
    return _dense_offsets(rows, device)


def linear_mfma(x, weight, bias=None, act=None, compute_dtype=None):
    """Dense act(x @ W.T + b) through the same MFMA tile (one group).  Used for the patch-embed
    GEMM and vision_projection (reference multimodal/module.py:102, core.py:1209)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    M = x2.shape[0]
    offsets = _dense_offsets(M, x.device)
    out = grouped_linear(x2, weight.unsqueeze(0), None if bias is None else bias.unsqueeze(0), offsets, M, act, 0.0, 0,
                         compute_dtype or x.dtype)
    return out.reshape(*lead, weight.shape[0])
