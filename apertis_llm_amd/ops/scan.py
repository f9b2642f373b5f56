"""Selective scan operators: the plain scan and the scan + skip + gate fused forms (staged / lean / look-back), their workspaces and error word.

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import os as _os

import torch

from .. import _lib
from .._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr
from ._base import _apply, _f32, _grad_out, _grad_wanted, _indexed, _launch, _require_gpu, _rows, _slot_of, _try_launch


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
        ctx.slots = (_slot_of(Bt), _slot_of(C))
        (Bt, bt_rs), (C, c_rs) = _rows(Bt, Dn), _rows(C, Dn)
        nch = lib.apertis_scan_num_chunks(B, L, Dn)
        dev = dlt.device
        y = torch.empty(B, L, Dn, device=dev, dtype=y_dtype)
        agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
        h_in = torch.empty(B, nch, Dn, device=dev, dtype=torch.float32)
        h_last = torch.empty(B, Dn, device=dev, dtype=torch.float32) if return_last else None
        work = B * L * (Dn * (2 * Bt.element_size() + y.element_size()) + 4 * h) + 4 * h * N   # algorithmic bytes
        _launch("apertis_selective_scan_fwd", lib.apertis_selective_scan_fwd,
                (ptr(dlt), ptr(A_log), ptr(Bt), bt_rs, ptr(C), c_rs, ptr(h0), ptr(y), Dn, ptr(h_last), ptr(agg),
                 ptr(h_in), B, L, h, N, dtype_code(Bt), dtype_code(y), int(delta_softplus), stream_ptr()), work)
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
        dBt, dbt_rs = _grad_out(ctx.slots[0], (B, L), Dn, Bt.dtype, dev)
        dC, dc_rs = _grad_out(ctx.slots[1], (B, L), Dn, C.dtype, dev)
        d_dlt = torch.empty(B, L, h, device=dev, dtype=torch.float32)
        dA_log = torch.empty(h, N, device=dev, dtype=torch.float32)
        agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
        mu_in = torch.empty(B, nch, Dn, device=dev, dtype=torch.float32)
        dA_part = torch.empty(B * nch, Dn, device=dev, dtype=torch.float32)
        work = B * L * (Dn * (4 * Bt.element_size() + dy.element_size()) + 8 * h) + 8 * h * N  # algorithmic bytes
        _launch("apertis_selective_scan_bwd", lib.apertis_selective_scan_bwd,
                (ptr(dlt), ptr(A_log), ptr(Bt), Bt.stride(-2), ptr(C), C.stride(-2), ptr(dy), Dn, ptr(h_in), ptr(dBt), dbt_rs,
                 ptr(dC), dc_rs, ptr(d_dlt), ptr(dA_log), ptr(agg), ptr(mu_in), ptr(dA_part), B, L, h, N, dtype_code(Bt),
                 dtype_code(dy), int(sp), stream_ptr()), work)
        return d_dlt, dA_log, dBt, dC, None, None, None, None


def selective_scan(dlt, A_log, Bt, C, h0=None, delta_softplus=False, y_dtype=torch.float32, return_last=False):
    """y[b,t,c] = C*s,  s_t = exp(delta_t*A)*s_{t-1} + Bt_t  (reference core.py:337-353).

    dlt [B,L,h] fp32 (delta, or its pre-softplus logits when delta_softplus), A_log [h,N],
    Bt/C [B,L,h*N] fp32 or bf16 (strided column slices are taken as they are), h0 [B,h*N] or
    None.  Returns y [B,L,h*N] (and the final state [B,h*N] when return_last)."""
    return _SelectiveScan.apply(dlt, A_log, Bt, C, h0, delta_softplus, y_dtype, return_last)


# ----------------------------------------------------------------------------------------------
# scan with the skip + gate fused in (no fp32 y / dy in HBM), single launch per direction
# ----------------------------------------------------------------------------------------------
# APERTIS_SCAN_SINGLE_PASS=0 selects the two-launch form of the same kernels (state pass + replay; same bits)
SCAN_SINGLE_PASS = _os.environ.get("APERTIS_SCAN_SINGLE_PASS", "1") != "0"


# APERTIS_SCAN_LEAN (round 4, default on; 0 = off): three lean launches per direction (state pass, chunk prefix, replay: a lane
# owns four channels of a row, a wave one 64-token item, nothing staged through LDS) for the shapes those kernels take (bf16,
# N = 16, 128 < Dn <= 256, 8-byte aligned slices); everything else on the forms above
SCAN_LEAN = _os.environ.get("APERTIS_SCAN_LEAN", "1") == "1"


SCAN_LEAN_BWD = _os.environ.get("APERTIS_SCAN_LEAN_BWD", "1") == "1"   # ... and the backward, from the lean forward's checkpoints


# APERTIS_SCAN_LOOKBACK (round 5; 1 = default, 0 = off, all): ONE launch per direction in the lean layout - a work-group per
# 64-token chunk, 16 tokens per wave held in registers, chunk carries by a decoupled look-back through the workspace below (every
# operand row read once; csrc/scan_lookback.hip) - for bf16, N = 16, Dn <= 256; takes precedence over the three-launch lean form.
# Default: where it is the fastest form measured (128 < Dn <= 256: 87 / 168 us against the staged kernels' 104 / 230 at the bench
# shape); narrower models stay on the staged kernels, which are as fast there (Dn = 64, B = 16, L = 4096: 32 / 53 us staged,
# 32 / 61 us look-back: profiles/r5_scan_lookback_vs_lean_vs_staged.log).  "all": every shape the entry points take (the tests).
_lb_env = _os.environ.get("APERTIS_SCAN_LOOKBACK", "1")


SCAN_LOOKBACK = "all" if _lb_env == "all" else _lb_env == "1"


SCAN_LOOKBACK_MIN_WGS = int(_os.environ.get("APERTIS_SCAN_LOOKBACK_MIN_WGS", "768"))   # (Dn <= 128: work-groups from which it is the default)
_gate_ws = {}      # (device, stream) -> [workspace (zeroed once), last epoch]


def _scan_gate_ws(lib, B, L, Dn, device):
    """Look-back workspace of the single-pass kernels: one per (device, stream), zero-filled once, and the epoch of the
    next launch on it (incremented by exactly one per launch: the two ticket counters in its head alternate)."""
    need = max(int(lib.apertis_scan_gate_workspace_bytes(B, L, Dn)), int(lib.apertis_scan_lookback_workspace_bytes(B, L, Dn)))
    device = _indexed(device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ent = _gate_ws.get(key)
    if ent is None or ent[0].numel() < need or ent[1] >= 0xFFFFFFF0:
        old = ent
        ent = _gate_ws[key] = [torch.zeros(need, device=device, dtype=torch.uint8), 0]
        if old is not None:
            ent[0][8:12].copy_(old[0][8:12])     # a larger workspace inherits the sticky error word of the one it replaces
    ent[1] += 1
    return ent[0], ent[1]


def _scan_gate_ws_unused(device):
    """The epoch handed out last was not launched with (the entry point declined the shape): take it back - the two ticket
    counters alternate with the epoch's parity, so an epoch that is skipped would leave the next launch a stale counter."""
    device = _indexed(device)
    _gate_ws[(device, torch.cuda.current_stream(device).cuda_stream)][1] -= 1


def scan_gate_error(device=None):
    """Non-zero if a single-pass scan launch hit its bounded-wait timeout on any workspace of `device` (host sync)."""
    bad = 0
    device = None if device is None else _indexed(device)
    for (dev, _), ent in _gate_ws.items():
        if device is None or dev == device:
            torch.cuda.synchronize(dev)
            bad |= int(ent[0][8:12].view(torch.int32).item())
    return bad


def scan_gate_error_word(device):
    """The look-back error word(s) of `device`'s single-pass scan workspaces as ONE device int32 tensor of shape [1]
    (None when no single-pass scan has run there) - no host sync.  `ApertisAdamW.step` hands its address to
    `apertis_clip_coef` as the poison word, `TrainStep` turns the returned loss into NaN with it."""
    device = _indexed(device)
    words = [ent[0][8:12].view(torch.int32) for (dev, _), ent in _gate_ws.items() if dev == device]
    if not words:
        return None
    if len(words) == 1:
        return words[0]
    return torch.cat(words).ne(0).any().to(torch.int32).reshape(1)


def scan_gate_raise_on_error(device=None):
    """Raise ApertisHipError if a single-pass scan launch timed out in its look-back (host sync: call it where the host
    waits anyway, e.g. behind `loss.item()`); the activations of such a launch are wrong."""
    bad = scan_gate_error(device)
    if bad:
        raise ApertisHipError(f"single-pass scan: look-back wait timed out (error word {bad:#x}); the step's activations "
                              "are invalid - rerun with APERTIS_SCAN_SINGLE_PASS=0 (two-launch form, same bits)")


def scan_gate_clear_error(device=None):
    """Zero the look-back error word(s) of `device` (all devices: None) after the caller has dealt with a time-out: the word
    is sticky - it poisons every later optimizer step (`ApertisAdamW`) until it is cleared."""
    device = None if device is None else _indexed(device)
    for (dev, _), ent in _gate_ws.items():
        if device is None or dev == device:
            ent[0][8:12].zero_()


# APERTIS_SCAN_DT_FUSED=1 (round 4, N4's prologue; default OFF): dt_proj_head inside the lean forward's state pass
# (ops.scan_gate_dt -> apertis_scan_lean_fwd_dt) instead of its own launch.  Same bits; measured in the 1.5B step at B = 44
# (profiles/r4_dtproj_fused_ab.txt): the state pass grows by 28.6 us per layer (its 44-term dot per (token, head) costs
# registers: 134 VGPRs, three waves per SIMD instead of eight) against the 19 us launch it replaces - scan forward 89.9 -> 118.5 us,
# the step within noise (455.0 / 452.5 ms fused, 455.7 / 455.3 ms two launches): not taken.
SCAN_DT_FUSED = _os.environ.get("APERTIS_SCAN_DT_FUSED", "0") == "1"


class _ScanGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dlt, A_log, Bt, C, xc, z, D, h0, delta_softplus, return_last):
        return _scan_gate_forward(ctx, dlt, A_log, Bt, C, xc, z, D, h0, delta_softplus, return_last,
                                  _grad_wanted(ctx, 7), None)

    @staticmethod
    def backward(ctx, dout, *_unused):
        if dout is None:
            return (None,) * 10
        return _scan_gate_backward(ctx, dout) + (None, None, None)


class _ScanGateDt(torch.autograd.Function):
    """scan_gate with the delta logits formed inside it: dlt = dt_in @ W_dt.T + b_dt (core.py:382) is computed by the lean
    forward's state pass where that kernel takes the shape, by the stand-alone kernel otherwise - the same bits either way."""

    @staticmethod
    def forward(ctx, dt_in, W_dt, b_dt, A_log, Bt, C, xc, z, D, h0, delta_softplus, return_last):
        _require_gpu(dt_in, W_dt, b_dt)
        R, h = dt_in.shape[-1], W_dt.shape[0]
        ctx.dt_slot = _slot_of(dt_in)
        xr, ldx = _rows(dt_in, R)
        w = _f32(W_dt)
        b = None if b_dt is None else _f32(b_dt)
        ctx.dt_cfg = (ldx, tuple(dt_in.shape), W_dt.dtype, None if b_dt is None else b_dt.dtype)
        dlt = torch.empty(*dt_in.shape[:-1], h, device=dt_in.device, dtype=torch.float32)
        res = _scan_gate_forward(ctx, dlt, A_log, Bt, C, xc, z, D, h0, delta_softplus, return_last,
                                 _grad_wanted(ctx, 9), (xr, ldx, w, b, R))
        ctx.dt_saved = (xr, w)     # (kept on ctx beside the tensors _scan_gate_forward saved: xr is a view of the projection output that the
        return res                 #  scan's own saved Bt / C slices keep alive and version-checked; w is an fp32 copy or the parameter itself)

    @staticmethod
    def backward(ctx, dout, *_unused):
        lib = _lib.load()
        if dout is None:
            return (None,) * 12
        d_dlt, dA, dBt, dC, dxc, dz, dD = _scan_gate_backward(ctx, dout)
        xr, w = ctx.dt_saved
        ldx, xshape, wdt, bdt = ctx.dt_cfg
        N, K = w.shape
        zero_to = K
        if ctx.dt_slot is not None and ctx.dt_slot[0].widths[ctx.dt_slot[1]] == K:
            dx, Kp = ctx.dt_slot[0].out(ctx.dt_slot[1], xshape[:-1], xr.dtype, xr.device)
            dxp = dx
            if dx.data_ptr() == ctx.dt_slot[0].buf.data_ptr() + ctx.dt_slot[0].offsets[ctx.dt_slot[1]] * dx.element_size():
                zero_to = K + ctx.dt_slot[0].zero_next(ctx.dt_slot[1])    # the pad columns behind dt: zeroed by this kernel
        else:
            Kp = -(-K // 8) * 8
            dxp = torch.empty(*xshape[:-1], Kp, device=xr.device, dtype=xr.dtype)
            dx = dxp[..., :K]
        T = d_dlt.numel() // N
        nblk = lib.apertis_tiny_linear_bwd_blocks(T)
        part = torch.empty(nblk, N * K + N, device=xr.device, dtype=torch.float32)
        out = torch.empty(N * K + N, device=xr.device, dtype=torch.float32)
        check(lib.apertis_tiny_linear_bwd_pad(ptr(xr), ldx, ptr(w), ptr(d_dlt), ptr(dxp), Kp, ptr(part), ptr(out), T, K, N, zero_to,
                                              dtype_code(xr), stream_ptr()), "apertis_tiny_linear_bwd")
        return (dx, out[:N * K].reshape(N, K).to(wdt), (out[N * K:].to(bdt) if bdt is not None else None),
                dA, dBt, dC, dxc, dz, dD, None, None, None)


def _scan_gate_forward(ctx, dlt, A_log, Bt, C, xc, z, D, h0, delta_softplus, return_last, need_grad, dtp):
    """Body of the two Functions above.  dtp = (dt rows, row stride, W fp32, b fp32 | None, R): `dlt` is an empty buffer
    that the launch (or, where the fused kernel does not take the shape, apertis_tiny_linear_fwd) fills."""
    _require_gpu(dlt, A_log, Bt, C, xc, z, D, h0)
    lib = _lib.load()
    B, L, h = dlt.shape
    N = A_log.shape[1]
    Dn = h * N
    wB, wC = Bt.shape[-1], C.shape[-1]          # >= Dn: zero-padded slices of the projection output
    if (tuple(Bt.shape[:2]) != (B, L) or tuple(C.shape[:2]) != (B, L) or wB < Dn or wC != wB or A_log.shape[0] != h or
            tuple(xc.shape) != (B, L, Dn) or tuple(z.shape) != (B, L, Dn) or wB > -(-Dn // 64) * 64):
        raise ApertisHipError(f"scan_gate shapes: dlt {tuple(dlt.shape)} A_log {tuple(A_log.shape)} Bt {tuple(Bt.shape)} "
                              f"C {tuple(C.shape)} xc {tuple(xc.shape)} z {tuple(z.shape)}")
    if not (Bt.dtype == C.dtype == xc.dtype == z.dtype):
        raise ApertisHipError("Bt, C, xc and z must share a dtype")
    dlt = dlt.float().contiguous()
    A_log = A_log.float().contiguous()
    Df = _f32(D)
    if h0 is not None:
        h0 = h0.float().reshape(B, Dn).contiguous()
    ctx.slots = (_slot_of(Bt), _slot_of(C), _slot_of(z), _slot_of(xc))
    (Bt, bt_rs), (C, c_rs), (xc, xc_rs), (z, z_rs) = _rows(Bt, wB), _rows(C, wC), _rows(xc, Dn), _rows(z, Dn)
    nch = -(-L // int(lib.apertis_scan_gate_chunk_len()))
    dev = dlt.device
    out = torch.empty(B, L, Dn, device=dev, dtype=xc.dtype)
    h_in = torch.empty(B, nch, Dn, device=dev, dtype=torch.float32)
    h_last = torch.empty(B, Dn, device=dev, dtype=torch.float32) if return_last else None
    e = xc.element_size()
    work = B * L * (5 * Dn * e + 4 * h) + 4 * h * N          # algorithmic bytes, fused variant
    # the lean form where it takes the shape (bf16, N = 16, 128 < Dn <= 256): timed under the same name - the same op
    ckpt, lean = None, False
    dt_done = dtp is None
    kind = "staged"
    # Dn <= 128 (64 / (Dn / 4) sequences side by side in a wave): the look-back form wins once its grid fills the chip - at
    # config 3's shape (Dn = 64, L = 4096) per-GPU batch 72 = 1152 work-groups: 58.8 / 107.3 us against the staged kernels'
    # 78.9 / 133.1; at batch 16 = 256 work-groups the 16-hop carry chain is its whole time and staged ties / wins (32 / 61
    # against 32 / 53 us) - profiles/r6_scan_config3_forms.txt
    lb_small = Dn <= 128 and Dn % 4 == 0 and -(-B // max(1, 64 // max(Dn // 4, 1))) * nch >= SCAN_LOOKBACK_MIN_WGS
    if SCAN_LOOKBACK and xc.dtype == torch.bfloat16 and N == 16 and Dn <= 256 and (Dn > 128 or lb_small or SCAN_LOOKBACK == "all"):
        if not dt_done:
            _tiny_linear_into(lib, dtp, dlt)
            dt_done = True
        ws, epoch = _scan_gate_ws(lib, B, L, Dn, dev)
        ckpt = torch.empty(B, -(-L // 16), Dn, device=dev, dtype=torch.float32) if need_grad else None
        lean = _try_launch("apertis_scan_gate_fwd", lib.apertis_scan_lookback_fwd,
                           (ptr(dlt), ptr(A_log), ptr(Bt), bt_rs, ptr(C), c_rs, ptr(xc), xc_rs, ptr(z), z_rs, ptr(Df), ptr(h0),
                            ptr(out), out.stride(-2), ptr(h_last), ptr(None if need_grad else h_in), ptr(ckpt), ptr(ws), epoch,
                            B, L, h, N, int(delta_softplus), stream_ptr()), work, unwind=lambda: _scan_gate_ws_unused(dev))
        if lean:
            kind = "lookback"
            if need_grad:
                h_in = None              # (ckpt16[:, ::4] is the state entering every chunk)
        else:
            ckpt = None
    if not lean and SCAN_LEAN and xc.dtype == torch.bfloat16 and N == 16 and 128 < Dn <= 256:
        kind = "lean"
        agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
        ckpt = torch.empty(B, -(-L // 4), Dn, device=dev, dtype=torch.float32) if need_grad else None
        rc = []
        tail = (ptr(A_log), ptr(Bt), bt_rs, ptr(C), c_rs, ptr(xc), xc_rs, ptr(z), z_rs, ptr(Df), ptr(h0), ptr(out),
                out.stride(-2), ptr(h_last), ptr(agg), ptr(h_in), ptr(ckpt), B, L, h, N, int(delta_softplus), stream_ptr())
        if dtp is not None and SCAN_DT_FUSED and dtp[0].dtype == torch.bfloat16:
            xr, ldx, w, b, R = dtp
            _launch("apertis_scan_gate_fwd", lambda *a: rc.append(lib.apertis_scan_lean_fwd_dt(*a)) or (0 if rc[-1] == -2 else rc[-1]),
                    (ptr(xr), ldx, ptr(w), ptr(b), R, ptr(dlt)) + tail, work + B * L * (2 * R + 4 * h))
            dt_done = lean = rc[-1] == 0
        if not lean:
            if not dt_done:
                _tiny_linear_into(lib, dtp, dlt)
                dt_done = True
            _launch("apertis_scan_gate_fwd", lambda *a: rc.append(lib.apertis_scan_lean_fwd(*a)) or (0 if rc[-1] == -2 else rc[-1]),
                    (ptr(dlt),) + tail, work)
            lean = rc[-1] == 0                   # (-2 = APERTIS_ERR_UNSUPPORTED: alignment / size - the staged kernels below)
    if not dt_done:
        _tiny_linear_into(lib, dtp, dlt)
    if not lean:
        ckpt, kind = None, "staged"
        if SCAN_SINGLE_PASS:
            ws, epoch = _scan_gate_ws(lib, B, L, Dn, dev)
            agg = None
        else:
            ws, epoch = None, 0
            agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
        _launch("apertis_scan_gate_fwd", lib.apertis_scan_gate_fwd,
                (ptr(dlt), ptr(A_log), ptr(Bt), bt_rs, ptr(C), c_rs, ptr(xc), xc_rs, ptr(z), z_rs, ptr(Df), ptr(h0), ptr(out),
                 out.stride(-2), ptr(h_last), ptr(agg), ptr(h_in), ptr(ws), epoch, B, L, h, N, dtype_code(xc),
                 int(delta_softplus), int(SCAN_SINGLE_PASS), stream_ptr()), work,
                unwind=(lambda: _scan_gate_ws_unused(dev)) if SCAN_SINGLE_PASS else None)
    ctx.save_for_backward(dlt, A_log, Bt, C, xc, z, Df, h_in, ckpt)
    ctx.cfg = (B, L, h, N, bool(delta_softplus), wB, D.dtype)
    ctx.scan_kind = kind
    ctx.mark_non_differentiable(*([h_last] if return_last else []))
    ctx.set_materialize_grads(False)          # (autograd otherwise zero-fills a [B, Dn] gradient for h_last: one launch per layer)
    return (out, h_last) if return_last else out


def _tiny_linear_into(lib, dtp, dlt):
    xr, ldx, w, b, R = dtp
    check(lib.apertis_tiny_linear_fwd(ptr(xr), ldx, ptr(w), ptr(b), ptr(dlt), dlt.numel() // dlt.shape[-1], R, dlt.shape[-1],
                                      dtype_code(xr), stream_ptr()), "apertis_tiny_linear_fwd")


def _scan_gate_backward(ctx, dout):
    """-> (d_dlt, dA_log, dBt, dC, dxc, dz, dD)"""
    lib = _lib.load()
    dlt, A_log, Bt, C, xc, z, Df, h_in, ckpt = ctx.saved_tensors
    B, L, h, N, sp, wB, Ddt = ctx.cfg
    Dn = h * N
    dev = dlt.device
    dout, do_rs = _rows(dout.to(xc.dtype), Dn)
    kind = ctx.scan_kind
    nch = -(-L // int(lib.apertis_scan_gate_chunk_len()))
    dBt, dbt_rs = _grad_out(ctx.slots[0], (B, L), wB, Bt.dtype, dev)
    dC, dc_rs = _grad_out(ctx.slots[1], (B, L), wB, C.dtype, dev)
    dz, dz_rs = _grad_out(ctx.slots[2], (B, L), Dn, z.dtype, dev)
    dxc, dxc_rs = _grad_out(ctx.slots[3], (B, L), Dn, xc.dtype, dev)
    d_dlt = torch.empty(B, L, h, device=dev, dtype=torch.float32)
    dA_dD = torch.empty(2, Dn, device=dev, dtype=torch.float32)
    part = torch.empty(B * nch, 2 * Dn, device=dev, dtype=torch.float32)
    fold = torch.empty(64, 2 * Dn, device=dev, dtype=torch.float32)
    e = xc.element_size()
    work = B * L * (9 * Dn * e + 8 * h) + 8 * h * N          # algorithmic bytes, fused variant
    if kind == "lookback":                       # the look-back forward left the state entering every 16th token
        ws, epoch = _scan_gate_ws(lib, B, L, Dn, dev)
        if _try_launch("apertis_scan_gate_bwd", lib.apertis_scan_lookback_bwd,
                       (ptr(dlt), ptr(A_log), ptr(Bt), Bt.stride(-2), ptr(C), C.stride(-2), ptr(xc), xc.stride(-2), ptr(z),
                        z.stride(-2), ptr(Df), ptr(dout), do_rs, ptr(ckpt), ptr(dBt), dbt_rs, ptr(dC), dc_rs, wB, ptr(dxc), dxc_rs,
                        ptr(dz), dz_rs, ptr(d_dlt), ptr(dA_dD), ptr(fold), ptr(part), ptr(ws), epoch, B, L, h, N, int(sp),
                        stream_ptr()), work, unwind=lambda: _scan_gate_ws_unused(dev)):
            return d_dlt, dA_dD[0].reshape(h, N), dBt, dC, dxc, dz, dA_dD[1].to(Ddt)
        h_in = ckpt[:, ::4].contiguous()         # (declined: the staged kernels below, from the chunk-entry states)
    if kind == "lean" and ckpt is not None and SCAN_LEAN_BWD:       # the lean forward left its checkpoints: the lean backward
        agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
        mu_in = torch.empty(B, nch, Dn, device=dev, dtype=torch.float32)
        rc = []
        _launch("apertis_scan_gate_bwd", lambda *a: rc.append(lib.apertis_scan_lean_bwd(*a)) or (0 if rc[-1] == -2 else rc[-1]),
                (ptr(dlt), ptr(A_log), ptr(Bt), Bt.stride(-2), ptr(C), C.stride(-2), ptr(xc), xc.stride(-2), ptr(z), z.stride(-2),
                 ptr(Df), ptr(dout), do_rs, ptr(ckpt), ptr(dBt), dbt_rs, ptr(dC), dc_rs, wB, ptr(dxc), dxc_rs, ptr(dz), dz_rs,
                 ptr(d_dlt), ptr(dA_dD), ptr(agg), ptr(mu_in), ptr(fold), ptr(part), B, L, h, N, int(sp), stream_ptr()), work)
        if rc[-1] == 0:
            return d_dlt, dA_dD[0].reshape(h, N), dBt, dC, dxc, dz, dA_dD[1].to(Ddt)
    if SCAN_SINGLE_PASS:
        ws, epoch = _scan_gate_ws(lib, B, L, Dn, dev)
        agg = None
    else:
        ws, epoch = None, 0
        agg = torch.empty(B, nch, Dn, 2, device=dev, dtype=torch.float32)
    _launch("apertis_scan_gate_bwd", lib.apertis_scan_gate_bwd,
            (ptr(dlt), ptr(A_log), ptr(Bt), Bt.stride(-2), ptr(C), C.stride(-2), ptr(xc), xc.stride(-2), ptr(z), z.stride(-2),
             ptr(Df), ptr(dout), do_rs, ptr(h_in), ptr(dBt), dbt_rs, ptr(dC), dc_rs, wB, ptr(dxc), dxc_rs, ptr(dz), dz_rs,
             ptr(d_dlt), ptr(dA_dD), ptr(agg), ptr(fold), ptr(part), ptr(ws), epoch, B, L, h, N, dtype_code(xc), int(sp),
             int(SCAN_SINGLE_PASS), stream_ptr()), work, unwind=(lambda: _scan_gate_ws_unused(dev)) if SCAN_SINGLE_PASS else None)
    return d_dlt, dA_dD[0].reshape(h, N), dBt, dC, dxc, dz, dA_dD[1].to(Ddt)


def scan_gate(dlt, A_log, Bt, C, xc, z, D, h0=None, delta_softplus=False, return_last=False):
    """(C*s + D*xc) * silu(z) with s_t = exp(delta_t*A)*s_{t-1} + Bt_t: the recurrence (reference core.py:337-353) and
    the skip + gate (core.py:395-396) in ONE kernel per direction; y is never written (the backward recomputes it).

    dlt [B,L,h] fp32 (pre-softplus logits when delta_softplus), A_log [h,N], xc / z [B,L,h*N], D [h*N];
    Bt / C [B,L,w] with h*N <= w <= ceil(h*N/64)*64: the (possibly zero-padded) column slices of the projection output,
    of which the first h*N columns are used; their gradients come back [B,L,w] with zeros in the pad.
    Returns out [B,L,h*N] in the activations' dtype (and the final state [B,h*N] fp32 when return_last)."""
    return _apply(_ScanGate, dlt, A_log, Bt, C, xc, z, D, h0, delta_softplus, return_last)


def scan_gate_dt(dt_in, W_dt, b_dt, A_log, Bt, C, xc, z, D, h0=None, delta_softplus=True, return_last=False):
    """scan_gate(tiny_linear(dt_in, W_dt, b_dt), ...) as ONE op (reference core.py:382-396).  With APERTIS_SCAN_DT_FUSED=1
    dt_proj_head runs inside the lean forward's state pass where that kernel takes the shape (N4's pre-scan prologue; off by
    default - measured slower than the launch it replaces, see SCAN_DT_FUSED); otherwise the stand-alone kernel fills the
    logits.  The logits, outputs and gradients are the two-op form's bit for bit either way.  dt_in [B, L, R] (a column slice
    of the projection output is read in place), W_dt [h, R], b_dt [h] or None."""
    return _apply(_ScanGateDt, dt_in, W_dt, b_dt, A_log, Bt, C, xc, z, D, h0, delta_softplus, return_last)
