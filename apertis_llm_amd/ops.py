"""Host-side operators: torch.autograd.Functions over the C ABI (libapertis_hip.so).

torch is used for device memory, streams and autograd bookkeeping only; every computation
below is a HIP kernel launch through apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import torch

from . import _lib
from ._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr


def _f32(t):
    """`t` as a contiguous fp32 tensor for a kernel argument - `t` itself when it already is one: `Tensor.detach()` on a
    parameter costs ~14 us of host time (1 500 of them per step of the 76-layer configuration, 15 % of its enqueue time), and
    inside a Function.forward / under no_grad nothing is recorded anyway."""
    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    return t.detach().float().contiguous()


_TIMER = None


def set_kernel_timer(timer):
    """Install (or clear with None) a KernelTimer: the named C-ABI calls are bracketed with HIP
    events on the launch stream so bench.py can report per-launch durations."""
    global _TIMER
    _TIMER = timer


class KernelTimer:
    """Collects (name, start_event, end_event, work) for selected entry points.  `work` is the
    algorithmic bytes or flops of the call, or a callable evaluated after the run (for counts
    that live on the device)."""

    def __init__(self, names, every=1):
        """`every` = k: only every k-th step is bracketed (the caller counts steps with next_step()); two event records
        per call cost the host ~17 us - 24 ms of a 137 ms step on the launch-heavy H = 256 configuration."""
        self.names = set(names)
        self.records = []
        self.every, self._step, self.on = max(int(every), 1), 0, True

    def next_step(self):
        self._step += 1
        self.on = self._step % self.every == 0

    def summary(self):
        """{name: {launches, ms, work, by_shape}}; `by_shape` splits the calls that gave a shape tag (the dense
        projections: {tag: {launches, ms, work, bytes}} with the call's algorithmic bytes)."""
        torch.cuda.synchronize()
        out = {}
        for name, s, e, work, detail, nbytes in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0, "by_shape": {}})
            ms, wk = s.elapsed_time(e), float(work() if callable(work) else work)
            d["launches"] += 1
            d["ms"] += ms
            d["work"] += wk
            if detail is not None:
                b = d["by_shape"].setdefault(detail, {"launches": 0, "ms": 0.0, "work": 0.0, "bytes": 0.0})
                b["launches"] += 1
                b["ms"] += ms
                b["work"] += wk
                b["bytes"] += float(nbytes)
        return out


def _launch(name, fn, args, work=0.0, detail=None, nbytes=0.0, unwind=None):
    """`unwind` (optional) runs when the entry point returns non-zero, BEFORE check() raises: a caller that took a per-launch
    resource (the scan workspace's epoch) gives it back - no kernel went out, and a consumed epoch would leave the next launch
    on that stream a stale ticket counter."""
    t = _TIMER
    if t is None or not t.on or name not in t.names:
        rc = fn(*args)
        if rc != 0 and unwind is not None:
            unwind()
        check(rc, name.split("[")[0])
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn(*args)
    e.record()
    if rc != 0 and unwind is not None:
        unwind()
    check(rc, name.split("[")[0])
    t.records.append((name, s, e, work, detail, nbytes))


def _try_launch(name, fn, args, work=0.0, unwind=None):
    """_launch for an entry point that may decline the shape: returns False on APERTIS_ERR_UNSUPPORTED (-2) WITHOUT recording
    a timed launch (the caller then takes another form, which records its own), True when the launch went out.  `unwind` runs
    on EVERY non-zero return (declined, or an argument / launch error that check() is about to raise): see _launch."""
    t = _TIMER
    timed = t is not None and t.on and name in t.names
    if timed:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    rc = fn(*args)
    if rc != 0 and unwind is not None:
        unwind()
    if rc == -2:
        return False
    if timed:
        e.record()
    check(rc, name.split("[")[0])
    if timed:
        t.records.append((name, s, e, work, None, 0.0))
    return True


def _require_gpu(*ts):
    """Every tensor of a call lives on ONE ROCm device and that device is the current one: the library launches on the
    current HIP device and on torch's current stream of it (`_lib.stream_ptr`), so a tensor elsewhere would be touched
    by a kernel on another GPU's stream.  The model's forward switches to its tensors' device (`device_guard`); a
    direct caller of an op on a non-current device gets this error instead of a silent cross-device launch."""
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise ApertisHipError("apertis_llm_amd ops run on a ROCm device only (tensor on %s); "
                                  "there is no CPU fallback" % t.device)
        if dev is None:
            dev = t.device.index
        elif t.device.index != dev:
            raise ApertisHipError(f"tensors of one call on different devices (cuda:{dev} and {t.device})")
    if dev is not None and dev != torch.cuda.current_device():
        raise ApertisHipError(f"tensor on cuda:{dev} but the current device is cuda:{torch.cuda.current_device()}: "
                              "call torch.cuda.set_device() / run under `with torch.cuda.device(t.device)` "
                              "(apertis_llm_amd.ops.device_guard)")


class device_guard:
    """`with device_guard(t):` makes t's device the current one for the block (no-op when it already is, or off-GPU)."""
    __slots__ = ("idx", "prev")

    def __init__(self, t):
        self.idx = t.device.index if (t is not None and t.is_cuda) else None
        self.prev = None

    def __enter__(self):
        if self.idx is not None:
            cur = torch.cuda.current_device()
            if cur != self.idx:
                self.prev = cur
                torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
            self.prev = None
        return False


def _rows(t, width):
    """(tensor, row_stride): a [B,L,width] view usable by the kernels as is (unit inner stride,
    constant row stride, batch stride = L*row_stride), else a packed copy."""
    ok = t.stride(-1) == 1 and t.stride(-2) >= width
    if ok and t.dim() == 3 and t.shape[0] > 1 and t.stride(0) != t.shape[1] * t.stride(1):
        ok = False
    if not ok:
        t = t.contiguous()
    return t, t.stride(-2)


# ----------------------------------------------------------------------------------------------
# selective scan
# ----------------------------------------------------------------------------------------------
import os as _os
import threading

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


def _indexed(device):
    """torch.device with an explicit index ('cuda' -> the current device): the workspace tables are keyed on it."""
    d = torch.device(device)
    return torch.device("cuda", torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d


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


# Whether the CALLER of an op runs under autograd.  Inside Function.forward grad mode is always off, and
# ctx.needs_input_grad only says whether an input is a tensor that requires grad - a Parameter does, under torch.no_grad() too.
# Ops that decide in their forward what to keep for a backward (transposed weight copies, the pre-activation / saved-gradient
# output of the expert MLP, scan checkpoints) read the mode their public wrapper recorded: under no_grad (generate(), eval)
# they used to prepare for a backward that never comes - every token step of a decode re-cast the expert weights.
_tls = threading.local()


def _apply(fn, *args):
    _tls.grad = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _tls.grad = True          # (a direct Function.apply keeps the conservative default)


def _grad_wanted(ctx, n):
    return getattr(_tls, "grad", True) and any(ctx.needs_input_grad[:n])


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
    if SCAN_LOOKBACK and xc.dtype == torch.bfloat16 and N == 16 and Dn <= 256 and (Dn > 128 or SCAN_LOOKBACK == "all"):
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


class _ColSlot:
    """Gradient buffer shared by the column views of one split_cols call.  Ops of this module that consume such a
    view (`x._apertis_slot = (slot, i)`) write their input gradient straight into columns [off_i, off_i + w_i)
    of `buf` (their kernels take an output row stride), so split_cols' backward is the buffer itself instead of
    a concatenation."""
    __slots__ = ("widths", "offsets", "total", "buf", "claimed", "zeroed")

    def __init__(self, widths):
        self.widths = tuple(widths)
        self.offsets = tuple(sum(widths[:i]) for i in range(len(widths)))
        self.total = sum(widths)
        self.buf = None
        self.claimed = set()
        self.zeroed = set()

    def zero_next(self, i):
        """The op that wrote view i offers to zero the view behind it in the same kernel (the pad columns of a padded
        projection output): returns that view's width when nobody has claimed it - the op's kernel then writes zeros there and
        _SplitCols.backward skips its fill - else 0."""
        j = i + 1
        if self.buf is None or j >= len(self.widths) or j in self.claimed or self.widths[j] == 0:
            return 0
        self.zeroed.add(j)
        return self.widths[j]

    def out(self, i, lead_shape, dtype, device):
        """(gradient tensor for view i, its row stride in elements).  A view's columns are handed out once per
        backward pass: a second consumer of the same view gets a tensor of its own and autograd adds the two."""
        if self.buf is None:
            self.buf = torch.empty(*lead_shape, self.total, device=device, dtype=dtype)
            self.claimed = set()
            self.zeroed = set()
        if i in self.claimed or self.buf.dtype != dtype or tuple(self.buf.shape[:-1]) != tuple(lead_shape):
            return torch.empty(*lead_shape, self.widths[i], device=device, dtype=dtype), self.widths[i]
        self.claimed.add(i)
        return self.buf[..., self.offsets[i]:self.offsets[i] + self.widths[i]], self.total


def _slot_of(t):
    return getattr(t, "_apertis_slot", None)


def _grad_out(slot, lead_shape, width, dtype, device):
    """Gradient buffer for an input that may be a split_cols view: (tensor [..., width], row stride)."""
    if slot is not None and slot[0].widths[slot[1]] == width:
        return slot[0].out(slot[1], lead_shape, dtype, device)
    return torch.empty(*lead_shape, width, device=device, dtype=dtype), width


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


class _RowsGrad:
    """Hand-over between the gather-LN backward and the router backward of the SAME pass-through x: instead of a dense
    [T, H] gradient (written by a combine kernel, read back by the router kernel) the gather op leaves its gradient ROWS and
    the slot table here and returns a zero placeholder without storage; the router kernel gathers the rows itself.  If
    autograd adds other consumers' gradients to the placeholder they arrive as an ordinary dense `dres` on top."""
    __slots__ = ("rows", "slot_of", "K")

    def __init__(self):
        self.rows = self.slot_of = None
        self.K = 0

    def take(self):
        r = (self.rows, self.slot_of, self.K)
        self.rows = self.slot_of = None
        return r


_ZERO1 = {}


def _zero_placeholder(shape, device, dtype):
    z = _ZERO1.get((device, dtype))
    if z is None:
        z = _ZERO1[(device, dtype)] = torch.zeros(1, device=device, dtype=dtype)
    return z.expand(*shape)


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


def skinny_linear_supported(K, N):
    return N in (2, 4, 8, 16) and K % 4 == 0 and K <= 1024 and (N <= 8 or K <= 256)


def skinny_linear(x, weight, bias=None):
    """fp32 y = x @ W.T + b for a handful of output columns (the MoE router, reference core.py:482):
    bandwidth-bound row kernel instead of a GEMM-library call."""
    return _SkinnyLinear.apply(x, weight, bias)


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


# Prepared-weight cache of the INFERENCE path (generate() / chat decode every token through the same weights; reference
# core.py:1578-1603).  An entry belongs to one SOURCE tensor object - a parameter, or a tensor an earlier cached_prep call
# produced and therefore keeps alive - and is valid while that object is the same (weak reference: a temporary that happens
# to be allocated where a dead tensor lived never hits), its version counter has not moved and WEIGHT_EPOCH is the one it
# was made in: ApertisAdamW bumps the epoch on every step, because its kernels update parameters through raw pointers,
# which the version counter does not see.  Training never reads the cache (autograd Functions pass cache=False when an
# input needs a gradient): its compute copies change with every optimizer step anyway.
# The cache lives only INSIDE a prep_cache_scope() (generate(), the trainer's validation loop): an in-place write through
# `.data` (weight init, `resize_token_embeddings`, a DDP parameter broadcast, a user's `p.data.copy_(...)`) bumps neither the
# version counter nor the epoch, so outside a scope - where such writes happen between forwards - every forward prepares
# its copies afresh, and a scope drops its entries when it closes.
WEIGHT_EPOCH = 0
_prep_cache = {}
_prep_scope_depth = 0


class prep_cache_scope:
    """`with ops.prep_cache_scope():` - the span during which prepared inference copies of the weights may be reused
    (the weights must not be written inside it except through ApertisAdamW, which invalidates them)."""

    def __enter__(self):
        global _prep_scope_depth
        if _prep_scope_depth == 0:
            _prep_cache.clear()
        _prep_scope_depth += 1
        return self

    def __exit__(self, *exc):
        global _prep_scope_depth
        _prep_scope_depth -= 1
        if _prep_scope_depth == 0:
            _prep_cache.clear()
        return False


def note_weights_changed():
    """Parameters were updated behind torch's back (a HIP optimizer kernel): prepared inference copies are stale."""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1
    _prep_cache.clear()


def _stable_source(t):
    """The tensor object whose identity may key the cache: t itself or the tensor it is a view of, if that is a parameter
    or the product of a cached_prep call; else None."""
    base = t._base if t._base is not None else t
    return base if (isinstance(base, torch.nn.Parameter) or getattr(base, "_apertis_prepared", False)) else None


def cached_prep(tag, tensors, make, enable=None):
    """`make()` memoised on (tag, the source tensors' identity + version); only under torch.no_grad (`enable` overrides:
    inside an autograd Function's forward grad mode is off although the call may belong to a training step) and only
    for stable sources (_stable_source) - otherwise just `make()`."""
    import weakref
    if _prep_scope_depth == 0:
        return make()
    if enable is None:
        enable = not torch.is_grad_enabled()
    srcs = [_stable_source(t) for t in tensors] if enable else [None]
    if any(x is None for x in srcs):
        return make()
    key = (tag, tuple(id(x) for x in srcs))
    # (strides and offset: `w` and `w.t()` of a square weight share pointer, shape and dtype)
    state = tuple((t.data_ptr(), x._version, tuple(t.shape), tuple(t.stride()), t.storage_offset(), t.dtype)
                  for t, x in zip(tensors, srcs)) + (WEIGHT_EPOCH,)
    ent = _prep_cache.get(key)
    if ent is None or ent[0] != state or any(r() is not x for r, x in zip(ent[1], srcs)):
        if len(_prep_cache) > 8192:      # (models come and go in one process: do not keep their copies for ever)
            _prep_cache.clear()
        val = make()
        for v in (val if isinstance(val, (tuple, list)) else (val,)):
            if isinstance(v, torch.Tensor):
                v._apertis_prepared = True          # kept alive by the entry: a stable source for the next level
        ent = _prep_cache[key] = (state, [weakref.ref(x) for x in srcs], val)
    return ent[2]


# ----------------------------------------------------------------------------------------------
# Training-step weight preparation in ONE launch (round 4).  A training forward needs, per GEMM weight, a bf16 copy (K zero-
# padded to 64) and - for the data gradient - a transposed bf16 copy; the SSM block additionally stacks in_proj_x | in_proj_z
# and permutes / pads x_param_proj's rows.  Done per call that was 7 launches per layer and step (cat, scatter, five
# casts).  A TrainPrep holds persistent destination buffers for every registered weight of a model and a device table of
# them; `refresh()` - called by training.TrainStep at the START of every step, so a write to the weights between steps,
# through whatever door, is always seen - fills all of them with one apertis_weight_prep launch, and inside
# `with prep.active():` cast_transpose() / prepared_weight() hand out those buffers instead of making copies.
# Only bf16 compute copies; anything not registered (or another dtype) takes the per-call path as before.
# ----------------------------------------------------------------------------------------------
_ACTIVE_TRAIN_PREP = None
TRAIN_PREP = _os.environ.get("APERTIS_TRAIN_PREP", "1") == "1"


class _PrepEntry:
    __slots__ = ("sources", "rows", "cols", "plain", "tr", "kind", "rowmap", "rowmap64", "rows_out")


class TrainPrep:
    def __init__(self, device):
        self.device = _indexed(device)
        self.entries = []
        self.by_param = {}        # id(parameter) -> entry (plain weights: the parameter itself is what the op receives)
        self.by_key = {}          # (tag, id(module)) -> entry (stacked / row-mapped weights: the op receives a placeholder)
        self.table = None
        self.total_tiles = 0
        self._records = []

    @staticmethod
    def _ok(*ws):
        return all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.shape[-1] % 4 == 0 and w.data_ptr() % 16 == 0
                   for w in ws)

    def _alloc(self, E, rows, cols):
        Cp, Rp = -(-cols // 64) * 64, -(-rows // 64) * 64
        return (torch.zeros(E, rows, Cp, device=self.device, dtype=torch.bfloat16),
                torch.zeros(E, cols, Rp, device=self.device, dtype=torch.bfloat16))

    def add_plain(self, param):
        """[R, C] or [E, R, C] parameter -> copies shaped as cast_transpose's ([E, R, C'], [E, C, R'])."""
        w = param if param.dim() == 3 else param.unsqueeze(0)
        if param.dim() not in (2, 3) or not self._ok(param) or id(param) in self.by_param:
            return False
        E, R, C = w.shape
        en = _PrepEntry()
        en.sources, en.rows, en.cols, en.kind, en.rowmap, en.rowmap64, en.rows_out = (param,), R, C, "plain", None, None, R
        en.plain, en.tr = self._alloc(E, R, C)
        for e in range(E):
            self._records.append((param, e * R * C * 4, en.plain[e], en.tr[e], None, R, C, 0))
        self.entries.append(en)
        self.by_param[id(param)] = en
        return True

    def add_stack(self, key, params):
        """Row-stacked [sum R_i, C] weight of several [R_i, C] parameters (in_proj_x | in_proj_z)."""
        if key in self.by_key or not self._ok(*params) or len({p.shape[1] for p in params}) != 1:
            return False
        C, R = params[0].shape[1], sum(p.shape[0] for p in params)
        if C % 8 or R % 8:         # (model._mfma_linear's predicate: other shapes go to F.linear, which would READ the placeholder)
            return False
        en = _PrepEntry()
        en.sources, en.rows, en.cols, en.kind, en.rowmap, en.rowmap64, en.rows_out = tuple(params), R, C, "stack", None, None, R
        en.plain, en.tr = self._alloc(1, R, C)
        r0 = 0
        for p_ in params:
            self._records.append((p_, 0, en.plain[0], en.tr[0], None, p_.shape[0], C, r0))
            r0 += p_.shape[0]
        self.entries.append(en)
        self.by_key[key] = en
        return True

    def add_rowmap(self, key, param, dst_idx, rows_out):
        """[R, C] parameter whose row r lands in row dst_idx[r] of a [rows_out, C] weight, the other rows zero (x_param_proj in
        the scan's padded layout)."""
        if key in self.by_key or not self._ok(param) or param.shape[1] % 8 or rows_out % 8:   # (as add_stack)
            return False
        R, C = param.shape
        en = _PrepEntry()
        en.sources, en.rows, en.cols, en.kind, en.rows_out = (param,), R, C, "rowmap", rows_out
        en.rowmap = dst_idx.to(device=self.device, dtype=torch.int32).contiguous()
        en.rowmap64 = en.rowmap.long()            # (for the backward's index_select: not converted per step)
        en.plain, en.tr = self._alloc(1, rows_out, C)
        self._records.append((param, 0, en.plain[0], en.tr[0], en.rowmap, R, C, 0))
        self.entries.append(en)
        self.by_key[key] = en
        return True

    def finalize(self):
        """Builds the device table (sources' addresses are read here: the parameters must stay where they are)."""
        import struct
        lib = _lib.load()
        assert lib.apertis_weight_prep_entry_bytes() == 64
        blob, tile0 = bytearray(), 0
        self._addr = []
        for (src, off, plain, tr, rowmap, R, C, r0) in self._records:
            tiles_c = -(-C // 64)
            ldp, ldt = plain.shape[-1], tr.shape[-1]
            # a stacked source's rows start at row r0 of the destination: plain base + r0 rows, transposed base + r0 columns
            blob += struct.pack("<QQQQiiiiiiii", src.data_ptr() + off, plain.data_ptr() + r0 * ldp * 2, tr.data_ptr() + r0 * 2,
                                0 if rowmap is None else rowmap.data_ptr(), R, C, ldp, ldt, tiles_c, tile0, 0, 0)
            tile0 += -(-R // 64) * tiles_c
            self._addr.append((src, src.data_ptr()))
        self.total_tiles = tile0
        self.n_records = len(self._records)
        if self.n_records:
            self.table = torch.frombuffer(blob, dtype=torch.uint8).to(self.device)      # (bytearray: writable)
        return self

    def refresh(self):
        """All copies from the parameters' current values: one launch."""
        if not self.n_records:
            return
        for src, addr in self._addr:
            if src.data_ptr() != addr:
                raise ApertisHipError("TrainPrep: a registered parameter moved (model.to(...) after the first step?): build a new "
                                      "TrainStep / TrainPrep")
        check(_lib.load().apertis_weight_prep(ptr(self.table), self.n_records, self.total_tiles, stream_ptr()), "apertis_weight_prep")

    def active(self):
        return _TrainPrepScope(self)


class _TrainPrepScope:
    def __init__(self, prep):
        self.prep = prep

    def __enter__(self):
        global _ACTIVE_TRAIN_PREP
        self.prev, _ACTIVE_TRAIN_PREP = _ACTIVE_TRAIN_PREP, self.prep
        return self.prep

    def __exit__(self, *exc):
        global _ACTIVE_TRAIN_PREP
        _ACTIVE_TRAIN_PREP = self.prev
        return False


class _PreparedWeight(torch.autograd.Function):
    """Stand-in for a stacked / row-mapped fp32 weight whose compute copies a TrainPrep holds: an UNINITIALISED [rows, C]
    tensor that only carries shape, dtype and the gradient route - the GEMM ops find the copies on it (`_apertis_prep`) and
    never read its values.  backward: the weight's gradient back to the source parameters' layouts."""

    @staticmethod
    def forward(ctx, en, *sources):
        ctx.en = en
        return torch.empty(en.rows_out, en.cols, device=sources[0].device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, dw):
        en = ctx.en
        if en.kind == "stack":
            outs, r0 = [], 0
            for p_ in en.sources:
                outs.append(dw[r0:r0 + p_.shape[0]])
                r0 += p_.shape[0]
            return (None, *outs)
        return None, dw.index_select(0, en.rowmap64)


def prepared_weight(key, sources):
    """The placeholder of a registered stacked / row-mapped weight inside an active TrainPrep scope under bf16 autocast, else
    None (the caller then builds the weight itself)."""
    tp = _ACTIVE_TRAIN_PREP
    if tp is None or not torch.is_autocast_enabled() or torch.get_autocast_dtype("cuda") != torch.bfloat16:
        return None
    en = tp.by_key.get(key)
    if en is None or len(en.sources) != len(sources) or any(a is not b for a, b in zip(en.sources, sources)):
        return None
    w = _PreparedWeight.apply(en, *sources)
    w._apertis_prep = en
    return w


def _train_prep_lookup(w, dtype):
    tp = _ACTIVE_TRAIN_PREP
    if tp is None or dtype != torch.bfloat16:
        return None
    base = w._base if w._base is not None else w
    en = getattr(base, "_apertis_prep", None)
    if en is None:
        en = tp.by_param.get(id(base))
        if en is not None and en.sources[0] is not base:
            en = None
    if en is None:
        return None
    E, R, C = w.shape
    if (E, R, C) != (en.plain.shape[0], en.rows_out, en.cols):
        return None
    return en.plain, en.tr


def cast_transpose(w, dtype, want_plain=True, want_transposed=True, cache=False):
    """Compute copies of an fp32 master weight [E,R,C]: ([E,R,C'], [E,C,R']) in `dtype`.  In bf16 the
    last dimension is zero-padded to a multiple of 64 (C', R'): the GEMM's W operand then has whole
    64-wide K steps whatever K is; pass `.shape[-1]` as its row pitch (ldw).  cache=True (inference: no input of the
    calling op needs a gradient): the result is kept per weight (cached_prep), and an fp32 plain copy of an fp32 weight
    is the weight itself."""
    if _ACTIVE_TRAIN_PREP is not None and w.is_cuda and w.dim() == 3:
        hit = _train_prep_lookup(w, dtype)
        if hit is not None:
            return hit
    if cache and w.is_cuda:
        if dtype == torch.float32 and w.dtype == torch.float32 and want_plain and not want_transposed and w.is_contiguous():
            return w.detach(), None
        return cached_prep(("cast", dtype, want_plain, want_transposed), (w,),
                           lambda: _cast_transpose(w, dtype, want_plain, want_transposed), enable=True)
    return _cast_transpose(w, dtype, want_plain, want_transposed)


def _cast_transpose(w, dtype, want_plain=True, want_transposed=True):
    _require_gpu(w)
    lib = _lib.load()
    w = w.detach()
    if w.dtype != torch.float32:
        w = w.float()
    w = w.contiguous()
    E, R, C = w.shape
    pad = dtype == torch.bfloat16
    Cp, Rp = (-(-C // 64) * 64, -(-R // 64) * 64) if pad else (C, R)
    plain = torch.empty(E, R, Cp, device=w.device, dtype=dtype) if want_plain else None
    tr = torch.empty(E, C, Rp, device=w.device, dtype=dtype) if want_transposed else None
    code = _lib.BF16 if dtype == torch.bfloat16 else _lib.F32
    check(lib.apertis_cast_transpose(ptr(w), ptr(plain), ptr(tr), E, R, C, Cp, Rp, code, stream_ptr()),
          "apertis_cast_transpose")
    return plain, tr


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
        dblk = torch.empty(T, H, device=y.device, dtype=odt)
        nw = lib.apertis_layernorm_bwd_blocks(T, H)
        part = torch.empty(nw, 2, H, device=y.device, dtype=torch.float32)
        dg = torch.empty(H, device=y.device, dtype=torch.float32)
        db = torch.empty(H, device=y.device, dtype=torch.float32)
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
FUSED_ROUTER_BWD_CALLS = 0        # (times the one-pass form ran: the tests look at it)
FUSE_ACT_BWD = not _os.environ.get("APERTIS_NO_FUSE_ACT_BWD")
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
                 act_code | (_lib.ACT_SAVE_GRAD if saved_grad else 0), float(drop_p), int(seed), code, code, stream_ptr()),
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
                 max_rows, E, ptr(ws), ws_bytes, code, int(GEMM_DYNAMIC_QUEUE and TN_DYNAMIC_QUEUE), stream_ptr()), _RowsWork(offsets, E, 4.0 * I * H))
        return dxg, dw1.to(w1dt), db1, dw2.to(w2dt), db2, None, None, None, None, None, None


def expert_mlp(xg, w1, b1, w2, b2, offsets, max_rows, act="gelu", drop_p=0.0, seed=0, compute_dtype=None):
    """Grouped expert MLP (reference core.py:437-440): Linear(H->I) -> act -> Dropout -> Linear(I->H) for
    expert-sorted rows xg [R,H]; w1 [E,I,H], b1 [E,I], w2 [E,H,I], b2 [E,H] (fp32 masters)."""
    return _apply(_ExpertMLP, xg, w1, b1, w2, b2, offsets, max_rows, act, drop_p, seed, compute_dtype or xg.dtype)


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
        dw = torch.zeros(V, H, device=dev, dtype=torch.float32) if need else None
        nb = max(1, _LCE_CHUNK_ROWS // L)                                     # sequences per chunk
        lse = torch.empty(nb * L, device=dev, dtype=torch.float32)
        row_loss = torch.empty(nb * L, device=dev, dtype=torch.float32)
        for b0 in range(0, B, nb):
            n = min(nb, B - b0)
            xb = x[b0:b0 + n].reshape(n * L, H)
            logits = torch.matmul(xb, w.t()).reshape(n, L, V)                 # n sequences of logits
            lab = labels[b0:b0 + n]
            check(lib.apertis_cross_entropy_fwd(ptr(logits), ptr(lab), ptr(lse), ptr(row_loss), n, L, V, labels.shape[1], n_pos,
                                                ignore_index, code, stream_ptr()), "apertis_cross_entropy_fwd")
            loss_sum += row_loss[:n * L].sum()
            if need:
                check(lib.apertis_cross_entropy_bwd(ptr(logits), ptr(lab), ptr(lse), ptr(gscale), ptr(logits), n, L, V,
                                                    labels.shape[1], n_pos, ignore_index, code, stream_ptr()),
                      "apertis_cross_entropy_bwd")
                dl = logits.reshape(n * L, V)
                torch.matmul(dl, w, out=dx[b0:b0 + n].reshape(n * L, H))
                dw.add_(torch.matmul(dl.t(), xb))
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
