"""Shared plumbing of the operators: kernel timers, launch helpers, device checks, the column-slot / rows-gradient hand-over objects.

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import threading

import torch

from .._lib import ApertisHipError, check, dtype_code, ptr, stream_ptr


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


def _indexed(device):
    """torch.device with an explicit index ('cuda' -> the current device): the workspace tables are keyed on it."""
    d = torch.device(device)
    return torch.device("cuda", torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d


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
