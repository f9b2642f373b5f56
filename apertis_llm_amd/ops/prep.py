"""Prepared copies of the weights: the per-call cache, the one-launch training-step preparation (TrainPrep), cast / transpose.

Part of apertis_llm_amd.ops (split by subsystem in round 6; `from apertis_llm_amd import ops` exposes every name as before).
torch is used for device memory, streams and autograd bookkeeping only; every computation is a HIP kernel launch through
apertis_llm_amd._lib.  Tensors must live on a ROCm device.
"""
import os as _os

import torch

from .. import _lib
from .._lib import ApertisHipError, check, ptr, stream_ptr
from ._base import _indexed, _require_gpu


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
