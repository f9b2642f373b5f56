"""The data-parallel training step of the hot path (what bench.py times).

Mirrors the step of the reference's ApertisTrainer (src/training/pipeline.py:469-551): AdamW with
two parameter groups (no weight decay on names containing 'bias' / 'LayerNorm.*'), OneCycleLR
(pct_start 0.1, cos, div 25, final_div 1e4), autocast, gradient clipping at max_grad_norm, one
optimizer step per `gradient_accumulation_steps` micro-batches.  Deviations, both stated in
DESIGN.md: autocast dtype is bf16 (the reference's fp16 + GradScaler is a CUDA-era choice; bf16
needs no loss scaling), and gradients are reduced once per optimizer step (DDP in the reference
reduces on every micro-step; the results are identical).
"""
import math
import os
from typing import Optional

import torch
import torch.distributed as dist

from .parallel import BucketedDataParallel

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def build_optimizer(model, lr: float = 5e-5, weight_decay: float = 0.01, fused: Optional[bool] = None):
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": weight_decay},
              {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]
    on_gpu = all(p.is_cuda and p.dtype == torch.float32 for _, p in named)
    if fused is None and on_gpu and not os.environ.get("APERTIS_TORCH_ADAMW"):
        return ApertisAdamW(groups, lr=lr)             # clip + AdamW on the HIP kernels (csrc/optimizer.hip)
    if fused is None:
        fused = named[0][1].is_cuda
    return torch.optim.AdamW(groups, lr=lr, fused=fused)


def clip_and_step(optimizer, params, max_grad_norm):
    """clip_grad_norm_ + optimizer.step() (pipeline.py:544-546); one fused call on ApertisAdamW."""
    if isinstance(optimizer, ApertisAdamW):
        optimizer.step(max_grad_norm=max_grad_norm)
    else:
        torch.nn.utils.clip_grad_norm_(params, max_grad_norm, foreach=True if params and params[0].is_cuda else None)
        optimizer.step()


class ApertisAdamW(torch.optim.Optimizer):
    """AdamW with the gradient-norm clip folded in, on the HIP kernels `apertis_grad_sumsq` / `apertis_clip_coef` /
    `apertis_adamw_step` (csrc/optimizer.hip): what the reference does as clip_grad_norm_ + optimizer.step
    (pipeline.py:544-546) in two passes over the gradients.  Same update rule, parameter-group options and state keys
    (`step`, `exp_avg`, `exp_avg_sq`) as torch.optim.AdamW, so LR schedulers and optimizer state dicts carry over.
    `step(max_grad_norm=...)` clips by the global norm over ALL groups first; `last_grad_norm` (a device scalar) holds
    that norm afterwards.  fp32 CUDA parameters only - anything else raises.

    A poisoned step: if the fused scan's look-back time-out word (`ops.scan_gate_error`) is set when `step` runs, the norm
    comes out NaN and the update kernel returns without touching parameters or moments - but the host-side `step` counts
    (bias correction) and any LR scheduler the caller steps still advance, and the word is STICKY: every later step is
    skipped the same way until the caller clears it (`ops.scan_gate_clear_error()`).  `ApertisTrainer` raises behind its
    `loss.item()` (`ops.scan_gate_raise_on_error()`), `TrainStep` returns a NaN loss; a bare user of this optimizer should
    poll `ops.scan_gate_error()` as well, or check `last_grad_norm` for NaN."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = None
        self.last_grad_norm = None

    def load_state_dict(self, state_dict):
        """The loaded moments are new tensors: the cached device tables point at the old ones."""
        super().load_state_dict(state_dict)
        self._tables = None

    def _build_tables(self):
        """Device tables of the three kernels: per launch (= parameters of one group that share a step count) a record
        array `{p, g, m, v, numel}` and the chunk -> (tensor, chunk) maps.  Everything but the gradient addresses is fixed
        between optimizer steps, so the tables are built ONCE; `_refresh` only rewrites the gradient column when autograd
        handed out other buffers, through pinned staging and a non-blocking copy (the first form rebuilt the tables from
        pageable numpy arrays on every step: a synchronous copy, i.e. a full host <-> GPU sync per step, and 7-20 ms of
        Python on the 76-layer configuration)."""
        from . import _lib
        import numpy as np
        lib = _lib.load()
        chunk = int(lib.apertis_opt_chunk_elems())
        launches, recs, cts, cis, plist, n_total = [], [], [], [], [], 0
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _lib.ApertisHipError("ApertisAdamW needs contiguous fp32 CUDA parameters and gradients")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif st["step"].is_cuda:         # a state dict saved by torch's fused / capturable AdamW
                    st["step"] = st["step"].detach().cpu()
            # torch.optim.AdamW keeps `step` per parameter (bias correction of a parameter that first gets a gradient
            # late, e.g. the vision tower after text-only steps, starts at 1): one launch per distinct step value
            by_step = {}
            for p in ps:
                by_step.setdefault(int(self.state[p]["step"].item()), []).append(p)
            for step0, sub in sorted(by_step.items()):
                # the step counts of a launch's parameters as views of ONE host tensor: a step is one add (a
                # `torch._foreach_add_` over the 1 300 separate host tensors of the 76-layer configuration was 1.4 ms of host
                # time per group and step); `optimizer.state[p]["step"]` stays current and 0-dim as torch's AdamW keeps it
                steps_all = torch.full((len(sub),), float(step0))
                for i, p in enumerate(sub):
                    self.state[p]["step"] = steps_all[i]
                rec = np.zeros((len(sub), 5), dtype=np.int64)
                ct, ci = [], []
                for i, p in enumerate(sub):
                    st = self.state[p]
                    rec[i] = (p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
                    nc = -(-p.numel() // chunk)
                    ct.append(np.full(nc, i, dtype=np.int32))
                    ci.append(np.arange(nc, dtype=np.int32))
                ct, ci = np.concatenate(ct), np.concatenate(ci)
                launches.append(dict(group=gi, params=sub, steps=[self.state[p]["step"] for p in sub], steps_all=steps_all, step0=step0,
                                     n=len(ct), first=n_total, row0=len(plist), chunk0=sum(len(c) for c in cts)))
                recs.append(rec), cts.append(ct), cis.append(ci)
                plist.extend(sub)
                n_total += len(ct)
        if not plist:
            self._tables = dict(launches=[], plist=[], pp=[], mv=[], partials=None,
                                none=[p for g in self.param_groups for p in g["params"]])
            return
        dev = plist[0].device
        if any(p.device != dev for p in plist):
            raise _lib.ApertisHipError("ApertisAdamW: parameters on more than one device")
        rec = np.concatenate(recs)
        pins = [torch.from_numpy(rec.copy()).pin_memory() for _ in range(2)]       # staging, used alternately
        T = dict(launches=launches, plist=plist, n_total=n_total,
                 none=[p for g in self.param_groups for p in g["params"] if p.grad is None],
                 pp=[p.data_ptr() for p in plist],
                 mv=[(self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in plist],
                 gp=None, pins=pins, pin_np=[t.numpy() for t in pins], pin_ev=[None, None], turn=0,
                 rec=torch.empty(rec.shape, dtype=torch.int64, device=dev),
                 ct=torch.from_numpy(np.concatenate(cts)).to(dev), ci=torch.from_numpy(np.concatenate(cis)).to(dev),
                 partials=torch.empty(max(n_total, 1), device=dev, dtype=torch.float32),
                 norm_coef=torch.zeros(2, device=dev, dtype=torch.float32))
        for g in launches:                                   # each launch's slices of the shared device tables
            g["rec"] = T["rec"][g["row0"]:]
            g["ct"], g["ci"] = T["ct"][g["chunk0"]:], T["ci"][g["chunk0"]:]
        self._tables = T

    def _tables_current(self):
        """The fixed part of the tables still describes the optimizer: the same parameters have gradients, parameters
        and moments live where they did, and every launch's parameters still share their step count."""
        T = self._tables
        if T is None:
            return False
        state = self.state
        try:
            if any(p.grad is None for p in T["plist"]) or any(p.grad is not None for p in T["none"]):
                return False
            if [p.data_ptr() for p in T["plist"]] != T["pp"]:
                return False
            for p, (m, v) in zip(T["plist"], T["mv"]):
                st = state[p]
                if st["exp_avg"] is not m or st["exp_avg_sq"] is not v:
                    return False
            for g in T["launches"]:
                if any(state[p]["step"] is not s for p, s in zip(g["params"], g["steps"])):
                    return False
                # ... and still COUNT what the kernel's bias correction assumes (an in-place `state['step'].zero_()` keeps
                # the tensor's identity); host tensors only - reading a device step count would be a sync
                s0 = g["steps"][0] if g["steps"] else None
                if s0 is not None and not s0.is_cuda and int(s0.item()) != g["step0"]:
                    return False
        except KeyError:
            return False
        n_groups = sum(len(g["params"]) for g in self.param_groups)
        return n_groups == len(T["plist"]) + len(T["none"])

    def _refresh(self):
        """Gradient addresses into the device tables (only when they moved since the last step)."""
        from . import _lib
        T = self._tables
        grads = [p.grad for p in T["plist"]]
        for g, p in zip(grads, T["plist"]):        # (every step: a gradient may change dtype / layout at an unchanged address)
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                raise _lib.ApertisHipError("ApertisAdamW needs contiguous fp32 CUDA parameters and gradients")
        gp = [g.data_ptr() for g in grads]
        if gp == T["gp"]:
            return
        i = T["turn"] = T["turn"] ^ 1
        if T["pin_ev"][i] is not None:
            T["pin_ev"][i].synchronize()          # the copy issued from this staging buffer two refreshes ago has run
        T["pin_np"][i][:, 1] = gp
        T["rec"].copy_(T["pins"][i], non_blocking=True)
        ev = T["pin_ev"][i] = T["pin_ev"][i] or torch.cuda.Event()
        ev.record()
        T["gp"] = gp

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        from . import _lib, ops
        from ._lib import check, ptr, stream_ptr
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._tables_current():       # moments were replaced, a parameter joined or moved, or first call
            self._build_tables()
        T = self._tables
        if T["partials"] is None:
            return loss
        lib = _lib.load()
        coef = None
        with torch.cuda.device(T["partials"].device):
            self._refresh()
            if max_grad_norm is not None:
                for g in T["launches"]:
                    check(lib.apertis_grad_sumsq(ptr(g["rec"]), ptr(g["ct"]), ptr(g["ci"]), g["n"],
                                                 ptr(T["partials"][g["first"]:]), stream_ptr()), "apertis_grad_sumsq")
                # the scan's look-back time-out word poisons the step (NaN norm, the AdamW pass skipped): no sync
                poison = ops.scan_gate_error_word(T["partials"].device)
                check(lib.apertis_clip_coef(ptr(T["partials"]), T["n_total"], float(max_grad_norm), ptr(T["norm_coef"]),
                                            ptr(poison), stream_ptr()), "apertis_clip_coef")
                coef = T["norm_coef"]
                self.last_grad_norm = coef[0]
            for g in T["launches"]:
                group = self.param_groups[g["group"]]
                g["step0"] += 1
                g["steps_all"].add_(1)                       # the per-parameter `step` tensors of torch.optim.AdamW's state: views
                b1, b2 = group["betas"]
                check(lib.apertis_adamw_step(ptr(g["rec"]), ptr(g["ct"]), ptr(g["ci"]), g["n"], float(group["lr"]), float(b1),
                                             float(b2), float(group["eps"]), float(group["weight_decay"]), g["step0"], ptr(coef),
                                             stream_ptr()), "apertis_adamw_step")
        ops.note_weights_changed()      # (the kernels write the parameters through raw pointers: no version bump)
        return loss


def build_train_prep(model):
    """ops.TrainPrep over every module of `model` that offers `register_train_prep` (SSM blocks, expert systems, dense FFNs), or
    None (CPU model, APERTIS_TRAIN_PREP=0, nothing to register)."""
    from . import ops
    p0 = next((p for p in model.parameters()), None)
    if not ops.TRAIN_PREP or p0 is None or not p0.is_cuda:
        return None
    prep = ops.TrainPrep(p0.device)
    for m in model.modules():
        reg = getattr(m, "register_train_prep", None)
        if callable(reg):
            reg(prep)
    prep.finalize()
    return prep if prep.n_records else None


class prepared_step:
    """`with prepared_step(prep):` around a training micro-step's forward + backward: refreshes the prepared weight copies (one
    launch) and makes them visible to the ops; a no-op for prep = None."""

    def __init__(self, prep):
        self.prep, self.scope = prep, None

    def __enter__(self):
        if self.prep is not None:
            self.prep.refresh()
            self.scope = self.prep.active()
            self.scope.__enter__()
        return self

    def __exit__(self, *exc):
        if self.scope is not None:
            self.scope.__exit__(*exc)
        return False


class TrainStep:
    """loss = model(**batch)[0]; backward; (all-reduce); clip; AdamW; OneCycleLR."""

    def __init__(self, model, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0, total_steps=1000, bf16=True,
                 gradient_accumulation_steps=1, bucket_bytes=128 << 20, reduce_dtype=None, fused_lm_head_loss=True):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        force = os.environ.get("APERTIS_FORCE_DP") == "1" and dist.is_initialized()   # exercise the DP path at world 1
        self.dp = (BucketedDataParallel(model, bucket_bytes=bucket_bytes, reduce_dtype=reduce_dtype)
                   if (self.world > 1 or force) else None)
        self.model = model
        if hasattr(model, "fused_lm_head_loss"):
            model.fused_lm_head_loss = bool(fused_lm_head_loss)   # the step reads only the loss: LM head + CE without the logits tensor
        self.optimizer = build_optimizer(model, lr, weight_decay)
        total_steps = max(int(total_steps), 2)
        if abs(0.1 * total_steps - 1.0) < 1e-9:      # OneCycleLR divides by (pct_start*total - 1): avoid the 0/0 case
            total_steps += 1
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(self.optimizer, max_lr=lr, total_steps=total_steps,
                                                             pct_start=0.1, anneal_strategy="cos", div_factor=25.0,
                                                             final_div_factor=10000.0)
        self.max_grad_norm = max_grad_norm
        self.bf16 = bf16
        self.accum = max(1, gradient_accumulation_steps)
        self._micro = 0
        self._params = [p for p in model.parameters() if p.requires_grad]
        self.prep = build_train_prep(model) if bf16 else None     # (bf16 compute copies only)

    def __call__(self, **batch):
        """One micro-batch.  Returns the detached loss tensor (no host sync)."""
        self._micro += 1
        last = self._micro % self.accum == 0
        ctx = self.dp.no_sync() if (self.dp is not None and not last) else _null()
        with ctx, prepared_step(self.prep):
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.bf16):
                loss = self.model(**batch)[0]
            (loss / self.accum).backward()
        if last:
            if self.dp is not None:
                self.dp.finish()
            clip_and_step(self.optimizer, self._params, self.max_grad_norm)
            self.scheduler.step()
            if self.dp is not None:
                self.dp.zero_grad()
            else:
                self.optimizer.zero_grad(set_to_none=True)
        loss = loss.detach()
        if loss.is_cuda:
            from . import ops
            word = ops.scan_gate_error_word(loss.device)      # a timed-out scan look-back makes the loss NaN (no host sync)
            if word is not None:
                loss = torch.where(word[0] != 0, torch.full_like(loss, float("nan")), loss)
        return loss


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
