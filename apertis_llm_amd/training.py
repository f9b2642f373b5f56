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
    that norm afterwards.  fp32 CUDA parameters only - anything else raises."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = None
        self.last_grad_norm = None

    def load_state_dict(self, state_dict):
        """The loaded moments are new tensors: the cached device tables point at the old ones."""
        super().load_state_dict(state_dict)
        self._tables = None

    def _key(self):
        """Everything the cached device tables depend on: parameter, gradient and both moment addresses, and the
        per-parameter step (parameters that share a step share a launch)."""
        key = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.get(p) or {}
                m, v, stp = st.get("exp_avg"), st.get("exp_avg_sq"), st.get("step")
                key.append((p.data_ptr(), p.grad.data_ptr(), None if m is None else m.data_ptr(),
                            None if v is None else v.data_ptr(), None if stp is None else id(stp)))
        return key

    def _build_tables(self):
        from . import _lib
        import numpy as np
        lib = _lib.load()
        chunk = int(lib.apertis_opt_chunk_elems())
        tables, n_total = [], 0
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            for p in ps:
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous()
                        and p.grad.is_contiguous()):
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
                rec = np.zeros((len(sub), 5), dtype=np.int64)
                ct, ci = [], []
                for i, p in enumerate(sub):
                    st = self.state[p]
                    rec[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                              p.numel())
                    nc = -(-p.numel() // chunk)
                    ct.append(np.full(nc, i, dtype=np.int32))
                    ci.append(np.arange(nc, dtype=np.int32))
                dev = sub[0].device
                ct, ci = np.concatenate(ct), np.concatenate(ci)
                tables.append(dict(group=gi, params=sub, n=len(ct), first=n_total,
                                   rec=torch.from_numpy(rec.view(np.uint8).reshape(-1)).to(dev),
                                   ct=torch.from_numpy(ct).to(dev), ci=torch.from_numpy(ci).to(dev)))
                n_total += len(ct)
        dev = next((t["params"][0].device for t in tables), None)
        self._tables = dict(groups=tables, key=self._key(), n_total=n_total,
                            partials=torch.empty(max(n_total, 1), device=dev, dtype=torch.float32) if dev is not None else None,
                            norm_coef=torch.zeros(2, device=dev, dtype=torch.float32) if dev is not None else None)

    def _tables_current(self):
        return self._tables is not None and self._key() == self._tables["key"]

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None):
        from . import _lib, ops
        from ._lib import check, ptr, stream_ptr
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not self._tables_current():       # gradients / moments were reallocated, a parameter joined, or first call
            self._build_tables()
        T = self._tables
        if T["partials"] is None:
            return loss
        lib = _lib.load()
        coef = None
        with torch.cuda.device(T["partials"].device):
            if max_grad_norm is not None:
                for g in T["groups"]:
                    check(lib.apertis_grad_sumsq(ptr(g["rec"]), ptr(g["ct"]), ptr(g["ci"]), g["n"],
                                                 ptr(T["partials"][g["first"]:]), stream_ptr()), "apertis_grad_sumsq")
                # the scan's look-back time-out word poisons the coefficient (NaN step, like a non-finite norm): no sync
                poison = ops.scan_gate_error_word(T["partials"].device)
                check(lib.apertis_clip_coef(ptr(T["partials"]), T["n_total"], float(max_grad_norm), ptr(T["norm_coef"]),
                                            ptr(poison), stream_ptr()), "apertis_clip_coef")
                coef = T["norm_coef"]
                self.last_grad_norm = coef[0]
            for g in T["groups"]:
                group = self.param_groups[g["group"]]
                step = int(self.state[g["params"][0]]["step"].item()) + 1
                for p in g["params"]:
                    self.state[p]["step"] += 1
                b1, b2 = group["betas"]
                check(lib.apertis_adamw_step(ptr(g["rec"]), ptr(g["ct"]), ptr(g["ci"]), g["n"], float(group["lr"]), float(b1),
                                             float(b2), float(group["eps"]), float(group["weight_decay"]), step, ptr(coef),
                                             stream_ptr()), "apertis_adamw_step")
        ops.note_weights_changed()      # (the kernels write the parameters through raw pointers: no version bump)
        return loss


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

    def __call__(self, **batch):
        """One micro-batch.  Returns the detached loss tensor (no host sync)."""
        self._micro += 1
        last = self._micro % self.accum == 0
        ctx = self.dp.no_sync() if (self.dp is not None and not last) else _null()
        with ctx:
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
