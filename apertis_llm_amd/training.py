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
    if fused is None:
        fused = named[0][1].is_cuda
    return torch.optim.AdamW(groups, lr=lr, fused=fused)


class TrainStep:
    """loss = model(**batch)[0]; backward; (all-reduce); clip; AdamW; OneCycleLR."""

    def __init__(self, model, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0, total_steps=1000, bf16=True,
                 gradient_accumulation_steps=1, bucket_bytes=128 << 20, reduce_dtype=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        force = os.environ.get("APERTIS_FORCE_DP") == "1" and dist.is_initialized()   # exercise the DP path at world 1
        self.dp = (BucketedDataParallel(model, bucket_bytes=bucket_bytes, reduce_dtype=reduce_dtype)
                   if (self.world > 1 or force) else None)
        self.model = model
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
            torch.nn.utils.clip_grad_norm_(self._params, self.max_grad_norm, foreach=True)
            self.optimizer.step()
            self.scheduler.step()
            if self.dp is not None:
                self.dp.zero_grad()
            else:
                self.optimizer.zero_grad(set_to_none=True)
        return loss.detach()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
