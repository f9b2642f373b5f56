"""Data parallelism for the Apertis hot path: one process per GPU, gradients averaged with
bucketed all-reduce (RCCL over xGMI on MI355X; `backend="nccl"` is RCCL on ROCm) on a side HIP
stream, overlapped with the rest of backward.

Replaces torch DistributedDataParallel as the reference uses it (src/training/pipeline.py:463:
DDP(find_unused_parameters=False), gradient MEAN over ranks, default 25 MB buckets).  Choices
made for point-to-point xGMI (7 links x ~153 GB/s per GPU, ring collectives are per-link bound):
  - large buckets (default 128 MiB) so each all-reduce is bandwidth- not latency-bound;
  - gradients end up INSIDE the flat bucket buffers: the post-accumulate hook moves a fresh gradient into its
    slice (one read + one write) and makes param.grad that view, so the optimizer and further accumulation
    micro-steps work on the bucket in place and a bucket is ready the moment its last gradient has arrived.
    Between optimizer steps the gradients are None (zero_grad drops the views instead of zeroing 4 bytes per
    parameter, and autograd assigns instead of accumulating into zeros: two passes over the gradients fewer);
  - buckets are filled in reverse parameter order = the order backward produces gradients;
  - the collective is issued on a dedicated stream behind an event recorded on the compute
    stream; `finish()` makes the compute stream wait before clipping / the optimizer;
  - optional bf16 wire format (halves bytes on the links) with fp32 accumulation buffers.
MoE capacity and auxiliary losses are rank-local, exactly as under DDP: N-rank DP equals the mean
of independently computed per-shard gradients.
"""
import contextlib
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "work", "wire", "late_dst", "late_src")

    def __init__(self, flat, params, offsets):
        self.flat, self.params, self.offsets = flat, params, offsets
        self.pending = len(params)
        self.work = None
        self.wire = None
        self.late_dst, self.late_src = [], []   # gradients autograd produced outside the bucket: copied in ONE pass (flush)

    def flush(self):
        """Move the gradients that did not land in the bucket by themselves into their slices - one multi-tensor copy per
        bucket instead of one copy launch per parameter (the 1.5B model has ~1 100 small parameters - norms, biases, conv and
        dt weights: 1 246 `copyBuffer` launches, 6 ms of the forced-DP step in profiles/r6_forced_dp_n1_kernel_stats_before.csv)."""
        if self.late_dst:
            torch._foreach_copy_(self.late_dst, self.late_src)
            self.late_dst, self.late_src = [], []


class BucketedDataParallel(nn.Module):
    def __init__(self, module: nn.Module, bucket_bytes: int = 128 << 20, process_group=None,
                 reduce_dtype: Optional[torch.dtype] = None, broadcast_parameters: bool = True):
        super().__init__()
        self.module = module
        self.group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.reduce_dtype = reduce_dtype
        self._sync = True
        self.copied_bytes = 0     # gradient bytes the hooks had to move into the buckets (diagnostic)
        self.reduced_bytes = 0    # bytes handed to all_reduce so far, in the wire dtype (diagnostic; bench.py reports it)
        self._warned_partial = False
        self._flush_queued = False
        import os as _os
        self._force = _os.environ.get("APERTIS_FORCE_DP") == "1"
        if self.world_size > 1 or self._force:
            # collectives will share the GPU with the persistent GEMM kernels: let those take their tiles from a queue,
            # so a work-group whose CU an RCCL kernel holds does not walk a full static share alone at the end
            # (APERTIS_FORCE_DP=1 - the N > 1 step rehearsed on one rank - runs the very kernels the N > 1 step runs)
            from . import ops as _ops
            _ops.GEMM_DYNAMIC_QUEUE = _os.environ.get("APERTIS_DP_STATIC_WALK") != "1"   # (=1: A/B switch, static walks under DP)
        params = [p for p in module.parameters() if p.requires_grad]
        self.device = params[0].device
        self._cuda = self.device.type == "cuda"
        self.comm_stream = torch.cuda.Stream(device=self.device) if self._cuda else None
        self._avg_op = bool(self._cuda and dist.is_initialized() and dist.get_backend(process_group) == "nccl"
                            and not _os.environ.get("APERTIS_DP_NO_AVG"))
        if broadcast_parameters and self.world_size > 1:
            # DDP semantics: every replica starts from rank 0's parameters and buffers
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                                   group=process_group)
        self.buckets: List[_Bucket] = []
        self._slot = {}
        self._build_buckets(list(reversed(params)), bucket_bytes)

    # ------------------------------------------------------------------------------------------
    def _build_buckets(self, params, bucket_bytes):
        cur, cur_bytes = [], 0
        groups = []
        for p in params:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        for g in groups:
            total = sum(-(-p.numel() // 64) * 64 for p in g)          # 256-byte aligned slices (fp32)
            flat = torch.zeros(total, device=self.device, dtype=g[0].dtype)
            offsets, off = [], 0
            for p in g:
                offsets.append(off)
                p.grad = None
                # backward kernels that can write their weight gradient anywhere (ops.grad_destination) write it here
                p._apertis_grad_view = flat[off:off + p.numel()].view_as(p)
                off += -(-p.numel() // 64) * 64
            b = _Bucket(flat, g, offsets)
            bi = len(self.buckets)
            self.buckets.append(b)
            for i, p in enumerate(g):
                self._slot[id(p)] = i
                p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _make_hook(self, bi):
        def hook(p):
            b = self.buckets[bi]
            i = self._slot[id(p)]
            view = b.flat[b.offsets[i]:b.offsets[i] + p.numel()].view_as(p)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                # autograd replaced the gradient tensor: fold it back into the bucket (the data moves when the bucket is
                # complete - _Bucket.flush; nothing reads p.grad before the reduction / the optimizer)
                if p.grad is not None:
                    b.late_dst.append(view)
                    b.late_src.append(p.grad.detach())
                    self.copied_bytes += p.numel() * p.element_size()
                    if not self._flush_queued:
                        # a bucket that never completes in this backward (a parameter without a gradient) must still hold its
                        # data before the NEXT backward accumulates into the views: flush whatever is left when this one ends
                        self._flush_queued = True
                        torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                p.grad = view
            b.pending -= 1
            if b.pending == 0:
                b.flush()
                if self._sync and (self.world_size > 1 or self._force):
                    self._launch(b)
        return hook

    def _end_of_backward(self):
        self._flush_queued = False
        for b in self.buckets:
            b.flush()

    def _launch(self, b: _Bucket):
        inv = 1.0 / self.world_size
        wire_dt = self.reduce_dtype if (self._cuda and self.reduce_dtype is not None) else b.flat.dtype
        self.reduced_bytes += b.flat.numel() * torch.empty((), dtype=wire_dt).element_size()
        if self._cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                # RCCL averages inside the collective (ReduceOp.AVG): no separate 1/world pass over the bucket
                avg = self._avg_op
                if self.reduce_dtype is not None and self.reduce_dtype != b.flat.dtype:
                    b.wire = b.flat.to(self.reduce_dtype)
                    dist.all_reduce(b.wire, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group)
                    b.flat.copy_(b.wire)
                    if not avg:
                        b.flat.mul_(inv)
                    b.wire.record_stream(self.comm_stream)
                else:
                    dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group)
                    if not avg:
                        b.flat.mul_(inv)
        else:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    # ------------------------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def finish(self):
        """Call after backward(), before clipping / optimizer.step(): waits for the reductions."""
        for b in self.buckets:
            b.flush()          # (a bucket some parameter of which got no gradient never reached pending == 0)
        if (self.world_size > 1 or self._force) and self._sync:
            for b in self.buckets:
                if b.pending != 0:
                    # some parameter of this bucket received no gradient this step (e.g. the vision tower
                    # on a text-only batch): its slice still holds an earlier step's values - reduce zeros instead
                    for i, p in enumerate(b.params):
                        if p.grad is None:
                            b.flat[b.offsets[i]:b.offsets[i] + p.numel()].zero_()
                    if b.pending != len(b.params) and not self._warned_partial:
                        self._warned_partial = True
                        import warnings
                        warnings.warn("BucketedDataParallel: a bucket was reduced with parameters that got no "
                                      "gradient this step (their gradients are zeros)")
                    self._launch(b)
            if self._cuda:
                torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
            else:
                for b in self.buckets:
                    if b.work is not None:
                        b.work.wait()
                        b.flat.mul_(1.0 / self.world_size)
                        b.work = None
        for b in self.buckets:
            b.pending = len(b.params)

    def zero_grad(self, set_to_none: bool = True):
        """Drop the gradients (the next backward assigns fresh ones, which the hooks move into the buckets).
        set_to_none=False keeps the views and zeroes the flat buffers instead."""
        for b in self.buckets:
            if set_to_none:
                for p in b.params:
                    p.grad = None
            else:
                b.flat.zero_()

    @contextlib.contextmanager
    def no_sync(self):
        """Skip the all-reduce for this backward (gradient-accumulation micro-steps)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old
            for b in self.buckets:
                b.pending = len(b.params)

    def gradient_bytes(self) -> int:
        return sum(b.flat.numel() * b.flat.element_size() for b in self.buckets)
