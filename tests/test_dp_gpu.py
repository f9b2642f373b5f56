"""Data-parallel path on the real HIP kernels: two ranks (gloo rendezvous, both on cuda:0 - the test box has one
card; the bench uses RCCL with one card per rank) run the SSM + MoE model through `BucketedDataParallel` and
`TrainStep`.  What only a world_size > 1 run on the GPU exercises: expert weight gradients written by the TN kernel
straight into their bucket slices (ops.grad_destination), the hook path for every other gradient, the side stream,
the dynamic tile queue of the persistent NT kernel (switched on by the wrapper when world_size > 1).
Oracle for DP: the mean of the independently computed per-shard gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
           attention_type="selective_ssm", use_expert_system=True, hidden_dropout_prob=0.0,
           attention_probs_dropout_prob=0.0, use_noisy_top_k_routing=False, use_expert_dropout=False)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(rank, dev):
    g = torch.Generator().manual_seed(1000 + rank)
    ids = torch.randint(4, 512, (2, 256), generator=g).to(dev)
    return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # (the box's hostname may not resolve: gloo's pair connections over loopback)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import apertis_llm_amd as A
        from apertis_llm_amd.parallel import BucketedDataParallel
        from apertis_llm_amd.training import TrainStep
        dev = torch.device("cuda:0")
        torch.manual_seed(0)
        init = A.ApertisForCausalLM(A.ApertisConfig(**CFG)).state_dict()
        batch = _batch(rank, dev)

        def fresh():
            m = A.ApertisForCausalLM(A.ApertisConfig(**CFG))
            m.load_state_dict(init)
            return m.to(dev).train()

        # (1) this rank's own shard gradient, no wrapper
        m0 = fresh()
        m0(**batch)[0].backward()
        local = [p.grad.detach().float().cpu().numpy().copy() for p in m0.parameters()]
        del m0
        # (2) the same backward under the wrapper: buckets, hooks, grad_destination, side stream
        m1 = fresh()
        dp = BucketedDataParallel(m1, bucket_bytes=256 << 10)
        from apertis_llm_amd import ops as _ops
        assert _ops.GEMM_DYNAMIC_QUEUE is True      # the persistent NT GEMM takes its tiles from the per-stream queue counter
        dp(**batch)[0].backward()
        dp.finish()
        torch.cuda.synchronize()
        reduced = [p.grad.detach().float().cpu().numpy().copy() for p in m1.parameters()]
        in_bucket = all(p.grad.data_ptr() == p._apertis_grad_view.data_ptr() for p in m1.parameters())
        nb, copied, total = len(dp.buckets), dp.copied_bytes, dp.gradient_bytes()
        del dp, m1
        # (3) three optimizer steps through TrainStep (bf16 autocast, clip + AdamW kernels on the bucket views)
        m2 = fresh()
        step = TrainStep(m2, lr=1e-3, total_steps=10, bf16=True, bucket_bytes=256 << 10)
        assert step.dp is not None
        g = torch.Generator().manual_seed(2000 + rank)
        losses = []
        for _ in range(3):
            ids = torch.randint(4, 512, (2, 256), generator=g).to(dev)
            losses.append(float(step(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids)))
        torch.cuda.synchronize()
        params = [p.detach().float().cpu().numpy().copy() for p in m2.parameters()]
        out.put((rank, local, reduced, in_bucket, nb, copied, total, losses, params))
    finally:
        dist.destroy_process_group()


def _probe_worker(rank, world, port, out):
    """Two gloo ranks on ONE card exchanging CUDA tensors - nothing of this repo in it: can the box do what the test needs?"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.full((4096,), float(rank + 1), device="cuda:0")
        dist.broadcast(x, src=0)
        y = torch.full((4096,), float(rank + 1), device="cuda:0")
        dist.all_reduce(y)
        torch.cuda.synchronize()
        out.put((rank, float(x[0]), float(y[0])))
    finally:
        dist.destroy_process_group()


def _run_ranks(target, world, timeout):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = None
    try:
        import queue
        try:
            res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda t: t[0])
        except queue.Empty:
            res = None
    finally:
        for p in procs:
            p.join(timeout=30 if res is None else 120)
            if p.is_alive():
                p.kill()
    return res, procs


def test_two_ranks_on_hip_kernels_reduce_to_mean_and_stay_identical(dev):
    import numpy as np
    world = 2
    # (seen once on this pool: for half an hour, on six boxes in a row, the two ranks of this test hung INSIDE gloo's broadcast /
    #  all_reduce of CUDA tensors - first in the wrapper's parameter broadcast, with no kernel of this repo in flight - and then
    #  not again on the same code.  A box that cannot pass the plain probe below cannot run this test: skip it, loudly, instead of
    #  hanging the suite; a time-out of the real workers on a box that DOES pass the probe is a failure)
    probe, _ = _run_ranks(_probe_worker, world, 90)
    if probe is None or [t[1:] for t in probe] != [(1.0, 3.0), (1.0, 3.0)]:
        # a skipped data-parallel test reads green in GPUTEST.  It is RED unless the operator allows the skip
        # explicitly (APERTIS_ALLOW_DP_SKIP=1: a box known to have the gloo problem above).
        msg = f"two gloo ranks on one card cannot exchange CUDA tensors on this box (probe: {probe})"
        if os.environ.get("APERTIS_ALLOW_DP_SKIP") == "1":
            pytest.skip(msg)
        pytest.fail(msg + " - set APERTIS_ALLOW_DP_SKIP=1 to skip the data-parallel GPU test on such a box")
    res, procs = _run_ranks(_worker, world, 420)
    assert res is not None, "the two data-parallel ranks did not finish within 420 s (the gloo probe in front of them did)"
    assert all(p.exitcode == 0 for p in procs)
    (_, l0, r0, ib0, nb0, cp0, tot0, loss0, p0), (_, l1, r1, ib1, nb1, cp1, tot1, loss1, p1) = res
    assert ib0 and ib1, "after finish() every gradient must live in its bucket slice"
    assert nb0 == nb1 and nb0 > 1
    # the expert weights (most of the bytes) must not have gone through a hook copy
    assert cp0 < 0.5 * tot0, f"{cp0} of {tot0} gradient bytes were copied by hooks"
    for i, (a, b, ra, rb) in enumerate(zip(l0, l1, r0, r1)):
        assert np.array_equal(ra, rb), f"parameter {i}: ranks hold different reduced gradients"
        ref = (a.astype(np.float64) + b.astype(np.float64)) / 2
        tol = 1e-5 * max(np.abs(ref).max(), 1e-30) + 1e-4 * np.abs(ref)     # float atomics at expert boundaries
        assert (np.abs(ra - ref) <= tol).all(), f"parameter {i}: reduced gradient is not the mean of the shard gradients " \
                                                f"(max abs diff {np.abs(ra - ref).max():.3e}, ref max {np.abs(ref).max():.3e})"
    assert all(np.isfinite(loss0)) and all(np.isfinite(loss1)) and loss0 != loss1
    for i, (a, b) in enumerate(zip(p0, p1)):
        assert np.array_equal(a, b), f"parameter {i}: replicas diverged after three steps"


def _nccl_worker(port, out):
    """One rank on the RCCL backend (`nccl` on ROCm) with the wrapper forced on: the collective itself is
    ReduceOp.AVG on the side stream - what every rank of the multi-GPU run executes per bucket."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", APERTIS_FORCE_DP="1",
                      RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        import apertis_llm_amd as A
        from apertis_llm_amd.parallel import BucketedDataParallel
        torch.manual_seed(0)
        init = A.ApertisForCausalLM(A.ApertisConfig(**CFG)).state_dict()
        batch = _batch(0, dev)

        def fresh():
            m = A.ApertisForCausalLM(A.ApertisConfig(**CFG))
            m.load_state_dict(init)
            return m.to(dev).train()
        m0 = fresh()
        m0(**batch)[0].backward()
        local = [p.grad.detach().float().cpu() for p in m0.parameters()]
        m1 = fresh()
        dp = BucketedDataParallel(m1, bucket_bytes=256 << 10)
        assert dp._avg_op and dp._force and dist.get_backend() == "nccl"
        dp(**batch)[0].backward()
        dp.finish()
        torch.cuda.synchronize()
        red = [p.grad.detach().float().cpu() for p in m1.parameters()]
        names = [n for n, _ in m1.named_parameters()]
        bad = [(n, float((a - b).abs().max()), float(a.abs().max())) for n, a, b in zip(names, local, red)
               if not torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max()) + 1e-30)]
        ok = not bad
        if bad:
            print("RCCL AVG vs plain backward, parameters outside tolerance (name, max abs diff, max abs):", bad, flush=True)
        res = (ok, dp.reduced_bytes, dp.gradient_bytes(), len(dp.buckets))
        del dp, m0, m1
        # (2) the step the multi-GPU bench runs, at a size where the PERSISTENT kernels and BOTH item queues are in play
        # (>= 4096 routed rows for the NT tile queue, >= 2048 rows per expert for the weight-gradient queue): three
        # TrainSteps with the RCCL all-reduce on the side stream, against three plain steps - bit-identical losses and weights
        from apertis_llm_amd import ops
        from apertis_llm_amd.training import TrainStep
        big = dict(CFG, hidden_size=512, num_attention_heads=4, intermediate_size=1024, num_hidden_layers=2, num_experts=2,
                   experts_per_token=2)
        torch.manual_seed(1)
        init2 = A.ApertisForCausalLM(A.ApertisConfig(**big)).state_dict()
        def three_steps(force):
            os.environ["APERTIS_FORCE_DP"] = "1" if force else "0"
            ops.GEMM_DYNAMIC_QUEUE = bool(force)          # what BucketedDataParallel switches on when world_size > 1
            ops.TN_DYNAMIC_QUEUE = bool(force)            # (opt-in since round 4: exercised here so that the path stays tested)
            m = A.ApertisForCausalLM(A.ApertisConfig(**big))
            m.load_state_dict(init2)
            m = m.to(dev).train()
            step = TrainStep(m, lr=1e-3, total_steps=10, bf16=True, bucket_bytes=4 << 20)
            assert (step.dp is not None) == bool(force)
            g = torch.Generator().manual_seed(77)
            losses = []
            for _ in range(3):
                ids = torch.randint(4, 512, (4, 1024), generator=g).to(dev)
                losses.append(step(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids))
            torch.cuda.synchronize()
            nb = len(step.dp.buckets) if force else 0
            return [float(x) for x in losses], [p.detach().clone() for p in m.parameters()], nb
        try:
            l_plain, p_plain, _ = three_steps(False)
            l_dp, p_dp, nb2 = three_steps(True)
        finally:
            ops.GEMM_DYNAMIC_QUEUE = False
            ops.TN_DYNAMIC_QUEUE = False
            os.environ["APERTIS_FORCE_DP"] = "1"
        same = l_plain == l_dp and all(torch.equal(a, b) for a, b in zip(p_plain, p_dp))
        out.put(res + (same, l_plain, l_dp, nb2, int(ops.scan_gate_error(dev))))
    finally:
        dist.destroy_process_group()


def test_rccl_avg_all_reduce_on_the_side_stream(dev):
    """ReduceOp.AVG on the RCCL backend through BucketedDataParallel (world 1, wrapper forced: the one-card box cannot host
    two RCCL ranks): every bucket goes through dist.all_reduce(op=AVG) on the communication stream and the gradients that
    come back equal the plain backward's (mean over one rank)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    try:
        ok, reduced, total, nb, same, l_plain, l_dp, nb2, scan_err = q.get(timeout=600)
    finally:
        p.join(timeout=120)
        if p.is_alive():
            p.kill()
    assert p.exitcode == 0
    assert ok, "gradients after the RCCL AVG all-reduce differ from the plain backward"
    assert nb > 1 and reduced == total, (reduced, total)       # every bucket was handed to all_reduce exactly once
    # three data-parallel TrainSteps (RCCL AVG on the side stream, NT tile queue and weight-gradient item queue on) give
    # the plain steps' losses and weights bit for bit, and the scan's look-back never timed out next to the collectives
    assert all(l == l for l in l_plain) and nb2 > 1 and scan_err == 0, (l_plain, nb2, scan_err)
    assert same, (l_plain, l_dp)
