"""Data-parallel path on CPU: world_size-2 gloo processes (the GPU path is the same code with
RCCL and a side stream).  Oracle for DP = mean of independently computed per-shard gradients
(SURVEY.md §8e)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(1234)
    return nn.Sequential(nn.Linear(16, 64), nn.GELU(), nn.LayerNorm(64), nn.Linear(64, 64), nn.GELU(), nn.Linear(64, 8))


def _data(rank):
    g = torch.Generator().manual_seed(1000 + rank)
    return torch.randn(12, 16, generator=g), torch.randn(12, 8, generator=g)


def _worker(rank, world, port, bucket_bytes, reduce_dtype, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # (the container's hostname may not resolve: pair connections over loopback)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from apertis_llm_amd.parallel import BucketedDataParallel
        torch.set_num_threads(1)
        model = _model()
        if rank == 1:                      # replicas must be re-synchronised from rank 0
            with torch.no_grad():
                for p in model.parameters():
                    p.add_(1.0)
        dp = BucketedDataParallel(model, bucket_bytes=bucket_bytes, reduce_dtype=reduce_dtype)
        x, y = _data(rank)
        # two accumulation micro-steps: reduce only on the last one
        with dp.no_sync():
            ((dp(x[:6]) - y[:6]) ** 2).mean().backward()
        ((dp(x[6:]) - y[6:]) ** 2).mean().backward()
        dp.finish()
        grads = [p.grad.clone() for p in model.parameters()]
        params = [p.detach().clone() for p in model.parameters()]
        dp.zero_grad()
        assert all(p.grad is None for p in model.parameters())
        ((dp(x) - y) ** 2).mean().backward()       # a second step reuses the buckets
        dp.finish()
        np_ = lambda ts: [t.detach().numpy().copy() for t in ts]     # plain pickles: no fd passing
        out.put((rank, np_(grads), np_(params), len(dp.buckets), np_([p.grad for p in model.parameters()])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,reduce_dtype", [(1 << 30, None), (4096, None), (4096, torch.bfloat16)])
def test_bucketed_allreduce_equals_mean_of_shard_gradients(bucket_bytes, reduce_dtype):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_bytes, reduce_dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, [torch.from_numpy(a) for a in g], [torch.from_numpy(a) for a in pr], nb, [torch.from_numpy(a) for a in g2])
           for r, g, pr, nb, g2 in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reference: per-shard gradients computed in this process, then averaged
    ref1, ref2 = [], []
    for rank in range(world):
        m = _model()
        x, y = _data(rank)
        ((m(x[:6]) - y[:6]) ** 2).mean().backward()
        ((m(x[6:]) - y[6:]) ** 2).mean().backward()
        ref1.append([p.grad.clone() for p in m.parameters()])
        m.zero_grad()
        ((m(x) - y) ** 2).mean().backward()
        ref2.append([p.grad.clone() for p in m.parameters()])
    tol = dict(rtol=1e-5, atol=1e-6) if reduce_dtype is None else dict(rtol=2e-2, atol=2e-3)
    base = [p.detach() for p in _model().parameters()]
    for rank, grads, params, nb, grads2 in res:
        assert nb == (1 if bucket_bytes > (1 << 20) else nb) and (bucket_bytes > (1 << 20) or nb > 1)
        for g, a, b in zip(grads, ref1[0], ref1[1]):
            assert torch.allclose(g, (a + b) / 2, **tol)
        for g, a, b in zip(grads2, ref2[0], ref2[1]):
            assert torch.allclose(g, (a + b) / 2, **tol)
        for p, b0 in zip(params, base):
            assert torch.equal(p, b0)                # rank 1 was reset to rank 0's weights
    for g0, g1 in zip(res[0][1], res[1][1]):
        assert torch.equal(g0, g1)                   # every rank holds the same reduced gradient


def _train_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # (the container's hostname may not resolve: pair connections over loopback)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import apertis_llm_amd as A
        from apertis_llm_amd.training import TrainStep
        torch.set_num_threads(1)
        torch.manual_seed(0)
        cfg = A.ApertisConfig(vocab_size=128, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                              intermediate_size=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        model = A.ApertisForCausalLM(cfg).train()      # standard_mha: the stock-torch path runs on CPU
        step = TrainStep(model, lr=1e-2, total_steps=10, bf16=False, bucket_bytes=8192)
        g = torch.Generator().manual_seed(1000 + rank)
        losses = []
        for _ in range(3):
            ids = torch.randint(4, 128, (2, 16), generator=g)
            losses.append(float(step(input_ids=ids, labels=ids)))
        out.put((rank, losses, [p.detach().numpy().copy() for p in model.parameters()]))
    finally:
        dist.destroy_process_group()


def test_train_step_two_ranks_keep_replicas_identical():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    res = [(r, l, [torch.from_numpy(a) for a in ps]) for r, l, ps in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)
    assert res[0][1] != res[1][1]                     # different data shards -> different local losses


# ----------------------------------------------------------------------------------------------------------------
# the trainer's data-parallel wiring (DistributedSampler shards, no_sync on accumulation micro-steps, one reduction per
# optimizer step, checkpoints on rank 0 only): two gloo ranks must end with identical weights, equal to ONE process
# training on the union of the two shards with the two ranks' micro-batches averaged.
class _ToyLM(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(7)
        self.emb = nn.Embedding(16, 8)
        self.LayerNorm = nn.LayerNorm(8)
        self.out = nn.Linear(8, 16)

    def forward(self, input_ids, attention_mask=None, labels=None):
        logits = self.out(self.LayerNorm(self.emb(input_ids)))
        loss = nn.functional.cross_entropy(logits[:, :-1].reshape(-1, 16), labels[:, 1:].reshape(-1), ignore_index=-100)
        return (loss, logits)


def _toy_dataset(path):
    import json
    import numpy as np
    from apertis_llm_amd import data as D
    rng = np.random.RandomState(0)
    with open(path, "w") as f:
        for _ in range(8):
            f.write(json.dumps({"text": " ".join(rng.choice(list("abcdefgh"), size=8))}) + "\n")
    return D.ApertisPretrainDataset(path, {c: i + 4 for i, c in enumerate("abcdefgh")}, 16, max_length=8)


def _trainer_worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # (the container's hostname may not resolve: pair connections over loopback)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from apertis_llm_amd.trainer import ApertisTrainer
        torch.set_num_threads(1)
        ds = _toy_dataset(os.path.join(tmp, f"d{rank}.jsonl"))
        t = ApertisTrainer(_ToyLM(), ds, None, output_dir=os.path.join(tmp, "out"), batch_size=2, learning_rate=1e-2,
                           num_epochs=2, gradient_accumulation_steps=2, fp16=False, device="cpu", checkpoint_steps=0,
                           distributed_training=True, local_rank=rank, num_workers=0)
        assert t.dp is not None and t.world_size == 2 and t.is_main_process == (rank == 0)
        order = [b["input_ids"].tolist() for b in t.train_dataloader]       # this rank's shard of epoch 0 (set_epoch(0))
        t.train()
        out.put((rank, [p.detach().numpy().copy() for p in t.model.parameters()], t.history["checkpoints"], order,
                 len(t.history["loss"])))
    finally:
        dist.destroy_process_group()


def test_trainer_two_ranks_equal_and_checkpoint_on_rank0(tmp_path):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, str(tmp_path), out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, ck0, order0, n0), (_, w1, ck1, order1, n1) = res
    for a, b in zip(w0, w1):
        assert (a == b).all(), "replicas diverged"
    # 8 items over 2 ranks = 4 per rank = 2 batches per epoch = 1 optimizer step per epoch (accumulation 2)
    assert n0 == n1 == 2
    assert ck0 == ["epoch-1", "epoch-2", "final"] and ck1 == []
    flat0 = [tuple(x) for b in order0 for x in b]
    flat1 = [tuple(x) for b in order1 for x in b]
    assert len(flat0) == len(flat1) == 4 and not set(flat0) & set(flat1), "the sampler must shard the epoch"
    assert sorted(os.listdir(tmp_path / "out")) == ["epoch-1", "epoch-2", "final"]


def test_late_gradient_copies_land_before_the_next_backward():
    """Round 6: gradients autograd produces outside the bucket are moved in with ONE multi-tensor copy per bucket
    (parallel._Bucket.flush) instead of one copy per parameter.  A bucket that never completes in a backward (one of its
    parameters got no gradient) must still hold its data when that backward ends - the next backward accumulates INTO the
    views.  Single process, no process group: two backwards without finish() in between, gradients must be their sum."""
    from apertis_llm_amd.parallel import BucketedDataParallel
    torch.manual_seed(0)

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.unused = nn.Linear(8, 8), nn.Linear(8, 4), nn.Linear(4, 4)

        def forward(self, x):
            return self.b(torch.tanh(self.a(x)))
    m, ref = M(), M()
    ref.load_state_dict(m.state_dict())
    dp = BucketedDataParallel(m, bucket_bytes=1 << 20, broadcast_parameters=False)   # one bucket: `unused` keeps it incomplete
    xs = [torch.randn(5, 8), torch.randn(5, 8)]
    for x in xs:
        dp(x).square().sum().backward()
        ref(x).square().sum().backward()
        assert all(not b.late_dst for b in dp.buckets), "the end-of-backward callback must have flushed the late copies"
    dp.finish()
    for (n, p), q in zip(m.named_parameters(), ref.parameters()):
        if q.grad is None:
            continue
        assert p.grad is not None and torch.allclose(p.grad, q.grad, rtol=1e-6, atol=1e-7), n
        assert any(p.grad.data_ptr() >= b.flat.data_ptr() and p.grad.data_ptr() < b.flat.data_ptr() + b.flat.numel() * 4
                   for b in dp.buckets), "gradients live in their bucket slices"
