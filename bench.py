#!/usr/bin/env python3
"""bench.py — training-step throughput of the Apertis hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1.5b-moe] [--batch B]

A "step" is one data-parallel training step (bf16-autocast forward + loss + backward +
gradient all-reduce + clip + AdamW) of the BASELINE.json model on synthetic random tokens,
random-init weights, reference-default dropout / noisy routing / expert capacity.
N>1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ...`,
one rank per GPU (RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("PYTORCH_CUDA_ALLOC_CONF", "expandable_segments:True")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (target params, moe, multimodal, seq, default per-GPU batch)   BASELINE.json configs[1..4]
    "125m": ("125M", False, False, 2048, 32),
    "350m-moe": ("350M", True, False, 4096, 16),
    # per-GPU batch 32 x 4096 tokens: 181 GiB of the 288 GB HBM3E (measured); B=24 gives 244k tok/s, 32: 249k
    "1.5b-moe": ("1.5B", True, False, 4096, 32),
    "1.5b-moe-mm": ("1.5B", True, True, 2048, 16),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="1.5b-moe", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU micro-batch (0 = config default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--reduce-dtype", default="fp32", choices=["fp32", "bf16"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("APERTIS_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)

    import apertis_llm_amd as A
    from apertis_llm_amd import ops
    from apertis_llm_amd.training import TrainStep

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - T0:.1f}s] {msg}", file=sys.stderr, flush=True)

    T0 = time.perf_counter()
    target, moe, mm, seq, dbatch = CONFIGS[args.config]
    B = args.batch or dbatch
    torch.manual_seed(0)   # identical replicas (weights); data seeds differ per rank below
    model = A.create_apertis_model(target, vocab_size_override=32000, multimodal=mm, use_expert_system=moe,
                                   attention_type_override="selective_ssm")
    cfg = model.config
    log("model built on host")
    model = model.to(dev).train()
    n_params = sum(p.numel() for p in model.parameters())
    step = TrainStep(model, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0, total_steps=args.steps + args.warmup + 1,
                     bf16=True, reduce_dtype=torch.bfloat16 if args.reduce_dtype == "bf16" else None)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)

    def batch():
        ids = torch.randint(4, cfg.vocab_size, (B, seq), device=dev, generator=gen)
        b = {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids}
        if mm:
            b["pixel_values"] = torch.randn(B, 3, cfg.image_size, cfg.image_size, device=dev, generator=gen)
        return b

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if not os.environ.get("APERTIS_BENCH_NO_PRETOUCH"):
        # set-up, not a step: map the caching allocator's (expandable) segment once, so that no step - timed or
        # warm-up - pays for hipMemMap calls while its activations grow to their peak
        free_b, _total = torch.cuda.mem_get_info(dev)
        pre = torch.empty(int(free_b * 0.90), dtype=torch.uint8, device=dev)
        del pre
        torch.cuda.reset_peak_memory_stats(dev)
    log(f"model on device ({n_params / 1e6:.0f}M params); warmup")
    for i in range(args.warmup):
        tw = time.perf_counter()
        loss = step(**batch())
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {1e3 * (time.perf_counter() - tw):.0f} ms, peak mem "
            f"{torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    timer = None
    if not args.no_kernel_timers:
        timer = ops.KernelTimer(["apertis_grouped_gemm_nt", "apertis_grouped_gemm_tn", "apertis_selective_scan_fwd",
                                 "apertis_selective_scan_bwd", "apertis_grouped_gemm_nt[dense]",
                                 "apertis_grouped_gemm_tn[dense]"])
    sync()
    ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(**batch())
    sync()
    elapsed = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    log(f"timed region done: {elapsed:.2f}s for {args.steps} steps")
    if getattr(step, "dp", None) is not None:
        log(f"DP wrapper: {step.dp.copied_bytes / 2**30:.2f} GiB of gradients moved into buckets by hooks so far "
            f"({step.dp.gradient_bytes() / 2**30:.2f} GiB of gradients per step)")
    last_loss = float(loss)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    tokens = B * seq * world * args.steps
    result = {
        "metric": "train tokens/sec/node (seq=%d)" % seq, "value": tokens / elapsed, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.config}: Apertis selective-SSM{' + 8-expert top-2 MoE' if moe else ''}"
                               f"{' + multimodal' if mm else ''}, H={cfg.hidden_size} layers={cfg.num_hidden_layers} "
                               f"heads={cfg.num_attention_heads} I={cfg.intermediate_size} vocab={cfg.vocab_size}, "
                               f"{n_params / 1e6:.0f}M params, train step fwd+loss+bwd+allreduce+clip+AdamW, "
                               f"reference-default dropout/noise/capacity",
                   "global_batch": B * world, "per_gpu_batch": B, "seq_len": seq, "parallelism": f"dp{world}",
                   "final_loss": last_loss},
    }
    if rank == 0 and timer is not None:
        summ = timer.summary()
        rl = {}
        # HBM traffic per call from the committed rocprofv3 --pmc passes of the same launches
        # (tools/run_pmc.sh benchmix -> profiles/r1_pmc_traffic_*.json); only valid for the workload
        # those passes were taken on, else null
        traffic = {}
        tf = os.path.join(ROOT, "profiles", f"r1_pmc_traffic_{args.config}_b{B}.json")
        if os.path.exists(tf):
            with open(tf) as fh:
                traffic = {k: v.get("traffic_bytes_per_call") for k, v in json.load(fh).items()}
        for name, d in summ.items():
            if not d["launches"]:
                continue
            avg_ms = d["ms"] / d["launches"]
            per_launch = d["work"] / d["launches"]
            if "gemm" in name:
                ach = per_launch / (avg_ms * 1e-3) / 1e12
                rl[name] = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": traffic.get(name), "launches": d["launches"],
                            "avg_ms": avg_ms, "total_ms": d["ms"]}
            else:
                ach = per_launch / (avg_ms * 1e-3) / 1e9
                rl[name] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ach / HBM_PEAK_GBS, "traffic": traffic.get(name), "launches": d["launches"], "avg_ms": avg_ms,
                            "total_ms": d["ms"]}
        if rl:
            dom = max((k for k in rl if "[dense]" not in k), key=lambda k: rl[k]["total_ms"])
            result["roofline"] = dict(rl[dom], kernel=dom)
            result["roofline_all"] = rl
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU oracle in a CHILD process (never touches the GPU), bounded sample: one layer at
        # B=1, L=1024 of the same shape; a hard timeout keeps the default run within minutes
        import subprocess
        log("cpu baseline (oracle on host cores, child process)")
        argv = [sys.executable, "-m", "oracle.cpu_baseline"] + [str(int(v)) for v in (
            cfg.hidden_size, cfg.num_attention_heads, cfg.ssm_d_state, cfg.intermediate_size, max(cfg.num_experts, 1),
            max(cfg.experts_per_token, 1), int(moe), 1024, cfg.vocab_size, cfg.num_hidden_layers)]
        try:
            out = subprocess.run(argv, cwd=ROOT, capture_output=True, text=True, timeout=180)
            result["cpu_baseline"] = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as exc:  # timeout / parse error: report it, never fail the bench line
            result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": None, "kind": "port",
                                      "sample": f"oracle run did not finish: {type(exc).__name__}"}
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to stdout through C stdio (flushed at exit): push it out first so the
        # JSON line is the LAST line of stdout whatever else the runtime printed
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
