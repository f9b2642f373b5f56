#!/usr/bin/env python3
"""bench.py — training-step throughput of the Apertis hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1.5b-moe] [--batch B]

A "step" is one data-parallel training step (bf16-autocast forward + loss + backward +
gradient all-reduce + clip + AdamW) of the BASELINE.json model on synthetic random tokens,
random-init weights, reference-default dropout / noisy routing / expert capacity.
N>1: one rank per GPU over RCCL.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N ...`:
RANK / LOCAL_RANK / WORLD_SIZE in the environment) this file is a rank; invoked plainly as
`python bench.py --gpus N` it starts the N ranks itself as child processes (before any HIP call is made in
the parent; nothing is exec'ed from a process that has touched the GPU) and relays rank 0's line.
`--gpus` must equal the world size.  Rank 0 prints ONE JSON line (the last line of stdout).
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
    # configs 3 and 5 fill the chip too (round 6 sweeps, profiles/r6_batch_sweep_configs_3_5.txt): at per-GPU
    # batch 16 the 350m-moe step was bound by the HOST's launch rate (115 ms of thousands of small launches); 350m-moe 16 / 32 /
    # 48 / 64 / 72 / 76: 570 k / 693 k / 764 k / 795 k / 798 k / 811 k tokens/s at 55 / 105 / 155 / 203 / 227 / 239 GiB
    "350m-moe": ("350M", True, False, 4096, 72),
    # per-GPU batch 44 x 4096 tokens: ~232 GiB of the HBM3E (40: 214 GiB, 46: 242 GiB measured; the LM head + loss no longer
    # hold the [B, L, 32000] logits).  The batch follows the tile counts of the persistent expert GEMMs: with the 256 x 352
    # tile fc2 forward / fc1 dgrad have 40 * B tiles on 256 CUs - 6.25 rounds at B = 40 (run as 7), 6.875 at B = 44.  Same
    # box, end of round 2, 10 timed steps: 36: 348.3k tokens/s, 40: 348.5-352.6k, 44: 357.9-358.2k.  (Before that tile:
    # 34: 328.8k, 38: 333.7k, 40: 330.1-334.3k, 42: 332.4-337.3k, 46: 331.7-335.1k - inside the box-to-box spread.)
    "1.5b-moe": ("1.5B", True, False, 4096, 44),
    # 1.5b-moe-mm (224 x 224 image + 2048 text tokens per sequence) 16 / 32 / 48 / 64 / 72: 247 k / 304 k / 324 k / 338 k / 341 k
    # tokens/s at 68 / 112 / 155 / 199 / 220 GiB
    "1.5b-moe-mm": ("1.5B", True, True, 2048, 72),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak


def launch_ranks(n, argv):
    """Parent of a self-launched N-rank run (reference: DDP ranks of pipeline.py:435-447,463,501-505, which the
    reference leaves to an external launcher).  Starts N children of this file with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, never imports torch, relays rank 0's JSON line as its own last stdout line and returns non-zero if
    any child failed (the others are then terminated so that nobody waits in a rendezvous for ever)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # rank 0's stdout carries the result line; the other ranks' stdout joins stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno(), text=(r == 0)))
    out0 = []
    import threading
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout), daemon=True)
    reader.start()
    rc, live = 0, set(range(n))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.2)
    reader.join(timeout=10)
    lines = [ln.rstrip("\n") for ln in out0 if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)
    if rc == 0 and not (lines and lines[-1].lstrip().startswith("{")):
        print("bench.py: rank 0 printed no result line", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def launcher_selftest(args, rank, world):
    """The rank protocol of main() on CPU over gloo with a stub step (a tiny Linear model through
    BucketedDataParallel): rendezvous, barrier-bracketed timing, MAX over ranks, one JSON line from rank 0.
    Used by tests/test_bench_launcher_cpu.py; the line is marked as a self-test, never a measurement."""
    import torch
    import torch.distributed as dist
    from apertis_llm_amd.parallel import BucketedDataParallel
    if world > 1:
        dist.init_process_group(backend="gloo")
    torch.manual_seed(0)
    model = torch.nn.Linear(32, 32)
    dp = BucketedDataParallel(model) if world > 1 else None
    gen = torch.Generator().manual_seed(1000 + rank)

    def step():
        model(torch.randn(8, 32, generator=gen)).square().mean().backward()
        if dp is not None:
            dp.finish()
            dp.zero_grad()
        else:
            model.zero_grad()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    reduced = dp.reduced_bytes // max(args.steps + args.warmup, 1) if dp is not None else 0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        ranks = dist.get_world_size()
        dist.barrier()
        dist.destroy_process_group()
    else:
        ranks = 1
    if rank == 0:
        print(json.dumps({"metric": "LAUNCHER SELF-TEST (stub step on CPU over gloo; not a measurement)",
                          "value": 8 * world * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "selftest": True,
                          "config": {"workload": "stub", "parallelism": f"dp{world}", "rccl_ranks": ranks,
                                     "allreduce_bytes_per_step": reduced}}), flush=True)
    return 0


def decode_bench(args, model, dev, seq, log):
    """generate() as a measured path (reference core.py:1474-1644): prefill of 2048 random tokens, then 128
    greedy tokens through the cached single-token step.  tokens/s counts NEW tokens of all sequences; the roofline is the
    weight-streaming floor: every token step reads the model's active weights once (experts: the expected fraction
    1 - (1 - K/E)^B that B sequences' top-K choices touch under uniform routing) in the compute dtype."""
    import torch
    from apertis_llm_amd import ops
    cfg = model.config
    model = model.to(dev).eval()
    prefill, new = min(2048, seq), 128
    n_total = sum(p.numel() for p in model.parameters())
    n_exp = sum(p.numel() for n, p in model.named_parameters() if "expert_w" in n)
    e_b = 2 if args.decode_dtype == "bf16" else 4
    K, E = max(cfg.experts_per_token, 1), max(cfg.num_experts, 1)
    lines = []
    for B in ([args.batch] if args.batch else [1, 16]):
        gen = torch.Generator(device=dev).manual_seed(7)
        ids = torch.randint(4, cfg.vocab_size, (B, prefill), device=dev, generator=gen)

        def run(n_new):
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.decode_dtype == "bf16"):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = model.generate(ids, max_new_tokens=n_new, eos_token_id=[-1], use_cache=True)
                torch.cuda.synchronize()
                return time.perf_counter() - t0, out
        run(4)                                        # warm-up: prepared weights, allocator, kernels
        t_pre, _ = run(1)                             # prefill + first token
        t_all, out = run(new)
        assert out.shape == (B, prefill + new)
        per_tok = (t_all - t_pre) / (new - 1)
        frac = 1.0 - (1.0 - K / E) ** B if n_exp else 0.0
        wbytes = e_b * ((n_total - n_exp) + n_exp * frac)
        lines.append({"batch": B, "prefill_ms": 1e3 * t_pre, "ms_per_token_step": 1e3 * per_tok, "tokens_per_s": B / per_tok,
                      "weight_bytes_per_step": wbytes, "weight_stream_GBps": wbytes / per_tok / 1e9,
                      "frac_of_hbm_peak": wbytes / per_tok / 1e9 / HBM_PEAK_GBS})
        log(f"decode B={B}: prefill {1e3 * t_pre:.1f} ms, {1e3 * per_tok:.2f} ms per token step, {B / per_tok:.1f} tokens/s, "
            f"{wbytes / per_tok / 1e9:.0f} GB/s of weights")
    best = max(lines, key=lambda d: d["tokens_per_s"])
    err = ops.scan_gate_error(dev)
    print(json.dumps({"metric": f"generate() new tokens/sec ({args.config}, {prefill}-token prefill, {new} new tokens, greedy)",
                      "value": best["tokens_per_s"], "unit": "tokens/s", "n_gpus": 1, "higher_is_better": True,
                      "dtype": args.decode_dtype, "data": "synthetic",
                      "config": {"workload": f"{args.config} decode", "prefill": prefill, "new_tokens": new,
                                 "scan_lookback_error_word": err},
                      "roofline": {"bound": "hbm", "achieved": best["weight_stream_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": best["frac_of_hbm_peak"], "traffic": None, "kernel": "weight stream of one token step"},
                      "per_batch": lines}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 10 warm-up, >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="1.5b-moe", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU micro-batch (0 = config default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--reduce-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--logits-path", action="store_true", help="A/B: materialise the [B, L, V] logits (LM head, then the loss)")
    ap.add_argument("--decode", action="store_true",
                    help="measure generate() instead of the training step: 2048-token prefill, then 128 greedy tokens "
                         "(reference core.py:1520-1644), per-GPU batch --batch (default 1 and 16), --decode-dtype")
    ap.add_argument("--decode-dtype", default="bf16", choices=["bf16", "fp32"],
                    help="bf16 = autocast (prepared bf16 weight copies, cached across tokens); fp32 = the model's own dtype")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU/gloo self-test of the rank launcher and the timing protocol (tests/): no GPU work, "
                         "the line says so and is not a measurement")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # not under a launcher: be the launcher.  Nothing above this line imports torch or touches HIP.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a "
                         "different number of ranks than was asked for")
    if args.launcher_selftest:
        raise SystemExit(launcher_selftest(args, rank, world))
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
    if args.decode:
        raise SystemExit(decode_bench(args, model, dev, seq, log))
    model = model.to(dev).train()
    n_params = sum(p.numel() for p in model.parameters())
    step = TrainStep(model, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0, total_steps=args.steps + args.warmup + 1,
                     bf16=True, reduce_dtype=torch.bfloat16 if args.reduce_dtype == "bf16" else None,
                     fused_lm_head_loss=not args.logits_path)
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
                                 "apertis_selective_scan_bwd", "apertis_scan_gate_fwd", "apertis_scan_gate_bwd",
                                 "apertis_grouped_gemm_nt[dense]", "apertis_grouped_gemm_tn[dense]"],
                                # per-call HIP events on every 4th timed step once there are enough of them (the event records
                                # are host work: 24 ms of a 137 ms step on the launch-heavy 350m-moe configuration)
                                every=4 if args.steps >= 20 else 1)
    sync()
    ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if timer is not None:
            timer.next_step()
        loss = step(**batch())
    sync()
    elapsed = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    log(f"timed region done: {elapsed:.2f}s for {args.steps} steps")
    if getattr(step, "dp", None) is not None:
        log(f"DP wrapper: {step.dp.copied_bytes / 2**30:.2f} GiB of gradients moved into buckets by hooks so far "
            f"({step.dp.gradient_bytes() / 2**30:.2f} GiB of gradients per step)")
    last_loss = float(loss)
    # the single-pass scan's bounded look-back wait leaves a non-zero word when it times out (its activations would be wrong;
    # TrainStep turns the loss into NaN and the clip coefficient rejects the step): a line from such a run is not reported
    scan_err = ops.scan_gate_error(dev)
    if scan_err:
        raise SystemExit(f"bench.py: rank {rank}: single-pass scan look-back timed out (error word {scan_err:#x})")
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
                   "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
                   "allreduce_bytes_per_step": (step.dp.reduced_bytes // max(args.steps + args.warmup, 1)
                                                if getattr(step, "dp", None) is not None else 0),
                   "final_loss": last_loss, "scan_lookback_error_word": scan_err},
    }
    if rank == 0 and timer is not None:
        summ = timer.summary()
        rl = {}
        # HBM traffic per call from the committed rocprofv3 --pmc passes of the same launches
        # (tools/run_pmc.sh benchmix -> profiles/r2_pmc_traffic_*.json); only valid for the workload
        # those passes were taken on, else null
        traffic, traffic_source = {}, None
        tf = next((f for f in (os.path.join(ROOT, "profiles", f"r{r}_pmc_traffic_{args.config}_b{B}.json") for r in (6, 5, 4, 3, 2, 1))
                   if os.path.exists(f)), "")
        if os.path.exists(tf):
            import hashlib
            with open(tf) as fh:
                tj = json.load(fh)
            # the counters were taken on particular kernel sources (tools/pmc_traffic.py records their hashes): with other
            # sources in the tree the field is null - a number from an older kernel is not this run's traffic
            want = (tj.get("_source") or {}).get("kernel_source_sha16") or {}
            have = {}
            for f in want:
                try:
                    with open(os.path.join(ROOT, "apertis_llm_amd", "csrc", f), "rb") as fh:
                        have[f] = hashlib.sha256(fh.read()).hexdigest()[:16]
                except OSError:
                    have[f] = None
            if want and want == have:
                traffic = {k: v.get("traffic_bytes_per_call") for k, v in tj.items() if not k.startswith("_")}
                traffic_source = {"file": os.path.relpath(tf, ROOT), "kernel_source_sha16": want}
            else:
                traffic_source = {"file": os.path.relpath(tf, ROOT), "stale": True,
                                  "note": "taken on other kernel sources than this tree's: traffic is null"}
        else:
            traffic_source = {"file": None,
                              "note": "no rocprofv3 --pmc passes in profiles/ for this configuration and batch (they were taken on the "
                                      "headline workload's launches only: tools/run_pmc.sh benchmix replays the 1.5b-moe shapes): traffic is null"}
        result["traffic_source"] = traffic_source
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
        # The expert GEMMs ALSO priced on their bytes (on the H = 256 family they are byte-bound - 114 flop per byte
        # against the chip's 312 - and the MFMA fraction alone says little there).  Algorithmic bytes per routed row over the four
        # NT calls of a layer: fc1 forward reads X and writes h and g' (2(H + 2I)), fc2 forward 2(I + H), the fused fc2 data
        # gradient reads dy and g' and writes dpre (2(H + 2I)), fc1 data gradient 2(I + H); + the bf16 weights once per call.
        # The weight-gradient pair reads dy, h, dpre, X once (2(2H + 2I) per row) and writes both fp32 gradients.
        if moe:
            H_, I_, E_ = cfg.hidden_size, cfg.intermediate_size, cfg.num_experts
            for name, per_row, fixed, flop_row in (("apertis_grouped_gemm_nt", 2.0 * (4 * H_ + 6 * I_) / 4, 2.0 * E_ * I_ * H_, 2.0 * I_ * H_),
                                                   ("apertis_grouped_gemm_tn", 2.0 * (2 * H_ + 2 * I_), 8.0 * E_ * I_ * H_, 4.0 * I_ * H_)):
                if name in rl and rl[name]["launches"]:
                    rows = summ[name]["work"] / summ[name]["launches"] / flop_row          # routed rows per launch (from the flops)
                    nbytes = rows * per_row + fixed
                    gbps = nbytes / (rl[name]["avg_ms"] * 1e-3) / 1e9
                    rl[name]["hbm_bound"] = {"bytes_per_launch": nbytes, "achieved": gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": gbps / HBM_PEAK_GBS, "flop_per_byte": flop_row * rows / nbytes}
                    # a GEMM whose arithmetic intensity sits under the chip's ridge (2.5 PFLOP/s / 8 TB/s = 312
                    # flop per byte) is bounded by HBM - the H = 256 family at 114 - and its headline fraction is the HBM one;
                    # the MFMA pricing stays beside it
                    if flop_row * rows / nbytes < MFMA_BF16_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
                        e_ = rl[name]
                        e_["mfma_bound"] = {"achieved": e_["achieved"], "peak": e_["peak"], "unit": e_["unit"], "frac": e_["frac"]}
                        e_.update(bound="hbm", achieved=gbps, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbps / HBM_PEAK_GBS)
        # the SSM block's dense projections are narrow (N, K <= 704: 88 - 235 flop per byte against the chip's 312): their
        # bound is HBM, so each shape is ALSO priced on its algorithmic bytes (X once, Y once, W once; weight gradient: both
        # operands once + the split-K partials)
        for name, d in summ.items():
            for tag, b in sorted(d.get("by_shape", {}).items()):
                if b["launches"] and b["ms"] > 0:
                    ach = b["bytes"] / (b["ms"] * 1e-3) / 1e9
                    rl[f"{name} {tag}"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": ach / HBM_PEAK_GBS, "traffic": None, "launches": b["launches"],
                                           "avg_ms": b["ms"] / b["launches"], "total_ms": b["ms"],
                                           "tflops": b["work"] / (b["ms"] * 1e-3) / 1e12}
        if rl:
            dom = max((k for k in rl if "[dense]" not in k), key=lambda k: rl[k]["total_ms"])
            result["roofline"] = dict(rl[dom], kernel=dom)
            result["roofline_all"] = rl
            result["roofline_steps"] = args.steps // timer.every    # the steps whose launches carry event pairs (`launches` counts those)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU oracle in a CHILD process (never touches the GPU), bounded sample: one layer of the same shape at
        # B=1 and the benchmark's sequence length, train mode: three legs (all cores / 32 threads / vectorised scan), each 1 warm-up + median of 2; a
        # hard timeout keeps the default run within minutes
        import subprocess
        log("cpu baseline (oracle on host cores, child process)")
        argv = [sys.executable, "-m", "oracle.cpu_baseline"] + [str(int(v)) for v in (
            cfg.hidden_size, cfg.num_attention_heads, cfg.ssm_d_state, cfg.intermediate_size, max(cfg.num_experts, 1),
            max(cfg.experts_per_token, 1), int(moe), seq, cfg.vocab_size, cfg.num_hidden_layers)]
        def last_json(text):
            for ln in reversed((text or "").strip().splitlines()):
                try:
                    return json.loads(ln)
                except ValueError:
                    continue
            return None
        try:
            out = subprocess.run(argv, cwd=ROOT, capture_output=True, text=True, timeout=300)
            got, note = last_json(out.stdout), None if out.returncode == 0 else f"child exited {out.returncode}: {out.stderr[-300:]}"
        except subprocess.TimeoutExpired as exc:   # the child emits a complete line after its first legs: keep that one
            txt = exc.stdout.decode() if isinstance(exc.stdout, bytes) else exc.stdout
            got, note = last_json(txt), "the all-usable-cores leg did not finish within 300 s"
        except Exception as exc:  # never fail the bench line on the baseline
            got, note = None, f"{type(exc).__name__}: {exc}"
        if got is None:
            got = {"value": None, "unit": "tokens/s", "cores": None, "kind": "port", "sample": "oracle run produced no line"}
        if note:
            got["note"] = note
        result["cpu_baseline"] = got
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
