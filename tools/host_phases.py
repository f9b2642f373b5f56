#!/usr/bin/env python3
"""Host wall-clock per phase of a training step (GPU box): forward, backward, optimizer pieces, with and without the
cyclic garbage collector.  python tools/host_phases.py --config 350m-moe"""
import argparse
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="350m-moe")
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()
    import torch
    import apertis_llm_amd as A
    from apertis_llm_amd.training import TrainStep, clip_and_step
    target, moe, mm, seq, dbatch = bench.CONFIGS[args.config]
    B = args.batch or dbatch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = A.create_apertis_model(target, vocab_size_override=32000, multimodal=mm, use_expert_system=moe,
                                   attention_type_override="selective_ssm").to(dev).train()
    step = TrainStep(model, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0, total_steps=200, bf16=True)
    gen = torch.Generator(device=dev).manual_seed(1)
    ids = [torch.randint(4, model.config.vocab_size, (B, seq), device=dev, generator=gen) for _ in range(4)]

    def batch(i):
        return {"input_ids": ids[i % 4], "attention_mask": torch.ones_like(ids[0]), "labels": ids[i % 4]}

    for i in range(4):
        step(**batch(i))
    torch.cuda.synchronize()
    opt = step.optimizer

    def phased(n, label):
        acc = {}
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for i in range(n):
            t = [time.perf_counter()]
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = model(**batch(i))[0]
            t.append(time.perf_counter())
            loss.backward()
            t.append(time.perf_counter())
            cur = opt._tables_current()
            t.append(time.perf_counter())
            clip_and_step(opt, step._params, 1.0)
            t.append(time.perf_counter())
            step.scheduler.step()
            opt.zero_grad(set_to_none=True)
            t.append(time.perf_counter())
            for k, a, b in zip(("forward", "backward", "tables_current", "clip+step", "sched+zero_grad"), t, t[1:]):
                acc[k] = acc.get(k, 0.0) + (b - a)
        t_host = time.perf_counter() - t_all
        torch.cuda.synchronize()
        t_gpu = time.perf_counter() - t_all
        print(f"== {label}: host {1e3 * t_host / n:.1f} ms/step, GPU done {1e3 * t_gpu / n:.1f} ms/step; " +
              ", ".join(f"{k} {1e3 * v / n:.1f}" for k, v in acc.items()) + f"  (tables current: {cur})")

    phased(6, "gc on")
    phased(6, "gc on")
    gc.collect()
    gc.freeze()
    gc.disable()
    phased(6, "gc off")
    phased(6, "gc off")
    gc.enable()
    print("gc counts", gc.get_count(), "objects", len(gc.get_objects()), "frozen", gc.get_freeze_count())
    t0 = time.perf_counter()
    gc.collect()
    print(f"full collection: {1e3 * (time.perf_counter() - t0):.1f} ms")


if __name__ == "__main__":
    main()
