#!/usr/bin/env python3
"""Host-side profile of a training step (GPU box): where the Python / dispatcher time and the small framework
launches (fills, copies, cats) of a step come from.

    python tools/host_profile.py --config 350m-moe [--batch B] [--layers N]

1. cProfile over three steps, by own time and by cumulative time (the step is host-bound on the H = 256 family);
2. torch.profiler over one step with Python stacks: every aten::zeros / zero_ / fill_ / copy_ / cat / index op that
   launches a kernel, grouped by the innermost frames of this package, with counts.
Output goes to stdout; nothing is written to the repo."""
import argparse
import cProfile
import collections
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (CONFIGS only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="350m-moe")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--layers", type=int, default=0)
    args = ap.parse_args()
    import torch
    import apertis_llm_amd as A
    from apertis_llm_amd.training import TrainStep
    target, moe, mm, seq, dbatch = bench.CONFIGS[args.config]
    B = args.batch or dbatch
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = A.create_apertis_model(target, vocab_size_override=32000, multimodal=mm, use_expert_system=moe,
                                   attention_type_override="selective_ssm")
    if args.layers:
        model.model.layers = model.model.layers[:args.layers]
    model = model.to(dev).train()
    step = TrainStep(model, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0, total_steps=64, bf16=True)
    gen = torch.Generator(device=dev).manual_seed(1)

    def batch():
        ids = torch.randint(4, model.config.vocab_size, (B, seq), device=dev, generator=gen)
        return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids}

    for _ in range(3):
        step(**batch())
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(3):
        step(**batch())
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"== 3 steps: host returned after {1e3 * t_host / 3:.1f} ms/step, GPU done after {1e3 * t_all / 3:.1f} ms/step")

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step(**batch())
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
        print(f"== cProfile, 3 steps, by {key}")
        print(s.getvalue())

    from torch.profiler import profile, ProfilerActivity
    try:
        xc = torch._C._profiler._ExperimentalConfig(verbose=True)
    except Exception:
        xc = None
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=xc) as prof:
        step(**batch())
        torch.cuda.synchronize()
    names = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::cat", "aten::index", "aten::zeros_like",
             "aten::new_zeros", "aten::full", "aten::add", "aten::mul", "aten::sum", "aten::to", "aten::_to_copy",
             "aten::clone", "aten::contiguous", "aten::index_copy_", "aten::index_add_", "aten::gather", "aten::where")
    groups = collections.Counter()
    for ev in prof.events():
        if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::cat") and ev.device_time_total > 0:
            frames = [f for f in (ev.stack or []) if "apertis_llm_amd" in f or "bench" in f]
            allf = [f for f in (ev.stack or [])]
            where = " <- ".join(f.split("apertis_llm_amd/")[-1] for f in frames[:3]) or ("(no package frame) " + " <- ".join(a.split("/")[-1] for a in allf[:3]))
            groups[(ev.name, where)] += 1
    print("== framework ops that launch kernels, one step, by call site")
    for (name, where), n in sorted(groups.items(), key=lambda kv: -kv[1])[:80]:
        print(f"{n:6d}  {name:20s} {where}")
    print("== kernels by count, one step")
    kc = collections.Counter()
    kt = collections.Counter()
    for ev in prof.events():
        if ev.device_type is not None and str(ev.device_type).endswith("CUDA"):
            kc[ev.name[:90]] += 1
            kt[ev.name[:90]] += ev.device_time_total
    for name, n in kc.most_common(60):
        print(f"{n:6d} {kt[name] / 1e3:9.2f} ms  {name}")
    print("total kernel launches:", sum(kc.values()), " kernel time %.1f ms" % (sum(kt.values()) / 1e3))


if __name__ == "__main__":
    main()
