"""Host time to ENQUEUE a training step against the time the GPU takes for it, with the one-launch weight preparation
(ops.TRAIN_PREP) on and off, in one process: python tools/host_enqueue.py [config=350m-moe] [steps=12]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYTORCH_CUDA_ALLOC_CONF", "expandable_segments:True")
import torch
import apertis_llm_amd as A
from apertis_llm_amd import ops
from apertis_llm_amd.training import TrainStep
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "350m-moe"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
target, moe, mm, seq, B = bench.CONFIGS[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = A.create_apertis_model(target, vocab_size_override=32000, multimodal=mm, use_expert_system=moe,
                               attention_type_override="selective_ssm").to(dev).train()
gen = torch.Generator(device=dev).manual_seed(1)
ids = torch.randint(4, 32000, (B, seq), device=dev, generator=gen)
for on in (True, False, True, False):
    ops.TRAIN_PREP = on
    step = TrainStep(model, lr=5e-5, total_steps=10 * steps)
    for _ in range(4):
        step(input_ids=ids, labels=ids)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = []
    e0.record()
    t_all = time.perf_counter()
    for _ in range(steps):
        t0 = time.perf_counter()
        step(input_ids=ids, labels=ids)
        host.append(time.perf_counter() - t0)
    e1.record(); e1.synchronize()
    wall = (time.perf_counter() - t_all) / steps
    host.sort()
    print(f"{name} TRAIN_PREP={int(on)}: host enqueue median {host[len(host) // 2] * 1e3:7.1f} ms (min {host[0] * 1e3:.1f}), "
          f"step {e0.elapsed_time(e1) / steps:7.1f} ms (wall {wall * 1e3:.1f})", flush=True)
    del step
