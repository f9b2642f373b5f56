"""Which torch (non-library) kernels a training step still launches, by op and Python source line: a 4-layer model of the 1.5B
family (H=704, 8 experts) under torch.profiler.  python tools/torch_ops_per_step.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import apertis_llm_amd as A
from apertis_llm_amd.training import TrainStep
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = A.ApertisConfig(vocab_size=32000, hidden_size=704, num_hidden_layers=4, num_attention_heads=11, intermediate_size=2816,
                      attention_type="selective_ssm", use_expert_system=True, num_experts=8, experts_per_token=2)
model = A.ApertisForCausalLM(cfg).to(dev).train()
step = TrainStep(model, total_steps=20)
ids = torch.randint(4, 32000, (4, 4096), device=dev)
for _ in range(3):
    step(input_ids=ids, labels=ids)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(input_ids=ids, labels=ids)
    torch.cuda.synchronize()
ops = collections.Counter()
where = collections.defaultdict(collections.Counter)
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::"):
        continue
    kids = [k for k in ev.cpu_children if k.name.startswith("aten::")]
    if not ev.kernels or kids and any(k.kernels for k in kids):
        continue          # count the innermost aten op that launches
    st = [s for s in (ev.stack or []) if "torch/" not in s and "<built-in" not in s and "runpy" not in s]
    src = " < ".join(x.split("/")[-1][:44] for x in st[:3]) if st else ""
    if not src:          # backward: no Python stack - name the autograd node the op runs under
        par = ev.cpu_parent
        while par is not None and "evaluate_function" not in par.name and "Backward" not in par.name:
            par = par.cpu_parent
        src = par.name.replace("autograd::engine::evaluate_function: ", "bwd of ") if par is not None else "?"

    ops[ev.name] += len(ev.kernels)
    where[ev.name][(src, str(ev.input_shapes)[:60])] += len(ev.kernels)
print("torch ops that launch kernels in one step of a 4-layer model (kernel launches):")
for name, n in ops.most_common(25):
    print(f"  {name:34s} {n:5d}")
    for (src, shp), c in where[name].most_common(10):
        print(f"        {c:4d}  {shp:40s} {src}")
