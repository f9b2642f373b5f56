"""cProfile of the host side of training steps (where the Python time of a launch-bound configuration goes):
python tools/host_profile.py [config=350m-moe] [steps=6]"""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PYTORCH_CUDA_ALLOC_CONF", "expandable_segments:True")
import torch
import apertis_llm_amd as A
from apertis_llm_amd.training import TrainStep
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "350m-moe"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
target, moe, mm, seq, B = bench.CONFIGS[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = A.create_apertis_model(target, vocab_size_override=32000, multimodal=mm, use_expert_system=moe,
                               attention_type_override="selective_ssm").to(dev).train()
ids = torch.randint(4, 32000, (B, seq), device=dev)
step = TrainStep(model, lr=5e-5, total_steps=100)
for _ in range(3):
    step(input_ids=ids, labels=ids)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step(input_ids=ids, labels=ids)
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
ps = pstats.Stats(pr, stream=st).sort_stats("tottime")
ps.print_stats(28)
print(st.getvalue()[:6000])
