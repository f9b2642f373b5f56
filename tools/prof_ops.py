"""torch.profiler attribution of the stock-torch kernels in one training step of the bench model
(which aten op / which input shapes the non-HIP kernels come from).  Usage: python tools/prof_ops.py [B] [layers] [size=1.5B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import apertis_llm_amd as A
from apertis_llm_amd.training import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
torch.manual_seed(0)
SIZE = sys.argv[3] if len(sys.argv) > 3 else "1.5B"
model = A.create_apertis_model(SIZE, vocab_size_override=32000, multimodal=False, use_expert_system=True,
                               attention_type_override="selective_ssm")
if len(sys.argv) > 2 and int(sys.argv[2]) > 0:
    model.model.layers = model.model.layers[:int(sys.argv[2])]
model = model.to(dev).train()
step = TrainStep(model, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0, total_steps=10, bf16=True)
ids = torch.randint(4, 32000, (B, 4096), device=dev)
batch = {"input_ids": ids, "attention_mask": torch.ones_like(ids), "labels": ids}
for _ in range(2):
    step(**batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(**batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=400, max_name_column_width=50,
                                                          max_shapes_column_width=70))
