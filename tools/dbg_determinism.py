import torch, sys
sys.path.insert(0, "/root/repo")
import apertis_llm_amd as A
CFG = dict(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
           attention_type="selective_ssm", use_expert_system=True, hidden_dropout_prob=0.0,
           attention_probs_dropout_prob=0.0, use_noisy_top_k_routing=False, use_expert_dropout=False)
dev = torch.device("cuda:0")
torch.manual_seed(0)
init = A.ApertisForCausalLM(A.ApertisConfig(**CFG)).state_dict()
g = torch.Generator().manual_seed(1000)
ids = torch.randint(4, 512, (2, 256), generator=g).to(dev)
def run():
    m = A.ApertisForCausalLM(A.ApertisConfig(**CFG)); m.load_state_dict(init); m = m.to(dev).train()
    out = m(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids)
    out[0].backward()
    return float(out[0]), {n: p.grad.detach().clone() for n, p in m.named_parameters()}
l0, g0 = run(); l1, g1 = run(); l2, g2 = run()
print("losses", l0, l1, l2)
for n in g0:
    d = float((g0[n] - g1[n]).abs().max()); mx = float(g0[n].abs().max())
    if d > 1e-5 * mx: print(n, d, mx)
