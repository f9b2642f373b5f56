import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
import apertis_llm_amd as A
from apertis_llm_amd import ops
from oracle import ref_cpu, seeded
dev = torch.device('cuda:0')
for gain in (1.0, 2.0):
  for seed in (31, 32):
    cfg = A.ApertisConfig(vocab_size=1024, hidden_size=704, num_hidden_layers=2, num_attention_heads=11, intermediate_size=2816,
                          attention_type="selective_ssm", use_expert_system=True, num_experts=8, experts_per_token=2, max_position_embeddings=512)
    model = A.ApertisForCausalLM(cfg)
    sd = {k: v.bfloat16().float() for k, v in seeded.fill_state_dict(model.state_dict(), gain=gain).items()}
    model.load_state_dict(sd); model = model.to(dev).eval()
    ids = torch.randint(4, cfg.vocab_size, (2, 512), generator=torch.Generator().manual_seed(seed))
    taken, orig = [], ops.moe_gate_topk
    def spy(*a, **k):
        r = orig(*a, **k); taken.append((r[0].detach().cpu().double(), r[1].detach().cpu().long())); return r
    ops.moe_gate_topk = spy
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        loss, logits = model(input_ids=ids.to(dev), labels=ids.to(dev), use_cache=False)[:2]
    ops.moe_gate_topk = orig
    aux_c, aux64 = [], []
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        loss_c, logits_c = ref_cpu.model_forward(sd, dict(cfg.to_dict()), ids, None, ids, aux_out=aux_c)
    with torch.no_grad():
        loss64, logits64 = ref_cpu.model_forward({k: v.double() for k, v in sd.items()}, dict(cfg.to_dict()), ids, None, ids, aux_out=aux64)
    for li in range(2):
        g64 = aux64[li]["gates"]; gh = taken[li][0]; gc = aux_c[li]["gates"].double()
        top3 = g64.topk(3, dim=-1).values; gap = top3[:, 1] - top3[:, 2]
        sh = (taken[li][1].reshape(-1, 2).sort(-1).values == aux64[li]["idx"].sort(-1).values).all(-1)
        sc = (aux_c[li]["idx"].sort(-1).values == aux64[li]["idx"].sort(-1).values).all(-1)
        print(f"gain {gain} seed {seed} layer {li}: gate err hip max {float((gh-g64).abs().max()):.2e} rms {float((gh-g64).pow(2).mean().sqrt()):.2e} | cpu-ac max {float((gc-g64).abs().max()):.2e} rms {float((gc-g64).pow(2).mean().sqrt()):.2e} | flips hip {int((~sh).sum())} (largest gap flipped {float(gap[~sh].max()) if (~sh).any() else 0:.2e}) cpu-ac {int((~sc).sum())} (largest {float(gap[~sc].max()) if (~sc).any() else 0:.2e}) | gate gap median {float(gap.median()):.3f}")
    def rr(a, b): return float((a.double().cpu()-b).pow(2).mean().sqrt()/b.pow(2).mean().sqrt())
    print(f"   logits rel rms: hip {rr(logits.float(), logits64):.3e} cpu-ac {rr(logits_c.float(), logits64):.3e}; loss {float(loss64):.5f} hip {float(loss):.5f} cpu {float(loss_c):.5f}")
