"""Can the single-token decode step be captured in a HIP graph?  A 1.5B-family model with N layers: greedy tokens eager vs graph
replay, and the time per token step.  python tools/decode_graph_try.py [layers=2] [B=1] [steps=32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import apertis_llm_amd as A
from apertis_llm_amd import ops

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = A.ApertisConfig(vocab_size=32000, hidden_size=704, num_hidden_layers=layers, num_attention_heads=11, intermediate_size=2816,
                      attention_type="selective_ssm", use_expert_system=True, num_experts=8, experts_per_token=2)
model = A.ApertisForCausalLM(cfg).to(dev).eval()
prompt = torch.randint(4, 32000, (B, 256), device=dev)


def step_fn(tok, past):
    out = model(input_ids=tok, past_key_values=past, use_cache=True)
    return out[1][:, -1, :].float().argmax(-1, keepdim=True), out[4]


with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16), ops.prep_cache_scope():
    out = model(input_ids=prompt, use_cache=True)
    tok0 = out[1][:, -1, :].float().argmax(-1, keepdim=True)
    past0 = [(c.clone(), s.clone()) for (c, s) in out[4]]
    # eager
    tok, past, eager = tok0.clone(), past0, []
    for _ in range(4):
        tok, past = step_fn(tok, past)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tok, past = tok0.clone(), past0
    for _ in range(steps):
        tok, past = step_fn(tok, past)
        eager.append(tok.clone())
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / steps
    print(f"layers {layers} B {B}: eager {te * 1e3:.2f} ms per token step", flush=True)
    # graph: static token + cache buffers, the step's new cache copied back into them inside the graph
    s_tok = tok0.clone()
    s_past = [(c.clone(), s.clone()) for (c, s) in past0]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            n_tok, n_past = step_fn(s_tok, s_past)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print("capturing ...", flush=True)
    with torch.cuda.graph(g):
        n_tok, n_past = step_fn(s_tok, s_past)
        s_tok.copy_(n_tok)
        for (sc, ss), (nc, ns) in zip(s_past, n_past):
            sc.copy_(nc); ss.copy_(ns)
    torch.cuda.synchronize()
    print("captured; replaying ...", flush=True)
    s_tok.copy_(tok0)
    for (sc, ss), (c, s) in zip(s_past, past0):
        sc.copy_(c); ss.copy_(s)
    graph = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
        graph.append(s_tok.clone())
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / steps
    same = all(torch.equal(a, b) for a, b in zip(eager, graph))
    print(f"layers {layers} B {B}: graph replay {tg * 1e3:.2f} ms per token step ({1.0 / tg * B:.0f} tokens/s); tokens equal to eager: {same}", flush=True)
