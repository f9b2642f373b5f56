"""BASELINE.json configurations at their STATED shapes on the GPU (configs_untested).

  config 1  125M (H896 / 10 layers / 14 heads / I3584, vocab 32000), B=2, L=512: loss and sampled logits captured from
            the reference (tests/golden/config1_125m.npz, weights rebuilt from oracle/seeded.py)
  config 2  125M-shaped SSM layer stack at L=2048, fp32 (rtol 1e-4) and under bf16 autocast (vs the oracle under CPU
            autocast: the reference's own dtype flow)
  config 3  350M MoE family (H=256, 4 heads -> Dn=64 = ONE 64-channel scan tile, R=16, I=1024, 8 experts top-2) at L=4096:
            a 2-layer model in fp32 eval vs the oracle (logits rtol 1e-4), and the MoE layer in train mode (capacity
            floor(S/8*1.25) = 640*B on, noise / dropout off) vs the oracle with the dropped-token set compared exactly
  config 5  1.5B multimodal shapes: patch-embed GEMM (B*196,768)@(768,768), vision_projection 768->704 on (B*197) rows,
            UnifiedMultimodalEncoder at 224^2 / patch 16, and a 2-layer H=704 / 11-head / 8-expert model on a 224^2 image +
            2048 text tokens (L = 2245 inside the layers: ragged 128/64-token scan chunks) vs the CPU oracle
Every logits comparison reports its achieved maximum relative error (conftest.rel_error_report) and holds it to the
1e-4 bar.  Reference lines: multimodal/module.py:35-40,102-110; core.py:1207-1227,1399-1406.
"""
import json

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_error_report

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ config 1
def test_config1_125m_ssm_matches_reference_capture(dev):
    import apertis_llm_amd as A
    from oracle import seeded
    g = load_golden("config1_125m")
    cfg = A.ApertisConfig.from_dict(json.loads(str(g["selective_ssm::config_json"])))
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size) == (896, 10, 14, 3584)
    model = A.ApertisForCausalLM(cfg)
    assert sum(p.numel() for p in model.parameters()) == int(g["selective_ssm::n_params"])
    model.load_state_dict(seeded.fill_state_dict(model.state_dict()))
    model = model.to(dev).eval()
    ids = g["input_ids"].to(dev)
    with torch.no_grad():
        loss, logits = model(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids)[:2]
    assert logits.shape == (2, 512, 32000)
    rel_error_report("config1_125m_ssm logits[:, ::37, ::251]", logits[:, ::37, ::251], g["selective_ssm::logits_sample"])
    assert abs(float(loss) - float(g["selective_ssm::loss"])) <= 1e-5 * float(g["selective_ssm::loss"])
    assert abs(float(logits.abs().max()) - float(g["selective_ssm::logits_absmax"])) <= 1e-4 * float(g["selective_ssm::logits_absmax"])


# ------------------------------------------------------------------------------------------------ config 2
def _cfg2(A, layers=2, vocab=1024):
    return A.ApertisConfig(vocab_size=vocab, hidden_size=896, num_hidden_layers=layers, num_attention_heads=14,
                           intermediate_size=3584, attention_type="selective_ssm", max_position_embeddings=2048)


def test_config2_layer_stack_fp32_L2048(dev):
    """H=896, 14 heads (Dn=224, R=56, x_param_proj width 504), dense FFN I=3584, L=2048, B=2, fp32: logits vs oracle."""
    import apertis_llm_amd as A
    from oracle import ref_cpu, seeded
    cfg = _cfg2(A)
    model = A.ApertisForCausalLM(cfg)
    sd = seeded.fill_state_dict(model.state_dict(), gain=2.0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    ids = torch.randint(4, cfg.vocab_size, (2, 2048), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        loss, logits = model(input_ids=ids.to(dev), labels=ids.to(dev), use_cache=False)[:2]
        o_loss, o_logits = ref_cpu.model_forward(sd, dict(cfg.to_dict()), ids, None, ids)
    rel_error_report("config2_fp32_L2048 logits", logits, o_logits)
    assert abs(float(loss) - float(o_loss)) <= 1e-5 * abs(float(o_loss))


def _err_vs(ref64, got):
    """Error of `got` against the fp64 yardstick: (relative RMS, max abs / max|ref|, 99.9th-percentile abs / max|ref|)."""
    r = ref64.detach().double().cpu().flatten()
    d = (got.detach().double().cpu().flatten() - r).abs()
    k = max(1, int(d.numel() * 0.999))
    return (float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()), float(d.max() / r.abs().max()),
            float(d.kthvalue(k).values / r.abs().max()))


def _bf16_report(name, ref64, hip, cpu_ac, factor=1.5, floor=2e-4):
    """Two mixed-precision runs of the same arithmetic - the HIP path under CUDA bf16 autocast and the oracle under CPU bf16
    autocast (the reference's own dtype flow, pipeline.py:533) - each measured against an fp64 run of the oracle on the same
    bf16-representable weights and inputs.  Held: the HIP path's error is at most `factor` x the CPU flow's own error (plus
    a floor of `floor` of the tensor's scale for quantities both flows get nearly exact), in RMS and at the 99.9th percentile;
    two bf16 paths are never compared with each other."""
    import json
    import os
    from conftest import ROOT
    eh, ec = _err_vs(ref64, hip), _err_vs(ref64, cpu_ac)
    rec = {"test": name, "dtype": "bf16-autocast", "yardstick": "fp64 oracle on bf16-rounded weights",
           "hip_rel_rms": eh[0], "hip_max_over_refmax": eh[1], "hip_p999_over_refmax": eh[2],
           "cpu_autocast_rel_rms": ec[0], "cpu_autocast_max_over_refmax": ec[1], "cpu_autocast_p999_over_refmax": ec[2],
           "factor": factor, "numel": int(ref64.numel())}
    print("PARITY " + json.dumps(rec))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    assert eh[0] <= factor * ec[0] + floor, rec
    assert eh[2] <= factor * ec[2] + floor, rec
    return rec


def test_config2_ssm_layer_bf16_autocast_L2048(dev):
    """One SelectiveLinearAttention at config-2 dims (H=896, 14 heads, L=2048) under bf16 autocast - the benchmark's dtype -
    forward and input gradient.  Yardstick: the oracle in fp64 on the same (bf16-representable) weights and input; the HIP
    path's error against it must not exceed 1.5 x the error of the oracle under CPU bf16 autocast, which is the reference's
    own mixed-precision flow (Linear / conv outputs bf16, softplus, exp and the state fp32).  Round 3 compared
    the two bf16 paths with each other (sig10 5.7e-2, unasserted)."""
    import apertis_llm_amd as A
    from oracle import ref_cpu, seeded
    cfg = _cfg2(A)
    mod = A.SelectiveLinearAttention(cfg)
    sd = {k: v.bfloat16().float() for k, v in seeded.fill_state_dict(mod.state_dict(), gain=2.0).items()}
    mod.load_state_dict(sd)
    mod = mod.to(dev).train()
    x = torch.randn(2, 2048, 896, generator=torch.Generator().manual_seed(4)).bfloat16().float()
    dout = torch.randn(2, 2048, 896, generator=torch.Generator().manual_seed(5)).bfloat16().float()
    xg = x.to(dev).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = mod(xg)[0]
    out.float().backward(dout.to(dev))
    xo = x.clone().requires_grad_(True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o = ref_cpu.ssm_layer(sd, "", xo, 14, 16, cfg.ssm_dt_rank)
    o.float().backward(dout)
    assert out.dtype == torch.bfloat16 and o.dtype == torch.bfloat16
    x64 = x.double().requires_grad_(True)
    o64 = ref_cpu.ssm_layer({k: v.double() for k, v in sd.items()}, "", x64, 14, 16, cfg.ssm_dt_rank)
    assert o64.dtype == torch.float64
    o64.backward(dout.double())
    _bf16_report("config2_bf16_L2048 ssm out", o64, out, o)
    _bf16_report("config2_bf16_L2048 ssm dx", x64.grad, xg.grad, xo.grad)


def test_bench_dtype_whole_model_bf16_autocast(dev):
    """The dtype the benchmark runs (pipeline.py:533: the reference trains under 16-bit autocast), whole model: 2 layers at the
    1.5B family's width (H=704, 11 heads, I=2816, 8 experts top-2; core.py:470-607), L=512, B=8, eval (dropout / noise /
    capacity off).  HIP logits, loss and input-embedding gradient under CUDA bf16 autocast against the fp64 oracle on the same
    bf16-representable weights, next to the oracle under CPU bf16 autocast (the reference's own flow).

    What bf16 does to such a model (tools/diag_bf16_routing.py): the first layer's gates move by ~1e-3, so tokens whose 2nd and
    3rd gate are closer than that change expert in EITHER bf16 flow (2-7 of 1024), and a token that changed expert feeds every
    later token of its sequence through the scan - the second layer then changes expert for gaps up to 0.1, in both flows alike,
    and the whole-tensor error (6-9 % RMS in both) measures those cascades, not arithmetic.  Held therefore:
      * first layer: the HIP routing equals the fp64 routing on every token whose 2nd / 3rd gate gap exceeds 1e-3;
      * every layer: the HIP flow changes expert on at most 1.5 x as many tokens as the CPU flow (+4);
      * logits on each sequence's CLEAN PREFIX - the tokens before the first token that changed expert in either flow (the
        model is causal: they depend on nothing that changed) - HIP error <= 1.5 x the CPU flow's, both at bf16 level;
      * whole tensors (logits, d inputs_embeds, loss): HIP error <= 1.5 x the CPU flow's (+ floor)."""
    import apertis_llm_amd as A
    from apertis_llm_amd import ops
    from oracle import ref_cpu, seeded
    B, L = 8, 512
    cfg = A.ApertisConfig(vocab_size=1024, hidden_size=704, num_hidden_layers=2, num_attention_heads=11, intermediate_size=2816,
                          attention_type="selective_ssm", use_expert_system=True, num_experts=8, experts_per_token=2,
                          max_position_embeddings=L)
    model = A.ApertisForCausalLM(cfg)
    sd = {k: v.bfloat16().float() for k, v in seeded.fill_state_dict(model.state_dict()).items()}
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    ids = torch.randint(4, cfg.vocab_size, (B, L), generator=torch.Generator().manual_seed(31))
    emb = F.embedding(ids, sd["model.token_embeddings.weight"])
    cfgd = dict(cfg.to_dict())

    taken, orig = [], ops.moe_gate_topk           # the routing the HIP forward takes, read off the gate op's outputs
    def spy(*a, **k):
        r = orig(*a, **k)
        taken.append(r[1].detach().cpu().long())
        return r
    ops.moe_gate_topk = spy
    try:
        eg = emb.to(dev).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss, logits = model(inputs_embeds=eg, labels=ids.to(dev), use_cache=False)[:2]
        loss.float().backward()
    finally:
        ops.moe_gate_topk = orig
    ec = emb.clone().requires_grad_(True)
    aux_c, aux64 = [], []
    with torch.autocast("cpu", dtype=torch.bfloat16):
        loss_c, logits_c = ref_cpu.model_forward(sd, cfgd, ids, None, ids, aux_out=aux_c, inputs_embeds=ec)
    loss_c.float().backward()
    e64 = emb.double().requires_grad_(True)
    loss64, logits64 = ref_cpu.model_forward({k: v.double() for k, v in sd.items()}, cfgd, ids, None, ids, aux_out=aux64,
                                             inputs_embeds=e64)
    assert logits64.dtype == torch.float64 and len(taken) == len(aux64) == len(aux_c) == 2
    loss64.backward()

    changed = torch.zeros(B * L, dtype=torch.bool)
    for li in range(2):
        ref_idx = aux64[li]["idx"].sort(dim=-1).values
        same_h = (taken[li].reshape(-1, 2).sort(dim=-1).values == ref_idx).all(dim=-1)
        same_c = (aux_c[li]["idx"].sort(dim=-1).values == ref_idx).all(dim=-1)
        top3 = aux64[li]["gates"].topk(3, dim=-1).values
        gap = top3[:, 1] - top3[:, 2]
        nh, nc = int((~same_h).sum()), int((~same_c).sum())
        print("whole_model_bf16 layer %d: tokens that change expert vs fp64: hip %d (largest gap %.2e), cpu-autocast %d (largest gap %.2e) of %d"
              % (li, nh, float(gap[~same_h].max()) if nh else 0.0, nc, float(gap[~same_c].max()) if nc else 0.0, B * L))
        if li == 0:
            assert bool(same_h[gap > 1e-3].all()), ("first layer", int((~same_h & (gap > 1e-3)).sum()))
        assert nh <= 1.5 * nc + 4, (li, nh, nc)
        changed |= ~same_h | ~same_c
    first = torch.where(changed.reshape(B, L).any(dim=1), changed.reshape(B, L).float().argmax(dim=1), torch.full((B,), L))
    clean = torch.arange(L).unsqueeze(0) < first.unsqueeze(1)            # [B, L]
    print("whole_model_bf16 clean prefixes:", first.tolist(), "=", int(clean.sum()), "tokens")
    assert int(clean.sum()) >= 64, first.tolist()
    r = _bf16_report("whole_model_bf16 logits, clean prefixes", logits64[clean], logits.float().cpu()[clean], logits_c.float()[clean])
    assert r["hip_rel_rms"] <= 2e-2, r                                   # bf16 level: nothing routed differently in here
    _bf16_report("whole_model_bf16 logits", logits64, logits.float(), logits_c.float(), floor=5e-3)
    _bf16_report("whole_model_bf16 d(inputs_embeds)", e64.grad, eg.grad, ec.grad, floor=5e-3)
    err_h, err_c = abs(float(loss) - float(loss64)), abs(float(loss_c) - float(loss64))
    print("whole_model_bf16 loss: fp64 %.6f  hip %.6f (err %.2e)  cpu-autocast %.6f (err %.2e)" % (float(loss64), float(loss), err_h, float(loss_c), err_c))
    assert err_h <= 1.5 * err_c + 1e-3 * abs(float(loss64)), (err_h, err_c)


# ------------------------------------------------------------------------------------------------ config 3
def _cfg3(A, layers=2, vocab=1024, **kw):
    return A.ApertisConfig(vocab_size=vocab, hidden_size=256, num_hidden_layers=layers, num_attention_heads=4,
                           intermediate_size=1024, attention_type="selective_ssm", use_expert_system=True, num_experts=8,
                           experts_per_token=2, max_position_embeddings=4096, **kw)


def test_config3_model_fp32_L4096(dev):
    """H=256, 4 heads (Dn=64: one channel tile, x_param_proj width 144 -> padded 192), I=1024, 8 experts top-2, L=4096,
    B=2, fp32 eval (no capacity: 16 384 assignments on [~2048]*8 rows): whole-model logits and loss vs the oracle."""
    import apertis_llm_amd as A
    from oracle import ref_cpu, seeded
    cfg = _cfg3(A)
    assert (cfg.ssm_d_inner, cfg.ssm_dt_rank) == (64, 16)
    model = A.ApertisForCausalLM(cfg)
    sd = seeded.fill_state_dict(model.state_dict(), gain=2.0)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    # data seed 21: the smallest gap between a token's 2nd and 3rd gate over both layers is 1.6e-5 (searched over 157
    # seeds with the CPU oracle; typical 1e-6 .. 1e-5) - ~100x the fp32 noise of a gate, so the routing is unambiguous
    ids = torch.randint(4, cfg.vocab_size, (2, 4096), generator=torch.Generator().manual_seed(21))
    aux = []
    with torch.no_grad():
        loss, logits = model(input_ids=ids.to(dev), labels=ids.to(dev), use_cache=False)[:2]
        o_loss, o_logits = ref_cpu.model_forward(sd, dict(cfg.to_dict()), ids, None, ids, aux_out=aux)
    # routing must be unambiguous for a logits comparison to mean anything: report the smallest top-2 / top-3 gap
    gaps = [float((a["gates"].topk(3, dim=-1).values[:, 1:].diff(dim=-1).abs()).min()) for a in aux]
    print("config3 min gate gap between the 2nd and 3rd choice per layer:", gaps)
    assert min(gaps) > 1e-5, gaps
    rel_error_report("config3_fp32_L4096 logits", logits, o_logits)
    assert abs(float(loss) - float(o_loss)) <= 1e-5 * abs(float(o_loss))


def test_config3_moe_layer_train_capacity_L4096(dev):
    """AdaptiveExpertSystem at config-3 dims in TRAIN mode, S = 2 x 4096 tokens, capacity floor(S/8*1.25) = 1280 per
    expert on (a skewed router bias makes experts overflow), noise / dropout / expert dropout off so that the kept set is
    deterministic: output (rtol 1e-4), both aux losses and the exact set of tokens dropped by every choice."""
    import apertis_llm_amd as A
    from oracle import ref_cpu, seeded
    cfg = _cfg3(A, hidden_dropout_prob=0.0, use_noisy_top_k_routing=False, use_expert_dropout=False)
    mod = A.AdaptiveExpertSystem(cfg, activation_function_override="gelu")
    sd = seeded.fill_state_dict(mod.state_dict(), gain=2.0)
    sd["router.bias"] = torch.tensor([0.9, 0.5, 0.0, 0.0, -0.2, 0.0, 0.3, -0.6])
    mod.load_state_dict(sd)
    mod = mod.to(dev).train()
    x = torch.randn(2, 4096, 256, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        out, lb, rz = mod(x.to(dev))
        o_out, o_lb, o_rz, aux = ref_cpu.moe_layer(sd, "", x, 8, 2, "gelu", cfg.layer_norm_eps, training=True)
    counts = (aux["offsets"][1:] - aux["offsets"][:-1]).tolist()
    assert max(counts) == 1280 and min(counts) < 1280, counts          # some experts overflow, some do not
    zero_ref = o_out.reshape(-1, 256).abs().sum(-1) == 0
    zero_got = out.cpu().reshape(-1, 256).abs().sum(-1) == 0
    assert int(zero_ref.sum()) > 0 and torch.equal(zero_ref, zero_got), "dropped-token set must match exactly"
    rel_error_report("config3 MoE layer (train, capacity) out", out, o_out)
    assert abs(float(lb) - float(o_lb)) <= 1e-5 * abs(float(o_lb)) and abs(float(rz) - float(o_rz)) <= 1e-5 * abs(float(o_rz))


# ------------------------------------------------------------------------------------------------ config 5
@pytest.mark.parametrize("rows,N,K", [(16 * 196, 768, 768), (16 * 197, 704, 768)])
def test_config5_vision_gemm_shapes(dev, rows, N, K):
    """The two GEMMs the vision path puts on the MFMA tile: patch embed (B*196,768)@(768,768)^T + b and
    vision_projection (B*197,768) -> 704, at B = 16.  fp32 path vs fp64 at rtol 1e-4; bf16 path vs fp64 on
    bf16-rounded operands (fp32 accumulation: only the output rounding remains)."""
    from apertis_llm_amd import ops
    g = torch.Generator().manual_seed(rows + N)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.05
    b = torch.randn(N, generator=g) * 0.1
    ref = F.linear(x.double(), w.double(), b.double())
    y32 = ops.linear_mfma(x.to(dev), w.to(dev), b.to(dev), compute_dtype=torch.float32)
    rel_error_report(f"config5 linear_mfma fp32 ({rows},{K})->({N})", y32, ref)
    xb, wb = x.bfloat16(), w.bfloat16()
    refb = F.linear(xb.double(), wb.double(), b.double())
    yb = ops.linear_mfma(xb.to(dev), w.to(dev), b.to(dev), compute_dtype=torch.bfloat16)
    assert yb.dtype == torch.bfloat16
    r = rel_error_report(f"config5 linear_mfma bf16 ({rows},{K})->({N})", yb.float(), refb, rtol=4e-3, check=False)
    assert r["max_abs_over_refmax"] <= 4e-3 and r["max_rel_sig10"] <= 8e-3, r       # one bf16 output rounding
    # backward of the fp32 path (dgrad NT + split-K wgrad TN) at this shape
    xg = x.to(dev).requires_grad_(True)
    wg, bg = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    dy = torch.randn(rows, N, generator=g)
    ops.linear_mfma(xg, wg, bg, compute_dtype=torch.float32).backward(dy.to(dev))
    rel_error_report("config5 linear_mfma fp32 dx", xg.grad, dy.double() @ w.double())
    rel_error_report("config5 linear_mfma fp32 dW", wg.grad, dy.double().t() @ x.double())
    rel_error_report("config5 linear_mfma fp32 db", bg.grad, dy.double().sum(0))


def test_config5_patch_embed_224_matches_conv2d(dev):
    """UnifiedMultimodalEncoder at its real geometry (224^2, patch 16, Dv 768): embed_patches == Conv2d(3,768,16,16)
    (module.py:35-40,102-103), and the whole encoder (12 stock-torch ViT layers) vs the oracle."""
    import apertis_llm_amd as A
    from oracle import ref_cpu, seeded
    cfg = A.ApertisConfig(hidden_size=704, num_attention_heads=11, multimodal=True)
    assert (cfg.image_size, cfg.vision_patch_size, cfg.vision_embed_dim, cfg.vision_layers, cfg.vision_heads) == (224, 16, 768, 12, 12)
    enc = A.UnifiedMultimodalEncoder(cfg)
    sd = seeded.fill_state_dict(enc.state_dict(), gain=2.0)
    enc.load_state_dict(sd)
    enc = enc.to(dev).eval()
    px = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        pe = enc.embed_patches(px.to(dev))
        feats = enc(px.to(dev))
    ref = F.conv2d(px.double(), sd["patch_embed.weight"].double(), sd["patch_embed.bias"].double(), stride=16)
    ref = ref.flatten(2).transpose(1, 2)                                        # [B, 196, 768], row = py*14 + px
    assert pe.shape == (4, 196, 768) and feats.shape == (4, 197, 768)
    rel_error_report("config5 embed_patches 224/p16 vs conv2d", pe, ref)
    with torch.no_grad():
        o_feats = ref_cpu.vision_encoder({k: v.double() for k, v in sd.items()}, "", px.double(), 16, 12, 12)
    rel_error_report("config5 encoder features (12 ViT layers, stock torch)", feats, o_feats, rtol=2e-4)


def test_config5_model_image_plus_2048_tokens(dev):
    """2 layers of the 1.5B multimodal shape (H=704, 11 heads, Dn=176, I=2816, 8 experts top-2, vocab 32000) on a
    224^2 image + 2048 text tokens: 197 image tokens are prepended (core.py:1207-1227), the layers see L = 2245 (17 full
    128-token scan chunks + a 69-token tail), logits are taken on the last 2048 positions (core.py:1399-1406)."""
    import apertis_llm_amd as A
    from oracle import ref_cpu, seeded
    cfg = A.ApertisConfig(vocab_size=32000, hidden_size=704, num_hidden_layers=2, num_attention_heads=11,
                          intermediate_size=2816, attention_type="selective_ssm", use_expert_system=True, num_experts=8,
                          experts_per_token=2, multimodal=True)
    model = A.ApertisForCausalLM(cfg)
    sd = seeded.fill_state_dict(model.state_dict())
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(4, 32000, (1, 2048), generator=g)
    px = torch.randn(1, 3, 224, 224, generator=g)
    aux = []
    with torch.no_grad():
        o_loss, o_logits = ref_cpu.model_forward(sd, dict(cfg.to_dict()), ids, px, ids, aux_out=aux)
        out = model(input_ids=ids.to(dev), pixel_values=px.to(dev), labels=ids.to(dev), use_cache=False)
    for a in aux:      # routing must not sit on a tie: a flipped expert choice is a discontinuity, not an error
        srt = torch.sort(a["gates"], dim=-1, descending=True).values
        assert a["gates"].shape[0] == 2245 and float((srt[:, :2] - srt[:, 1:3]).min()) > 5e-6
    assert out[1].shape == (1, 2048, 32000)
    rel_error_report("config5 model (224^2 + 2048 tokens, L=2245) logits", out[1], o_logits)
    assert abs(float(out[0]) - float(o_loss)) <= 1e-5 * abs(float(o_loss))
    # the same through bf16 autocast + backward (train mode, reference-default dropout / noise / capacity): finite
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        tl = model(input_ids=ids.to(dev), pixel_values=px.to(dev), labels=ids.to(dev))[0]
    tl.backward()
    assert torch.isfinite(tl)
    bad = [n for n, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not bad, bad[:5]
