"""Module- and model-level parity on the GPU against golden vectors captured from the reference
(tests/golden, tools/gen_golden.py) and against the CPU oracle's autograd.  fp32 everywhere:
the BASELINE tolerance is rtol 1e-4 on logits; router indices / permutation are exact."""
import json

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_error_report

pytestmark = pytest.mark.gpu


def _close(got, ref, name, rtol=1e-4, atol_scale=1e-5):
    ref = torch.as_tensor(ref).detach().cpu().to(torch.float64)
    got = got.detach().cpu().to(torch.float64)
    atol = atol_scale * float(ref.abs().max()) + 1e-30
    bad = (got - ref).abs() > atol + rtol * ref.abs()
    assert not bad.any(), f"{name}: {int(bad.sum())}/{bad.numel()} outside rtol {rtol}; max abs diff " \
                          f"{float((got - ref).abs().max()):.3e} (ref max {float(ref.abs().max()):.3e})"


def test_ssm_layer_golden(dev):
    import apertis_llm_amd as A
    g = load_golden("ssm_layer")
    cfg = A.ApertisConfig(hidden_size=48, num_attention_heads=3, ssm_d_state=16, attention_type="selective_ssm")
    mod = A.SelectiveLinearAttention(cfg)
    mod.load_state_dict(g["sd"])
    mod = mod.to(dev).eval()
    with torch.no_grad():
        out, y, cache = mod(g["x"].to(dev), output_attentions=True, use_cache=True)
    _close(out, g["out"], "out")
    rel_error_report("ssm_layer out vs reference capture", out, g["out"])
    _close(y, g["y_ssm"], "y_ssm")
    _close(cache[0], g["conv_state"], "conv_state")
    _close(cache[1].reshape(2, -1), g["ssm_state"], "ssm_state")


def test_ssm_layer_cached_decode_matches_reference_semantics(dev):
    """Prefill then one-token steps through the cache: compare with the oracle run the same way
    (state carried; conv window prepended and first-L outputs kept, reference core.py:369-373)."""
    import apertis_llm_amd as A
    from oracle import ref_cpu
    import torch.nn.functional as F
    g = load_golden("ssm_layer")
    cfg = A.ApertisConfig(hidden_size=48, num_attention_heads=3, ssm_d_state=16, attention_type="selective_ssm")
    mod = A.SelectiveLinearAttention(cfg)
    mod.load_state_dict(g["sd"])
    mod = mod.to(dev).eval()
    x = g["x"]
    sd = g["sd"]
    with torch.no_grad():
        _, _, cache = mod(x[:, :30].to(dev), use_cache=True)
        out1, _, cache = mod(x[:, 30:31].to(dev), past_key_value=cache, use_cache=True)
    # oracle: same two steps
    _, parts = ref_cpu.ssm_layer(sd, "", x[:, :30], 3, 16, int(g["dt_rank"]), return_parts=True)
    xp_all = F.linear(x, sd["in_proj_x.weight"])
    conv_prev = xp_all[:, 27:30]
    cat = torch.cat([conv_prev, xp_all[:, 30:31]], 1)
    xc = ref_cpu.dwconv_silu(cat, sd["conv1d.weight"], sd["conv1d.bias"])[:, :1]
    p = F.linear(xc, sd["x_param_proj.weight"])
    R = int(g["dt_rank"])
    delta = F.softplus(F.linear(p[..., :R], sd["dt_proj_head.weight"], sd["dt_proj_head.bias"]))
    y, hl = ref_cpu.scan_recurrent(delta, sd["A_log"], p[..., R:R + 48], p[..., R + 48:], parts["h_last"])
    o = F.linear((y + sd["D"] * xc) * F.silu(F.linear(x[:, 30:31], sd["in_proj_z.weight"])), sd["out_proj.weight"])
    _close(out1, o, "decode step out")
    _close(cache[1].reshape(2, -1), hl, "decode step state")


@pytest.mark.parametrize("name", ["moe_eval", "moe_train_overflow", "moe_eval_k3"])
def test_moe_layer_golden(dev, name):
    import apertis_llm_amd as A
    g = load_golden(name)
    E, K = int(g["E"]), int(g["K"])
    H = g["x"].shape[-1]
    cfg = A.ApertisConfig(hidden_size=H, intermediate_size=g["sd"]["experts.0.1.weight"].shape[0], num_attention_heads=2,
                          use_expert_system=True, num_experts=E, experts_per_token=K, hidden_dropout_prob=0.0,
                          use_noisy_top_k_routing=False, use_expert_dropout=False)
    mod = A.AdaptiveExpertSystem(cfg, activation_function_override="gelu")
    missing = mod.load_state_dict(g["sd"], strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"w_noise"}
    mod = mod.to(dev).train(bool(int(g["training"])))
    with torch.no_grad():
        out, lb, rz = mod(g["x"].to(dev))
    _close(out, g["out"], "out", rtol=1e-4, atol_scale=2e-5)
    rel_error_report(f"{name} layer out vs reference capture", out, g["out"])
    _close(lb, g["lb"], "lb_loss", rtol=1e-5)
    _close(rz, g["rz"], "rz_loss", rtol=1e-5)
    if int(g["training"]):
        assert float((g["out"].reshape(-1, H).abs().sum(-1) == 0).sum()) > 0   # some tokens dropped by capacity
        zero_ref = g["out"].reshape(-1, H).abs().sum(-1) == 0
        zero_got = out.cpu().reshape(-1, H).abs().sum(-1) == 0
        assert torch.equal(zero_ref, zero_got), "dropped-token set must match exactly"


def test_vision_golden(dev):
    import apertis_llm_amd as A
    g = load_golden("vision")
    cfg = A.ApertisConfig(hidden_size=48, num_attention_heads=3, multimodal=True, image_size=32, vision_embed_dim=32,
                          vision_patch_size=8, vision_layers=2, vision_heads=2)
    enc = A.UnifiedMultimodalEncoder(cfg)
    enc.load_state_dict(g["sd"])
    enc = enc.to(dev).eval()
    from apertis_llm_amd import ops
    with torch.no_grad():
        pe = enc.embed_patches(g["pixel_values"].to(dev))
        feats = enc(g["pixel_values"].to(dev))
        pr = ops.linear_mfma(feats, g["proj_weight"].to(dev), g["proj_bias"].to(dev))
    _close(pe, g["patch_embeds"], "patch_embeds")
    _close(feats, g["features"], "encoder features", rtol=2e-4, atol_scale=2e-5)
    _close(pr, g["projected"], "vision_projection", rtol=2e-4, atol_scale=2e-5)


@pytest.mark.parametrize("name", ["model_ssm_dense", "model_ssm_moe", "model_ssm_moe_mm"])
def test_model_logits_and_loss_golden(dev, name):
    import apertis_llm_amd as A
    g = load_golden(name)
    cfg = A.ApertisConfig.from_dict(json.loads(str(g["config_json"])))
    model = A.ApertisForCausalLM(cfg)
    model.load_state_dict(g["sd"])
    model = model.to(dev).eval()
    px = g["pixel_values"].to(dev) if "pixel_values" in g else None
    with torch.no_grad():
        out = model(input_ids=g["input_ids"].to(dev), pixel_values=px, labels=g["labels"].to(dev), use_cache=False)
    assert len(out) == 7
    _close(out[1], g["logits"], "logits", rtol=1e-4, atol_scale=2e-5)
    rel_error_report(f"{name} logits vs reference capture", out[1], g["logits"])       # achieved max relative error
    assert abs(float(out[0]) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    # default call path (use_cache from the config = True) returns the per-layer caches too
    with torch.no_grad():
        out_c = model(input_ids=g["input_ids"].to(dev), pixel_values=px)
    _close(out_c[1], g["logits"], "logits (use_cache default)", rtol=1e-4, atol_scale=2e-5)
    assert out_c[4] is not None and len(out_c[4]) == cfg.num_hidden_layers


def test_model_gradients_vs_oracle_autograd(dev):
    """Backward of the whole model (HIP scan/conv/gate/MoE backward kernels) against torch
    autograd through the CPU oracle, fp32, eval mode (no dropout/noise/capacity)."""
    import apertis_llm_amd as A
    from oracle import ref_cpu
    g = load_golden("model_ssm_moe")
    cfgd = json.loads(str(g["config_json"]))
    cfg = A.ApertisConfig.from_dict(cfgd)
    model = A.ApertisForCausalLM(cfg)
    model.load_state_dict(g["sd"])
    model = model.to(dev).eval()
    out = model(input_ids=g["input_ids"].to(dev), labels=g["labels"].to(dev), use_cache=False)
    out[0].backward()
    sd = {k: v.clone().double().requires_grad_(True) for k, v in g["sd"].items() if k != "lm_head.weight"}
    loss, _ = ref_cpu.model_forward(sd, cfgd, g["input_ids"], None, g["labels"])
    loss.backward()
    ours = {}
    for k, p in model.named_parameters():
        ours[k] = p.grad
    msd = model.state_dict(keep_vars=False)
    checked, worst = 0, 0.0
    for k, ref in sd.items():
        if ref.grad is None:
            continue
        if ".experts." in k:
            pre, rest = k.split(".experts.")
            e, suffix = rest.split(".", 1)
            stacked = dict(model.ffn_stack_names if hasattr(model, "ffn_stack_names") else A.AdaptiveExpertSystem._STACKED)[suffix]
            got = ours[f"{pre}.{stacked}"][int(e)]
        else:
            got = ours.get(k)
        if got is None:
            assert float(ref.grad.abs().max()) == 0.0, k
            continue
        # achieved error on the record (profiles/r3_parity_report.jsonl: worst 1.05e-5 relative on the significant elements,
        # worst excess 0.16 of this bound); the bound asserted is the BASELINE bar for logits, held for every gradient too:
        # rtol 1e-4 with an absolute floor of 1e-5 of the tensor's largest entry (fp32 kernels vs fp64 autograd through the oracle)
        rec = rel_error_report("grad " + k, got, ref.grad, rtol=1e-4, atol_scale=1e-5, check=False)
        worst = max(worst, rec["worst_excess"])
        assert rec["worst_excess"] <= 1.0, rec
        checked += 1
    print(f"whole-model gradients: {checked} tensors, worst excess over (rtol 1e-4, atol 1e-5*max) = {worst:.3f}")
    assert checked > 40


def test_train_step_bf16_autocast_runs(dev):
    """One training step under bf16 autocast with reference-default dropout/noise/capacity."""
    import apertis_llm_amd as A
    torch.manual_seed(0)
    cfg = A.ApertisConfig(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, attention_type="selective_ssm", use_expert_system=True)
    model = A.ApertisForCausalLM(cfg).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    ids = torch.randint(4, 512, (2, 256), device=dev)
    losses = []
    for _ in range(3):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(input_ids=ids, labels=ids)[0]
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("dt,B,L,V,Ll", [(torch.bfloat16, 3, 17, 512, 17), (torch.float32, 2, 9, 100, 9),
                                          (torch.bfloat16, 2, 33, 32000, 33), (torch.bfloat16, 2, 12, 256, 8),
                                          (torch.float32, 1, 2, 8, 2)])
def test_shifted_cross_entropy_matches_reference_formula(dev, dt, B, L, V, Ll):
    """ops.shifted_cross_entropy vs the reference's shift + CrossEntropyLoss(ignore_index=-100) in fp32
    (core.py:1407-1416) on the same stored logits: loss to 1e-5, gradient to the output dtype's rounding."""
    from apertis_llm_amd import ops
    torch.manual_seed(B * 1000 + L + V)
    logits = (torch.randn(B, L, V) * 3).to(dt)
    labels = torch.randint(0, V, (B, Ll))
    if Ll > 4:
        labels[0, 3] = -100
    if Ll > 5:
        labels[-1, 5] = -100
    n = min(L, Ll) - 1
    ref_in = logits.float().clone().requires_grad_(True)
    ref = F.cross_entropy(ref_in[:, :n].reshape(-1, V), labels[:, 1:n + 1].reshape(-1), ignore_index=-100)
    (ref * 1.7).backward()
    x = logits.to(dev).requires_grad_(True)
    assert ops.shifted_cross_entropy_supported(x, labels.to(dev))
    loss = ops.shifted_cross_entropy(x, labels.to(dev))
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert x.grad.dtype == dt and x.grad.shape == x.shape
    tol = 1e-6 if dt == torch.float32 else 2 ** -8
    diff = (x.grad.float().cpu() - ref_in.grad).abs()
    assert float((diff - tol * ref_in.grad.abs()).max()) <= 1e-7 + (0 if dt == torch.float32 else 1e-9), float(diff.max())
    assert float(x.grad[:, n:].abs().max()) == 0.0     # positions without a target get an exact zero gradient


@pytest.mark.parametrize("dt,B,L,H,V,Ll", [(torch.float32, 3, 33, 64, 32000, 33), (torch.bfloat16, 2, 257, 128, 32000, 257),
                                            (torch.float32, 2, 9, 32, 100, 12), (torch.bfloat16, 1, 2, 64, 512, 2)])
def test_linear_cross_entropy_matches_lm_head_then_loss(dev, dt, B, L, H, V, Ll):
    """N4: ops.linear_cross_entropy (LM head + shifted CE one sequence at a time, no [B, L, V] logits tensor, core.py:1412-1450)
    against F.linear -> shift -> CrossEntropyLoss(ignore_index=-100) on the same (dtype-rounded) operands, at the
    reference's vocabulary size: loss 1e-5 (fp32) / 2e-3 (bf16 logits), d hidden and d W to the compute dtype's rounding."""
    from apertis_llm_amd import ops
    torch.manual_seed(B * 100 + L + H)
    h = (torch.randn(B, L, H) * 0.7).to(dt)
    W = torch.randn(V, H) * 0.1
    labels = torch.randint(0, V, (B, Ll))
    if Ll > 4:
        labels[0, 3] = -100
    n = min(L, Ll) - 1
    hr, Wr = h.double().clone().requires_grad_(True), W.to(dt).double().clone().requires_grad_(True)
    ref = F.cross_entropy(F.linear(hr, Wr)[:, :n].reshape(-1, V), labels[:, 1:n + 1].reshape(-1), ignore_index=-100)
    (ref * 1.3).backward()
    hd, Wd = h.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
    assert ops.linear_cross_entropy_supported(hd, Wd, labels.to(dev))
    loss = ops.linear_cross_entropy(hd, Wd, labels.to(dev), compute_dtype=dt)
    (loss * 1.3).backward()
    ltol = 1e-5 if dt == torch.float32 else 2e-3
    assert abs(float(loss) - float(ref)) <= ltol * max(1.0, abs(float(ref))), (float(loss), float(ref))
    rtol, asc = (1e-4, 1e-5) if dt == torch.float32 else (3e-2, 2e-2)       # bf16: logits, dlogits and the outputs are rounded
    _close(hd.grad, hr.grad, "d hidden", rtol=rtol, atol_scale=asc)
    _close(Wd.grad, Wr.grad, "d W", rtol=rtol, atol_scale=asc)
    assert hd.grad.dtype == dt and Wd.grad.dtype == torch.float32
    assert float(hd.grad[:, n:].abs().max()) == 0.0     # positions without a target get an exact zero gradient


@pytest.mark.parametrize("dt,B,L,V,Ll", [(torch.bfloat16, 3, 37, 32000, 37), (torch.float32, 2, 19, 16384, 24),
                                         (torch.bfloat16, 2, 9, 512, 9), (torch.float32, 1, 5, 100, 5)])
def test_cross_entropy_one_pass_equals_the_two_kernels(dev, dt, B, L, V, Ll):
    """apertis_cross_entropy_fwd_bwd (round 6: log-sum-exp, loss and softmax - onehot of a row in one launch, the row kept in
    registers between the sweeps) against apertis_cross_entropy_fwd + apertis_cross_entropy_bwd on the same logits: lse,
    row losses and the gradient bit for bit - out of place and IN PLACE (dlogits = logits, as ops.linear_cross_entropy calls
    it) -, ignored / out-of-range / beyond-n_pos rows zero; a vocabulary too wide for the registers is declined (-2)."""
    from apertis_llm_amd import _lib
    from apertis_llm_amd._lib import ptr, stream_ptr, dtype_code
    lib = _lib.load()
    torch.manual_seed(V + L)
    logits = (torch.randn(B, L, V, device=dev) * 3).to(dt)
    labels = torch.randint(0, V, (B, Ll), device=dev)
    labels[0, 2] = -100
    if Ll > 4:
        labels[B - 1, 4] = V + 5            # out of range: skipped like an ignored one
    n_pos = min(L, Ll) - 1
    gscale = torch.tensor([0.37], device=dev)
    code = dtype_code(logits)
    lse_a, loss_a = torch.empty(B * L, device=dev), torch.empty(B * L, device=dev)
    d_a = torch.empty_like(logits)
    assert lib.apertis_cross_entropy_fwd(ptr(logits), ptr(labels), ptr(lse_a), ptr(loss_a), B, L, V, Ll, n_pos, -100, code, stream_ptr()) == 0
    assert lib.apertis_cross_entropy_bwd(ptr(logits), ptr(labels), ptr(lse_a), ptr(gscale), ptr(d_a), B, L, V, Ll, n_pos, -100, code,
                                         stream_ptr()) == 0
    lse_b, loss_b = torch.full((B * L,), 7.0, device=dev), torch.full((B * L,), 7.0, device=dev)
    d_b = torch.full_like(logits, 5.0)
    assert lib.apertis_cross_entropy_fwd_bwd(ptr(logits), ptr(labels), ptr(lse_b), ptr(loss_b), ptr(gscale), ptr(d_b), B, L, V, Ll,
                                             n_pos, -100, code, stream_ptr()) == 0
    assert torch.equal(lse_a, lse_b) and torch.equal(loss_a, loss_b) and torch.equal(d_a, d_b)
    inplace = logits.clone()
    lse_c, loss_c = torch.empty(B * L, device=dev), torch.empty(B * L, device=dev)
    assert lib.apertis_cross_entropy_fwd_bwd(ptr(inplace), ptr(labels), ptr(lse_c), ptr(loss_c), ptr(gscale), ptr(inplace), B, L, V, Ll,
                                             n_pos, -100, code, stream_ptr()) == 0
    assert torch.equal(inplace, d_a) and torch.equal(lse_c, lse_a) and torch.equal(loss_c, loss_a)
    assert float(d_b[0, 1].abs().max()) == 0.0 and float(d_b[:, n_pos:].abs().max()) == 0.0   # (target labels[0, 2] ignored)
    # against torch on the live rows
    ref = F.cross_entropy(logits.float()[:, :n_pos].reshape(-1, V), torch.where(labels[:, 1:n_pos + 1] >= V, -100,
                          labels[:, 1:n_pos + 1]).reshape(-1), ignore_index=-100, reduction="none").reshape(B, n_pos)
    assert torch.allclose(loss_b.reshape(B, L)[:, :n_pos], ref, rtol=2e-6, atol=2e-6)
    wide = 40000 if dt == torch.bfloat16 else 20000
    big = torch.zeros(1, 2, wide, device=dev, dtype=dt)
    lab2 = torch.zeros(1, 2, dtype=torch.int64, device=dev)
    assert lib.apertis_cross_entropy_fwd_bwd(ptr(big), ptr(lab2), ptr(lse_b), ptr(loss_b), ptr(gscale), ptr(big), 1, 2, wide, 2, 1, -100,
                                             dtype_code(big), stream_ptr()) == -2


@pytest.mark.parametrize("B,L,chunk_rows", [(5, 48, 96), (2, 1100, 16384)])
def test_linear_cross_entropy_weight_gradient_on_the_library_kernel(dev, monkeypatch, B, L, chunk_rows):
    """N4 at the bench's LM-head width (H = 704, V = 32000: apertis_grouped_gemm_tn_dense_variant accepts the shape), where the
    chunk's weight gradient dl.T @ x runs on the library's wide-tile TN kernel with fp32 partial sums (round 6) instead of a
    stock bf16 GEMM: three chunks with a short last one (first chunk written, later ones added; the 128 x 128 kernel: few rows)
    and one chunk of 2 200 rows (the wide-tile kernel with its workspace).  Against the fp64 reference as the test above, and
    against the stock-GEMM path of the same op (switch off): same loss bit for bit, d hidden bit for bit (its GEMM is untouched),
    d W to the bf16 rounding the stock path adds per chunk."""
    from apertis_llm_amd import ops
    from apertis_llm_amd import _lib as L_
    H, V = 704, 32000
    assert L_.load().apertis_grouped_gemm_tn_dense_variant(V, H) >= 0
    monkeypatch.setattr(ops.loss, "_LCE_CHUNK_ROWS", chunk_rows)
    torch.manual_seed(B * 1000 + L)
    h = (torch.randn(B, L, H) * 0.7).bfloat16()
    W = torch.randn(V, H) * 0.05
    labels = torch.randint(0, V, (B, L))
    labels[0, 3] = -100
    n = L - 1
    hr, Wr = h.double().clone().requires_grad_(True), W.bfloat16().double().clone().requires_grad_(True)
    ref = F.cross_entropy(F.linear(hr, Wr)[:, :n].reshape(-1, V), labels[:, 1:n + 1].reshape(-1), ignore_index=-100)
    (ref * 1.3).backward()
    out = {}
    for own in (True, False):
        monkeypatch.setattr(ops.loss, "LCE_OWN_WGRAD", own)
        hd, Wd = h.to(dev).requires_grad_(True), W.to(dev).requires_grad_(True)
        loss = ops.linear_cross_entropy(hd, Wd, labels.to(dev), compute_dtype=torch.bfloat16)
        (loss * 1.3).backward()
        out[own] = (loss.detach(), hd.grad.clone(), Wd.grad.clone())
    loss, dh, dW = out[True]
    assert abs(float(loss) - float(ref)) <= 2e-3 * max(1.0, abs(float(ref))), (float(loss), float(ref))
    _close(dh, hr.grad, "d hidden", rtol=3e-2, atol_scale=2e-2)
    _close(dW, Wr.grad, "d W", rtol=3e-2, atol_scale=2e-2)
    assert dW.dtype == torch.float32 and float(dh[:, n:].abs().max()) == 0.0
    assert torch.equal(loss, out[False][0]) and torch.equal(dh, out[False][1])
    _close(dW, out[False][2], "d W vs the stock-GEMM path", rtol=2e-2, atol_scale=1e-2)
    # fp32 partial sums are the closer ones to the fp64 reference
    e_own = float((dW.double().cpu() - Wr.grad).abs().max()), float((out[False][2].double().cpu() - Wr.grad).abs().max())
    assert e_own[0] <= e_own[1] * 1.05 + 1e-12, e_own


def test_fused_lm_head_loss_equals_the_logits_path(dev):
    """The model with fused_lm_head_loss (what TrainStep / ApertisTrainer switch on) against its own logits path: same loss,
    same parameter gradients (fp32, 1e-5), and the logits slot of the 7-tuple is None."""
    import apertis_llm_amd as A
    torch.manual_seed(0)
    cfg = A.ApertisConfig(vocab_size=512, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                          attention_type="selective_ssm", use_expert_system=True, num_experts=4, hidden_dropout_prob=0.0,
                          attention_probs_dropout_prob=0.0, use_noisy_top_k_routing=False, use_expert_dropout=False)
    model = A.ApertisForCausalLM(cfg).to(dev).train()
    ids = torch.randint(4, 512, (3, 40), device=dev)
    labels = ids.clone()
    labels[1, :5] = -100
    out_a = model(input_ids=ids, labels=labels)
    out_a[0].backward()
    ga = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    model.fused_lm_head_loss = True
    out_b = model(input_ids=ids, labels=labels)
    assert out_b[1] is None and len(out_b) == 7
    out_b[0].backward()
    assert abs(float(out_a[0]) - float(out_b[0])) <= 1e-6 * abs(float(out_a[0]))
    for n, p in model.named_parameters():
        _close(p.grad, ga[n], "grad " + n, rtol=1e-4, atol_scale=1e-5)


def test_shifted_cross_entropy_all_ignored_is_nan_like_torch(dev):
    from apertis_llm_amd import ops
    logits = torch.randn(1, 4, 16, device=dev)
    labels = torch.full((1, 4), -100, device=dev)
    assert torch.isnan(ops.shifted_cross_entropy(logits, labels))


def test_trainer_whole_run_matches_reference(dev, tmp_path):
    """§8(f) N2: a complete run of the reference's ApertisTrainer (captured on the CPU in fp32, dropout and routing noise
    off, DataLoader order fixed: tools/gen_golden.py gen_trainer_run) against this trainer on the GPU from the same initial
    weights and files: loss of every optimizer step, learning rates, validation losses, checkpoint directory layout,
    config.json key set and the final weights."""
    import numpy as np
    import apertis_llm_amd as A
    from apertis_llm_amd import data as D
    from apertis_llm_amd.trainer import ApertisTrainer
    g = load_golden("trainer_run")
    meta = json.loads(bytes(g["meta"].numpy().astype(np.uint8)).decode())
    init = {k[6:]: v for k, v in g.items() if isinstance(k, str) and k.startswith("init::")}
    final = {k[7:]: v for k, v in g.items() if isinstance(k, str) and k.startswith("final::")}
    vpath = tmp_path / "vocab.json"
    vpath.write_text(json.dumps(meta["vocab"]))
    (tmp_path / "train.jsonl").write_text("\n".join(meta["train_lines"]) + "\n")
    (tmp_path / "val.jsonl").write_text("\n".join(meta["val_lines"]) + "\n")
    vocab, n = D.load_vocabulary(str(vpath))
    tk = meta["trainer"]
    tr = D.ApertisPretrainDataset(str(tmp_path / "train.jsonl"), vocab, n, max_length=tk["max_length"])
    va = D.ApertisPretrainDataset(str(tmp_path / "val.jsonl"), vocab, n, max_length=tk["max_length"])
    model = A.ApertisForCausalLM(A.ApertisConfig(**meta["cfg"]))
    model.load_state_dict(init)
    out = str(tmp_path / "out")
    t = ApertisTrainer(model, tr, va, output_dir=out, batch_size=tk["batch_size"], learning_rate=tk["learning_rate"],
                       num_epochs=tk["num_epochs"], gradient_accumulation_steps=tk["gradient_accumulation_steps"], fp16=False,
                       device=str(dev), checkpoint_steps=tk["checkpoint_steps"],
                       iteration_checkpoint_steps=tk["iteration_checkpoint_steps"], use_gradient_checkpointing=False,
                       original_manual_vocab_path_for_ft=str(vpath), shuffle=False, num_workers=0)
    t.train()
    assert np.allclose(t.history["lr"], meta["lrs"], rtol=1e-9)
    assert len(t.history["loss"]) == len(meta["losses_4dp"])
    for a, b in zip(t.history["loss"], meta["losses_4dp"]):     # the reference keeps 4 decimals (its progress bar)
        assert abs(a - b) <= 6e-4, (t.history["loss"], meta["losses_4dp"])
    for a, b in zip(t.history["val_loss"], meta["val_losses"]):
        assert abs(a - b) <= 2e-4 * abs(b), (t.history["val_loss"], meta["val_losses"])
    import os
    listing = {name: sorted(os.listdir(os.path.join(out, name))) for name in sorted(os.listdir(out))}
    assert listing == meta["listing"]
    assert sorted(json.load(open(os.path.join(out, "final", "config.json")))) == meta["config_keys"]
    sd = torch.load(os.path.join(out, "final", "pytorch_model.bin"), map_location="cpu", weights_only=True)
    assert sorted(sd) == sorted(final)
    # Adam normalises every element's update to ~lr, so an element whose gradient is rounding noise moves by +-lr on
    # either side: compare the tensors' updates in the l2 sense, not element by element
    worst = 0.0
    for k, v in final.items():
        moved = (v.double() - init[k].double()).norm().item()
        err = (sd[k].double() - v.double()).norm().item()
        assert err <= 0.01 * moved + 1e-6, (k, err, moved)
        worst = max(worst, err / max(moved, 1e-12))
    print("worst relative l2 error of a tensor's update:", worst)


def test_adamw_kernels_match_torch(dev):
    """apertis_grad_sumsq / apertis_clip_coef / apertis_adamw_step (ApertisAdamW.step(max_grad_norm)) against
    torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over several steps: two parameter groups with different weight
    decay, tensor sizes that are not multiples of the vector width or of the chunk, a changing learning rate, a step
    where the clip is active and one where it is not."""
    from apertis_llm_amd.training import ApertisAdamW
    torch.manual_seed(3)
    shapes = [(5,), (1023,), (16384,), (16385,), (3, 7, 11), (70001,), (1,), (256, 130)]
    mine = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]

    def groups(ps):
        return [{"params": ps[:5], "weight_decay": 0.01}, {"params": ps[5:], "weight_decay": 0.0}]
    o1, o2 = ApertisAdamW(groups(mine), lr=1e-2), torch.optim.AdamW(groups(ref), lr=1e-2)
    norms = []
    for it in range(5):
        scale = 10.0 if it % 2 == 0 else 1e-3           # clip active / inactive
        for a, b in zip(mine, ref):
            g = torch.randn_like(a) * scale
            a.grad, b.grad = g.clone(), g.clone()
        for o in (o1, o2):
            for gr in o.param_groups:
                gr["lr"] = 1e-2 / (it + 1)
        o1.step(max_grad_norm=1.0)
        norms.append((float(o1.last_grad_norm), float(torch.nn.utils.clip_grad_norm_(ref, 1.0))))
        o2.step()
    for a, b in norms:
        assert abs(a - b) <= 1e-5 * b, norms
    for a, b, s in zip(mine, ref, shapes):
        _close(a.detach(), b.detach(), f"param {s}", rtol=2e-6, atol_scale=1e-6)
        _close(o1.state[a]["exp_avg"], o2.state[b]["exp_avg"], f"exp_avg {s}", rtol=2e-6, atol_scale=1e-6)
        _close(o1.state[a]["exp_avg_sq"], o2.state[b]["exp_avg_sq"], f"exp_avg_sq {s}", rtol=2e-6, atol_scale=1e-6)
        assert float(o1.state[a]["step"]) == 5.0
    # state dicts are interchangeable with torch.optim.AdamW's
    o3 = torch.optim.AdamW(groups([p.detach().clone().requires_grad_(True) for p in mine]), lr=1e-2)
    o3.load_state_dict(o1.state_dict())
    assert sorted(o3.state_dict()["state"][0]) == ["exp_avg", "exp_avg_sq", "step"]


def test_adamw_load_state_dict_late_parameters_and_nan_norm(dev):
    """(i) load_state_dict() after a step must make the kernels use the LOADED moments (the device tables
    cache raw pointers); (ii) a parameter that first receives a gradient later gets ITS OWN bias correction (torch keeps
    `step` per parameter); (iii) a non-finite gradient norm propagates (clip_grad_norm_ hands NaN through), it is not
    turned into an unclipped step."""
    from apertis_llm_amd.training import ApertisAdamW
    torch.manual_seed(5)
    shapes = [(33,), (4097,), (64, 9)]
    mine = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    ref = [p.detach().clone().requires_grad_(True) for p in mine]
    o1, o2 = ApertisAdamW(mine, lr=1e-2), torch.optim.AdamW(ref, lr=1e-2)

    def both(active, scale=1.0):
        for i, (a, b) in enumerate(zip(mine, ref)):
            if i in active:
                g = torch.randn_like(a) * scale
                a.grad, b.grad = g.clone(), g.clone()
            else:
                a.grad = b.grad = None
        o1.step(max_grad_norm=1.0)
        torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], 1.0)
        o2.step()

    both({0, 1})            # parameter 2 gets no gradient for two steps ...
    both({0, 1})
    both({0, 1, 2})         # ... then joins: its step must be 1, the others' 3
    assert [float(o1.state[p]["step"]) for p in mine] == [3.0, 3.0, 1.0]
    for a, b, s in zip(mine, ref, shapes):
        _close(a.detach(), b.detach(), f"late-parameter run, param {s}", rtol=2e-6, atol_scale=1e-6)
    # (i) swap in a state dict with different moments: the next step must start from them
    import copy
    sd = copy.deepcopy(o2.state_dict())       # load_state_dict adopts tensors of matching dtype/device without a copy
    for st in sd["state"].values():
        st["exp_avg"] = st["exp_avg"] * 3.0 + 0.5
        st["exp_avg_sq"] = st["exp_avg_sq"] * 2.0 + 0.25
    o1.load_state_dict(copy.deepcopy(sd))
    o2.load_state_dict(copy.deepcopy(sd))
    both({0, 1, 2})
    for a, b, s in zip(mine, ref, shapes):
        _close(a.detach(), b.detach(), f"after load_state_dict, param {s}", rtol=2e-6, atol_scale=1e-6)
        _close(o1.state[a]["exp_avg"], o2.state[b]["exp_avg"], f"exp_avg {s}", rtol=2e-6, atol_scale=1e-6)
    # (iii) NaN gradient: torch's clip makes every gradient NaN, so does the kernel's coefficient
    for a in mine:
        a.grad = torch.randn_like(a)
    mine[1].grad[7] = float("nan")
    o1.step(max_grad_norm=1.0)
    assert torch.isnan(o1.last_grad_norm)
    assert all(torch.isnan(p).all() for p in mine)


def test_ops_refuse_a_tensor_on_a_non_current_device(dev, monkeypatch):
    """the library launches on the CURRENT HIP device/stream; an op whose tensors live on another
    device must raise instead of launching on the wrong GPU.  One card on this box: pretend another device is current."""
    from apertis_llm_amd import ops
    from apertis_llm_amd._lib import ApertisHipError
    x = torch.randn(4, 64, device=dev)
    w, b = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    ops.layer_norm(x, w, b, 1e-5)                                  # fine on the current device
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    with pytest.raises(ApertisHipError, match="current device"):
        ops.layer_norm(x, w, b, 1e-5)


@pytest.mark.parametrize("name", ["generate_ssm_dense", "generate_ssm_moe"])
def test_generate_matches_reference_tokens_and_step_logits(dev, name):
    """N1: greedy generate() through the cache (prefill on the chunk kernels, then 15 single-token steps on the decode
    kernels apertis_ssm_decode_conv / apertis_ssm_decode_state) against the token sequence and the per-step logits captured
    from the reference's generate() (core.py:1520-1644; the cached conv window is front-sliced there, core.py:369-373, and
    here).  Tokens must be equal; logits within 1e-4."""
    import apertis_llm_amd as A
    g = load_golden(name)
    cfg = A.ApertisConfig.from_dict(json.loads(str(g["config_json"])))
    model = A.ApertisForCausalLM(cfg)
    model.load_state_dict(g["sd"])
    model = model.to(dev).eval()
    steps = []
    fwd = model.forward

    def spy(*a, **k):
        out = fwd(*a, **k)
        steps.append(out[1][:, -1, :].detach().float().cpu())
        return out
    model.forward = spy
    toks = model.generate(input_ids=g["prompt"].to(dev), max_new_tokens=16, do_sample=False, use_cache=True, eos_token_id=95)
    model.forward = fwd
    assert float(g["min_gap"]) > 5e-5                      # the greedy choice never sits on a near-tie in the capture
    assert torch.equal(toks.cpu(), g["tokens"]), (toks.cpu().tolist(), g["tokens"].tolist())
    logits = torch.stack(steps, dim=1)
    assert logits.shape == g["step_logits"].shape
    rel_error_report(f"{name}: last-position logits of all 16 generate() steps vs reference capture", logits, g["step_logits"])


@pytest.mark.parametrize("name", ["generate_ssm_dense_long", "generate_ssm_moe_long"])
def test_generate_graph_tail_matches_reference_long_capture(dev, name, monkeypatch):
    """the round-5 decode path pinned against the REFERENCE, not against its own eager form.  56 new tokens at
    B = 2, sequence 0 reaching eos mid-way (reference generate(), core.py:1520-1644, captured by tools/gen_golden.py
    generate_long): with more than DECODE_GRAPH_MIN_STEPS tokens left the product decodes through the captured HIP graph, the
    stacked caches and the cache-only pre-pass of every layer (csrc/decode_step.hip; core.py:369-373 is why that half of a
    step depends on the caches alone), and - MoE - the small-batch entrance kernel.  Tokens must be EQUAL (incl. the padding
    behind the eos) and every step's last-position logits within 1e-4; spies assert that the graph tail and the pre-pass ran."""
    import apertis_llm_amd as A
    from apertis_llm_amd import model as M
    g = load_golden(name)
    assert M.DECODE_GRAPH and M.DECODE_PREPASS, "this test pins the graph / pre-pass decode path: run it with the defaults"
    cfg = A.ApertisConfig.from_dict(json.loads(str(g["config_json"])))
    model = A.ApertisForCausalLM(cfg)
    model.load_state_dict(g["sd"])
    model = model.to(dev).eval()
    NEW = g["step_logits"].shape[1]
    V = cfg.vocab_size
    # every call of forward leaves its last-position logits in a static device buffer at a device-side counter (a captured
    # graph can replay that; a .cpu() inside the capture could not)
    buf = torch.zeros(2, NEW + 8, V, device=dev)
    ctr = torch.zeros(1, dtype=torch.long, device=dev)
    seen = {"calls": 0, "first_capture_call": None, "prepass": 0, "graph_tail": 0}
    fwd = model.forward

    def spy(*a, **k):
        out = fwd(*a, **k)
        if torch.cuda.is_current_stream_capturing() and seen["first_capture_call"] is None:
            seen["first_capture_call"] = seen["calls"]
        seen["calls"] += 1
        buf.index_copy_(1, ctr, out[1][:, -1:, :].float())
        ctr.add_(1)
        return out
    model.forward = spy
    pre = M.ApertisModel._decode_prepass
    tail = M.ApertisForCausalLM._generate_graph_tail

    def pre_spy(self, st):
        seen["prepass"] += 1
        return pre(self, st)

    def tail_spy(self, *a, **k):
        seen["graph_tail"] += 1
        return tail(self, *a, **k)
    monkeypatch.setattr(M.ApertisModel, "_decode_prepass", pre_spy)
    monkeypatch.setattr(M.ApertisForCausalLM, "_generate_graph_tail", tail_spy)
    toks = model.generate(input_ids=g["prompt"].to(dev), max_new_tokens=NEW, do_sample=False, use_cache=True,
                          eos_token_id=int(g["eos"]), pad_token_id=0)
    torch.cuda.synchronize()
    model.forward = fwd
    assert seen["graph_tail"] == 1, "the captured-graph tail did not engage"
    assert seen["prepass"] >= 1, "the stacked-cache pre-pass (_decode_prepass) did not run"
    assert float(g["min_gap"]) > 1e-3                      # no near-tie among the live greedy choices of the capture
    assert torch.equal(toks.cpu(), g["tokens"]), (toks.cpu().tolist(), g["tokens"].tolist())
    # device entries: the eager calls, the graph's two warm-up steps (same inputs as its first replayed step), the replays
    E = seen["first_capture_call"] - 2
    assert E >= 1 and seen["calls"] == E + 3
    n_dev = int(ctr.item())
    assert n_dev == E + 2 + (NEW - E), (n_dev, E)
    logits = torch.cat([buf[:, :E], buf[:, E + 2:n_dev]], dim=1).cpu()
    assert logits.shape == g["step_logits"].shape
    fin, b = int(g["eos_step"]), int(g["eos_seq"])
    assert int(toks[b, g["prompt"].shape[1] + fin]) == int(g["eos"]) and bool((toks[b, g["prompt"].shape[1] + fin + 1:] == 0).all())
    rel_error_report(f"{name}: last-position logits of all {NEW} generate() steps (graph tail + pre-pass) vs reference capture",
                     logits, g["step_logits"])


def test_scan_lookback_timeout_word_rejects_the_step(dev):
    """The single-pass scan's bounded look-back wait leaves a non-zero error word in its workspace when it times out
    (scan_gate.hip); the activations of such a launch are wrong.  The product path must not train on them silently:
    with the word planted, TrainStep returns a NaN loss, apertis_clip_coef reports a NaN norm and makes the AdamW pass a no-op
    (the step is visibly rejected and the parameters survive) - all without a host sync - and the checker ApertisTrainer calls
    behind its loss.item() raises."""
    import apertis_llm_amd as A
    from apertis_llm_amd import ops
    from apertis_llm_amd._lib import ApertisHipError
    from apertis_llm_amd.training import TrainStep
    torch.manual_seed(0)
    cfg = A.ApertisConfig(vocab_size=256, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                          attention_type="selective_ssm")
    model = A.ApertisForCausalLM(cfg).to(dev).train()
    step = TrainStep(model, total_steps=8)
    ids = torch.randint(4, 256, (2, 128), device=dev)
    loss = step(input_ids=ids, labels=ids)
    assert torch.isfinite(loss) and torch.isfinite(step.optimizer.last_grad_norm)
    assert ops.scan_gate_error(dev) == 0
    ops.scan_gate_raise_on_error(dev)                      # clean: no exception
    word = ops.scan_gate_error_word(dev)
    assert word is not None and word.dtype == torch.int32 and word.numel() == 1
    before = [p.detach().clone() for p in model.parameters()]
    try:
        word.fill_(1)                                      # what a timed-out wait leaves
        loss = step(input_ids=ids, labels=ids)
        assert torch.isnan(loss), "the returned loss must show the failure"
        assert torch.isnan(step.optimizer.last_grad_norm), "norm / clip coefficient must be poisoned"
        # ... and the step is SKIPPED, not turned into NaN parameters: a transient time-out must not destroy the model
        assert all(torch.equal(p.detach(), b) for p, b in zip(model.parameters(), before)), "a poisoned step must not touch the parameters"
        with pytest.raises(ApertisHipError, match="look-back"):
            ops.scan_gate_raise_on_error(dev)
        ops.scan_gate_clear_error(dev)                     # the word is sticky until the caller clears it ...
        assert ops.scan_gate_error(dev) == 0
        loss = step(input_ids=ids, labels=ids)             # ... and the next step trains again
        assert torch.isfinite(loss) and torch.isfinite(step.optimizer.last_grad_norm)
        assert any(not torch.equal(p.detach(), b) for p, b in zip(model.parameters(), before))
    finally:
        word.zero_()
    assert ops.scan_gate_error(dev) == 0
    assert all(torch.isfinite(b).all() for b in before)


def test_prepared_weight_cache_is_scoped_and_never_stale(dev):
    """prepared inference copies of the weights (stacked in_proj, padded x_param_proj, bf16 casts) are keyed on the
    parameter's version counter, which an in-place write through `.data` does not bump (weight init, DDP broadcast,
    `p.data.copy_`).  They are therefore reused only inside ops.prep_cache_scope() - generate(), the trainer's validation
    loop: two plain no_grad forwards around such a write must see it (fp32 and bf16 autocast), a scope must reuse its
    entries and drop them when it closes."""
    import apertis_llm_amd as A
    from apertis_llm_amd import ops
    torch.manual_seed(3)
    cfg = A.ApertisConfig(vocab_size=128, hidden_size=64, num_hidden_layers=2, num_attention_heads=2, intermediate_size=128,
                          attention_type="selective_ssm", use_expert_system=True, num_experts=4, experts_per_token=2)
    model = A.ApertisForCausalLM(cfg).to(dev).eval()
    ids = torch.randint(4, 128, (2, 48), device=dev)
    ssm = model.model.layers[0].attention.attention_mechanism_impl
    for autocast in (False, True):
        def fwd():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                return model(input_ids=ids, use_cache=False)[1].float().clone()
        a = fwd()
        assert len(ops._prep_cache) == 0                       # nothing is kept outside a scope
        for p in (ssm.in_proj_x.weight, ssm.x_param_proj.weight, ssm.out_proj.weight, model.model.layers[1].feed_forward.ffn.expert_w1):
            v0 = p._version
            p.data.mul_(1.5)                                   # behind autograd's back: no version bump
            assert p._version == v0
            b = fwd()
            assert not torch.equal(a, b), "a forward after an in-place .data write served stale prepared weights"
            p.data.div_(1.5)
            a = fwd()
        with ops.prep_cache_scope():
            c = fwd()
            n = len(ops._prep_cache)
            d = fwd()
            assert n > 0 and len(ops._prep_cache) == n and torch.equal(c, d)
        assert len(ops._prep_cache) == 0
    # generate() opens its own scope and leaves nothing behind
    out = model.generate(input_ids=ids[:, :8], max_new_tokens=4)
    assert out.shape == (2, 12) and len(ops._prep_cache) == 0


@pytest.mark.parametrize("experts", [0, 4])
def test_train_prep_one_launch_equals_per_call_preparation(dev, experts):
    """training.TrainStep prepares every GEMM weight's compute copies (stacked in_proj, padded x_param_proj, bf16 and transposed
    bf16 of the projections / experts / dense FFN) in ONE apertis_weight_prep launch at the start of a step.  Three steps with
    it against three steps with the per-call preparation (cat / scatter / apertis_cast_transpose per layer): losses and every
    parameter bit-identical; no per-call cast inside a prepared step; and a write to a weight BETWEEN steps - through `.data`,
    which no version counter sees - is picked up (the refresh runs at the start of every step)."""
    import apertis_llm_amd as A
    from apertis_llm_amd import ops, _lib
    from apertis_llm_amd.training import TrainStep
    cfg = dict(vocab_size=512, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
               attention_type="selective_ssm", use_expert_system=experts > 0, num_experts=experts, experts_per_token=2)
    torch.manual_seed(3)
    init = A.ApertisForCausalLM(A.ApertisConfig(**cfg)).state_dict()
    lib = _lib.load()
    counts = {"cast": 0, "prep": 0}
    real_cast, real_prep = lib.apertis_cast_transpose, lib.apertis_weight_prep

    def cast(*a):
        counts["cast"] += 1
        return real_cast(*a)

    def prep(*a):
        counts["prep"] += 1
        return real_prep(*a)

    def run(on, monkey_between):
        ops.TRAIN_PREP = on
        torch.manual_seed(5)          # (the dropout / router-noise seeds are drawn from the global CPU generator)
        m = A.ApertisForCausalLM(A.ApertisConfig(**cfg))
        m.load_state_dict(init)
        m = m.to(dev).train()
        step = TrainStep(m, lr=1e-3, total_steps=10, bf16=True)
        assert (step.prep is not None) == on
        g = torch.Generator().manual_seed(11)
        losses = []
        counts["cast"] = counts["prep"] = 0
        lib.apertis_cast_transpose, lib.apertis_weight_prep = cast, prep
        try:
            for i in range(3):
                ids = torch.randint(4, 512, (4, 1024), generator=g).to(dev)
                losses.append(step(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids))
                if monkey_between and i == 0:      # a write behind torch's back between two steps
                    blk = m.model.layers[0].attention.attention_mechanism_impl
                    blk.in_proj_z.weight.data.mul_(0.5)
                    blk.x_param_proj.weight.data.add_(0.01)
                    blk.out_proj.weight.data.mul_(1.5)
        finally:
            lib.apertis_cast_transpose, lib.apertis_weight_prep = real_cast, real_prep
        torch.cuda.synchronize()
        return [float(x) for x in losses], [p.detach().clone() for p in m.parameters()], dict(counts)

    try:
        for monkey in (False, True):
            l0, p0, c0 = run(False, monkey)
            l1, p1, c1 = run(True, monkey)
            assert c0["prep"] == 0 and c0["cast"] > 0
            assert c1["prep"] == 3, c1
            # what is left to the per-call path inside a prepared step: the LM head's pieces (not registered), nothing per layer
            assert c1["cast"] <= c0["cast"] // 4, (c0, c1)
            assert l0 == l1, (monkey, l0, l1)
            assert all(torch.equal(a, b) for a, b in zip(p0, p1)), f"parameters differ after three steps (write between steps: {monkey})"
            assert all(x == x for x in l0)
        assert ops.scan_gate_error(dev) == 0
    finally:
        ops.TRAIN_PREP = True


def test_train_prep_leaves_shapes_the_mfma_linear_refuses_to_the_per_call_path(dev):
    """ops.TrainPrep registers a stacked / row-mapped weight only when model._mfma_linear will take it to the MFMA tile
    (K % 8 == 0 and rows % 8 == 0): any other shape goes to stock F.linear, which READS its weight - and a registered weight is
    a placeholder of uninitialised memory (round 4 registered on shape[-1] % 4 == 0 alone: hidden_size 36, 100, 132 ... would
    have multiplied by garbage in every SSM block).  And _mfma_linear refuses a placeholder on the stock path outright."""
    from apertis_llm_amd import ops
    from apertis_llm_amd import model as M
    prep = ops.TrainPrep(dev)
    wx, wz = (torch.nn.Parameter(torch.randn(80, 100, device=dev)) for _ in range(2))          # K = 100: K % 8 == 4
    assert not prep.add_stack(("in_proj_xz", 1), (wx, wz))
    wx8, wz8 = (torch.nn.Parameter(torch.randn(80, 104, device=dev)) for _ in range(2))
    assert prep.add_stack(("in_proj_xz", 2), (wx8, wz8))
    wo = torch.nn.Parameter(torch.randn(84, 104, device=dev))                                  # 84 rows: rows % 8 == 4
    assert not prep.add_stack(("odd_rows", 3), (wo,))
    wp = torch.nn.Parameter(torch.randn(30, 100, device=dev))
    idx = torch.arange(30, device=dev)
    assert not prep.add_rowmap(("x_param_padded", 4), wp, idx, 64)                             # K % 8 == 4
    wp8 = torch.nn.Parameter(torch.randn(30, 104, device=dev))
    assert not prep.add_rowmap(("x_param_padded", 5), wp8, idx, 60)                            # rows_out % 8 == 4
    assert prep.add_rowmap(("x_param_padded", 6), wp8, idx, 64)
    # a placeholder never reaches F.linear
    ph = torch.empty(12, 100, device=dev)
    ph._apertis_prep = object()
    with pytest.raises(ops.ApertisHipError):
        M._mfma_linear(torch.randn(4, 100, device=dev), ph)


@pytest.mark.parametrize("experts,bf16,layers", [(0, False, 3), (4, True, 3), (0, True, 44)])
def test_generate_graph_replay_equals_eager_decoding(dev, monkeypatch, experts, bf16, layers):
    """generate(): from 24 remaining greedy tokens on, the single-token steps of an SSM model run as ONE captured HIP graph
    replayed per token (static token / cache / alive buffers, the host looks at the alive flags every 16 steps).  The tokens
    must be the eager loop's: without eos, with an eos that finishes the sequences at different steps (pad after it, the
    output cut where the last one finished) and with min_new_tokens holding the loop open."""
    import contextlib
    import apertis_llm_amd as A
    from apertis_llm_amd import model as M
    torch.manual_seed(11)
    # (layers = 44: the depth at which round 3's capture faulted - cause never found, did not reproduce in round 4; a tripwire)
    cfg = A.ApertisConfig(vocab_size=97, hidden_size=128, num_hidden_layers=layers, num_attention_heads=2, intermediate_size=256,
                          attention_type="selective_ssm", use_expert_system=experts > 0, num_experts=experts, experts_per_token=2,
                          pad_token_id=0)
    model = A.ApertisForCausalLM(cfg).to(dev).eval()
    prompt = torch.randint(4, 97, (3, 20), device=dev)
    ac = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if bf16 else contextlib.nullcontext

    def gen(graph, **kw):
        monkeypatch.setattr(M, "DECODE_GRAPH", graph)
        with ac():
            return model.generate(input_ids=prompt, max_new_tokens=70, do_sample=False, use_cache=True, **kw)

    base = gen(False, eos_token_id=-1)
    assert base.shape == (3, 90)
    assert torch.equal(gen(True, eos_token_id=-1), base)
    # an eos that every sequence emits somewhere in its continuation, at different steps where the model allows
    cont = base[:, 20:]
    common = set(cont[0].tolist()) & set(cont[1].tolist()) & set(cont[2].tolist())
    eos = min(common, key=lambda t: max(cont[b].tolist().index(t) for b in range(3))) if common else int(cont[0, 40])
    e_eager = gen(False, eos_token_id=eos)
    e_graph = gen(True, eos_token_id=eos)
    assert torch.equal(e_graph, e_eager), (e_graph.shape, e_eager.shape)
    m_eager = gen(False, eos_token_id=eos, min_new_tokens=60)
    m_graph = gen(True, eos_token_id=eos, min_new_tokens=60)
    assert torch.equal(m_graph, m_eager) and m_eager.shape[1] >= 80
    # the graph path really ran: the Python forward is entered a handful of times (prefill, warm-up, capture), not once per token
    calls = {"n": 0}
    fwd = model.forward

    def spy(*a, **k):
        calls["n"] += 1
        return fwd(*a, **k)

    model.forward = spy
    try:
        gen(True, eos_token_id=-1)
    finally:
        model.forward = fwd
    assert calls["n"] <= 6, calls


def test_train_step_overfits_one_batch(dev):
    """End to end through everything the bench step uses (bf16 autocast, one-launch weight preparation, lean scan, expert
    kernels with dropout and capacity, fused LM head + loss, clip + AdamW kernels, OneCycleLR): eighty steps on ONE batch
    drive the loss of a small 8-expert SSM model from ln(V) to well under half of it, monotonically on the whole, with a
    finite gradient norm and a clean look-back error word throughout."""
    import math
    import apertis_llm_amd as A
    from apertis_llm_amd import ops
    from apertis_llm_amd.training import TrainStep
    torch.manual_seed(21)
    cfg = A.ApertisConfig(vocab_size=512, hidden_size=704, num_hidden_layers=2, num_attention_heads=11, intermediate_size=1408,
                          attention_type="selective_ssm", use_expert_system=True, num_experts=8, experts_per_token=2)
    model = A.ApertisForCausalLM(cfg).to(dev).train()
    step = TrainStep(model, lr=3e-3, total_steps=100)
    assert step.prep is not None
    ids = torch.randint(4, 512, (4, 512), device=dev)
    losses = []
    for i in range(80):
        losses.append(step(input_ids=ids, labels=ids))
        if i % 20 == 19:
            assert torch.isfinite(step.optimizer.last_grad_norm)
    losses = [float(x) for x in losses]
    assert all(math.isfinite(x) for x in losses)
    assert 5.5 < losses[0] < 7.5, losses[0]                       # ~ ln(512) = 6.24 (+ the auxiliary losses)
    assert losses[-1] < 0.5 * losses[0], (losses[0], losses[-1])
    assert sum(losses[60:]) / 20 < sum(losses[20:40]) / 20 < sum(losses[:20]) / 20
    assert ops.scan_gate_error(dev) == 0


@pytest.mark.parametrize("experts,bf16,batch,hidden,heads", [(8, True, 1, 128, 4), (8, True, 16, 128, 4), (0, False, 3, 128, 4), (4, True, 5, 128, 4),
                                                             (8, True, 1, 704, 11), (8, True, 16, 704, 11), (0, True, 3, 704, 11)])
def test_decode_prepass_runs_the_cache_only_half_of_every_layer_at_once(dev, experts, bf16, batch, hidden, heads):
    """The reference keeps the FIRST conv output of [cached window | new xp] (core.py:369-373), so conv output, x_param_proj, dt
    projection and state update of a single-token step are functions of the caches alone: with the caches stacked
    (model._StackedPast) the model runs them for all layers in three launches at the start of the step (_decode_prepass,
    csrc/decode_step.hip).  Against the ordinary per-layer steps over ten tokens: logits, conv windows and states bit-identical."""
    import contextlib
    import apertis_llm_amd as A
    from apertis_llm_amd import model as M, ops
    torch.manual_seed(17)
    cfg = A.ApertisConfig(vocab_size=131, hidden_size=hidden, num_hidden_layers=4, num_attention_heads=heads, intermediate_size=2 * hidden,
                          attention_type="selective_ssm", use_expert_system=experts > 0, num_experts=max(experts, 1),
                          experts_per_token=2 if experts else 1, pad_token_id=0)
    model = A.ApertisForCausalLM(cfg).to(dev).eval()
    prompt = torch.randint(4, 131, (batch, 19), device=dev)
    ac = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if bf16 else contextlib.nullcontext

    def run(stacked):
        outs = []
        with torch.no_grad(), ac(), ops.prep_cache_scope():
            o = model(input_ids=prompt, use_cache=True)
            past = [(c.clone(memory_format=torch.contiguous_format), st.clone()) for c, st in o[4]]
            if stacked:
                past = M._StackedPast(torch.stack([c for c, _ in past]).contiguous(),
                                      torch.stack([st.reshape(batch, -1) for _, st in past]).contiguous(), past[0][1].shape[1], past[0][1].shape[2])
            tok = o[1][:, -1].argmax(-1, keepdim=True)
            for _ in range(10):
                o = model(input_ids=tok, past_key_values=past, use_cache=True)
                if not stacked:
                    past = o[4]
                tok = o[1][:, -1].argmax(-1, keepdim=True)
                outs.append(o[1].float().clone())
        torch.cuda.synchronize()
        return outs, [(c.clone(), s.clone()) for c, s in past]

    base, cache0 = run(False)
    calls = {"n": 0}
    real = model.model._decode_prepass

    def spy(st):
        calls["n"] += 1
        r = real(st)
        assert r is not None
        return r

    model.model._decode_prepass = spy
    try:
        fused, cache1 = run(True)
    finally:
        del model.model._decode_prepass
    assert calls["n"] == 10
    # (at 704 wide and bf16 the boundary in front of the SSM block runs as the prologue of the in_proj product: layers 1 .. 3)
    for a, b in zip(base, fused):
        assert torch.equal(a, b), float((a - b).abs().max())
    for (c0, s0), (c1, s1) in zip(cache0, cache1):
        assert torch.equal(c0.reshape(c1.shape), c1) and torch.equal(s0.reshape(s1.shape), s1)
