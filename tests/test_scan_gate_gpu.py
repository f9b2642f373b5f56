"""The scan with the skip + gate fused in (ops.scan_gate -> apertis_scan_gate_fwd/bwd) and the single-token decode
kernels, through the C ABI.

Checked against (i) the CPU oracle (sequential recurrence + gate, reference core.py:337-353,395-396) with autograd,
(ii) the stand-alone pair selective_scan + ssm_gate - bit for bit in fp32: the fused kernel does the same arithmetic,
it only keeps y on chip -, (iii) its own two forms: the single-launch form (ticket counter + published chunk
aggregates) must give the same BITS as the two-launch form, and (iv) the golden SSM layer captured from the reference
(tests/golden/ssm_layer.npz) through the module.  fp32: rtol 1e-4 (BASELINE north_star).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_error_report

pytestmark = pytest.mark.gpu


def _close(got, ref, name, rtol=1e-4, atol_scale=2e-6):
    ref = torch.as_tensor(ref).detach().cpu().to(torch.float64)
    got = got.detach().cpu().to(torch.float64)
    atol = atol_scale * float(ref.abs().max()) + 1e-30
    bad = (got - ref).abs() > atol + rtol * ref.abs()
    assert not bad.any(), f"{name}: {int(bad.sum())} / {bad.numel()} outside rtol {rtol}; max abs diff " \
                          f"{float((got - ref).abs().max()):.3e} (ref max {float(ref.abs().max()):.3e})"


def _inputs(B, L, h, N, seed, dtype=torch.float32):
    """p in the padded layout of the model ([Bt | 0 | C | 0 | dt | 0], blocks of 64 columns), xz = [xp | z]."""
    g = torch.Generator().manual_seed(seed)
    Dn = h * N
    R = max(1, math.ceil(h * N / 4) // 4 * 4) if Dn >= 16 else 4
    Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
    p = torch.randn(B, L, 2 * Wb + Wr, generator=g)
    p[..., Dn:Wb] = 0
    p[..., Wb + Dn:2 * Wb] = 0
    xz = torch.randn(B, L, 2 * Dn, generator=g)
    xc = torch.randn(B, L, Dn, generator=g)
    logits = torch.randn(B, L, h, generator=g) - 4.0
    A_log = torch.empty(h, N).uniform_(math.log(0.5), math.log(0.99), generator=g)
    D = 1.0 + 0.2 * torch.randn(Dn, generator=g)
    dout = torch.randn(B, L, Dn, generator=g)
    h0 = torch.randn(B, Dn, generator=g)
    return dict(p=p.to(dtype), xz=xz.to(dtype), xc=xc.to(dtype), logits=logits, A_log=A_log, D=D, dout=dout.to(dtype), h0=h0,
                Dn=Dn, Wb=Wb, R=R)


def _oracle(i, use_h0):
    """Reference math with autograd on CPU, fp32 on the (possibly bf16-rounded) inputs."""
    from oracle import ref_cpu
    Dn, Wb = i["Dn"], i["Wb"]
    leaves = {k: i[k].float().clone().requires_grad_(True) for k in ("p", "xz", "xc", "logits", "A_log", "D")}
    Bt, C, z = leaves["p"][..., :Dn], leaves["p"][..., Wb:Wb + Dn], leaves["xz"][..., Dn:]
    y, hl = ref_cpu.scan_recurrent(F.softplus(leaves["logits"]), leaves["A_log"], Bt, C, i["h0"] if use_h0 else None)
    out = (y + leaves["D"].view(1, 1, -1) * leaves["xc"]) * F.silu(z)             # core.py:395-396
    out.backward(i["dout"].float())
    return out.detach(), hl.detach(), {k: v.grad for k, v in leaves.items()}


KINDS = {"staged": (False, False), "lean": (True, False), "lookback": (False, "all")}      # (SCAN_LEAN, SCAN_LOOKBACK)


class _kind:
    """`with _kind("staged" | "lean" | "lookback"):` pins which form of the fused scan ops.scan_gate takes where the shape
    allows it (bf16, N = 16: lean for 128 < Dn <= 256, look-back for Dn <= 256; everything else is staged anyway)."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        from apertis_llm_amd import ops
        self.old = (ops.SCAN_LEAN, ops.SCAN_LEAN_BWD, ops.SCAN_LOOKBACK)
        lean, lb = KINDS[self.kind]
        ops.SCAN_LEAN = ops.SCAN_LEAN_BWD = lean
        ops.SCAN_LOOKBACK = lb

    def __exit__(self, *a):
        from apertis_llm_amd import ops
        ops.SCAN_LEAN, ops.SCAN_LEAN_BWD, ops.SCAN_LOOKBACK = self.old


def _fused(dev, i, use_h0, single_pass, kind="staged"):
    from apertis_llm_amd import ops
    Dn, Wb = i["Dn"], i["Wb"]
    old = ops.SCAN_SINGLE_PASS
    ops.SCAN_SINGLE_PASS = single_pass
    pin = _kind(kind)
    pin.__enter__()
    try:
        lv = {k: i[k].to(dev).requires_grad_(True) for k in ("p", "xz", "xc", "logits", "A_log", "D")}
        Btp, Cp, _dt = ops.split_cols(lv["p"], (Wb, Wb, lv["p"].shape[-1] - 2 * Wb))
        _xp, z = ops.split_cols(lv["xz"], (Dn, Dn))
        out, hl = ops.scan_gate(lv["logits"], lv["A_log"], Btp, Cp, lv["xc"], z, lv["D"], h0=i["h0"].to(dev) if use_h0 else None,
                                delta_softplus=True, return_last=True)
        out.backward(i["dout"].to(dev))
        assert ops.scan_gate_error() == 0
    finally:
        ops.SCAN_SINGLE_PASS = old
        pin.__exit__()
    return out.detach(), hl.detach(), {k: v.grad for k, v in lv.items()}


def _two_ops(dev, i, use_h0):
    """selective_scan (fp32 y) followed by ssm_gate: the path the fused kernel replaces."""
    from apertis_llm_amd import ops
    Dn, Wb = i["Dn"], i["Wb"]
    lv = {k: i[k].to(dev).requires_grad_(True) for k in ("p", "xz", "xc", "logits", "A_log", "D")}
    Bt, C, z = lv["p"][..., :Dn], lv["p"][..., Wb:Wb + Dn], lv["xz"][..., Dn:]
    y, hl = ops.selective_scan(lv["logits"], lv["A_log"], Bt, C, h0=i["h0"].to(dev) if use_h0 else None, delta_softplus=True,
                               y_dtype=torch.float32, return_last=True)
    out = ops.ssm_gate(y, lv["xc"], z, lv["D"])
    out.backward(i["dout"].to(dev))
    return out.detach(), hl.detach(), {k: v.grad for k, v in lv.items()}


SHAPES = [(2, 130, 11, 16), (1, 4096, 4, 16), (3, 64, 14, 16), (2, 100, 5, 8), (1, 77, 2, 64), (1, 1, 3, 16), (2, 257, 1, 16),
          (1, 2245, 11, 16)]


@pytest.mark.parametrize("B,L,h,N", SHAPES)
@pytest.mark.parametrize("use_h0", [False, True])
def test_scan_gate_fp32_vs_oracle_and_both_forms(dev, B, L, h, N, use_h0):
    i = _inputs(B, L, h, N, seed=L * 31 + h)
    Dn, Wb = i["Dn"], i["Wb"]
    o_out, o_hl, og = _oracle(i, use_h0)
    outs = {}
    for sp in (False, True):
        out, hl, g = _fused(dev, i, use_h0, sp)
        outs[sp] = (out, hl, g)
        tag = f"[single_pass={sp}] "
        _close(out, o_out, tag + "out")
        _close(hl, o_hl, tag + "h_last")
        _close(g["logits"], og["logits"], tag + "d_logits", atol_scale=2e-5)
        _close(g["A_log"], og["A_log"], tag + "dA_log", atol_scale=2e-5)
        _close(g["D"], og["D"], tag + "dD", atol_scale=2e-5)
        _close(g["xc"], og["xc"], tag + "dxc")
        _close(g["p"][..., :Dn], og["p"][..., :Dn], tag + "dBt")
        _close(g["p"][..., Wb:Wb + Dn], og["p"][..., Wb:Wb + Dn], tag + "dC")
        _close(g["xz"][..., Dn:], og["xz"][..., Dn:], tag + "dz")
        # the zero pad of the Bt / C slices stays exactly zero in the gradient buffer; dt columns get no gradient here
        assert float(g["p"][..., Dn:Wb].abs().max() if Wb > Dn else 0.0) == 0.0
        assert float(g["p"][..., Wb + Dn:2 * Wb].abs().max() if Wb > Dn else 0.0) == 0.0
        assert float(g["p"][..., 2 * Wb:].abs().max()) == 0.0 and float(g["xz"][..., :Dn].abs().max()) == 0.0
    # one launch == two launches, bit for bit (fixed composition order)
    a, b = outs[False], outs[True]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.parametrize("B,L,h,N", [(2, 130, 11, 16), (1, 700, 14, 16), (2, 64, 4, 16)])
def test_scan_gate_equals_scan_then_gate(dev, B, L, h, N):
    """fp32: the fused kernel performs the stand-alone pair's arithmetic with y kept in registers instead of a round
    trip through HBM.  The two differ only in how the tokens of a chunk are cut into per-wave segments (the fused fp32
    tiles are 64 tokens, the stand-alone forward's 128), i.e. in the last bits of the states: held to 1e-5, ten times
    inside the parity bar."""
    i = _inputs(B, L, h, N, seed=7 * L + N)
    Dn, Wb = i["Dn"], i["Wb"]
    f_out, f_hl, fg = _fused(dev, i, True, True)
    t_out, t_hl, tg = _two_ops(dev, i, True)
    _close(f_out, t_out, "out", rtol=1e-5)
    _close(f_hl, t_hl, "h_last", rtol=1e-5)
    for k in ("logits", "xc", "D", "A_log"):
        _close(fg[k], tg[k], "grad " + k, rtol=1e-5, atol_scale=1e-5)
    _close(fg["p"], tg["p"], "dp (dBt | dC)", rtol=1e-5)
    _close(fg["xz"], tg["xz"], "dxz (dz)", rtol=1e-5)


@pytest.mark.parametrize("kind", ["staged", "lean", "lookback"])
def test_scan_gate_bf16_config4_shape(dev, kind):
    """bf16 activations at the bench's per-layer shape (B=4 of the 44, L=4096, 11 heads, Dn=176), every form of the kernel
    (staged single-pass / its two-launch form, lean three-launch, look-back one-launch): against the oracle on the SAME
    bf16-rounded inputs (fp32 state; only the output rounding to bf16 differs), and the staged form's one launch == its two."""
    i = _inputs(4, 4096, 11, 16, seed=5, dtype=torch.bfloat16)
    Dn, Wb = i["Dn"], i["Wb"]
    o_out, o_hl, og = _oracle(i, False)
    res = {}
    for sp in (False, True):
        out, hl, g = _fused(dev, i, False, sp, kind)
        res[sp] = (out, hl, g)
        assert out.dtype == torch.bfloat16 and g["p"].dtype == torch.bfloat16
        _close(out.float(), o_out, "out", rtol=8e-3, atol_scale=1e-3)             # one bf16 rounding: 2^-8 = 3.9e-3
        _close(hl, o_hl, "h_last")
        _close(g["logits"], og["logits"], "d_logits", rtol=1e-3, atol_scale=1e-4)
        _close(g["A_log"], og["A_log"], "dA_log", rtol=1e-3, atol_scale=1e-4)
        _close(g["D"], og["D"], "dD", rtol=1e-3, atol_scale=1e-4)
        _close(g["xc"].float(), og["xc"], "dxc", rtol=8e-3, atol_scale=1e-3)
        _close(g["p"][..., :Dn].float(), og["p"][..., :Dn], "dBt", rtol=8e-3, atol_scale=1e-3)
        _close(g["p"][..., Wb:Wb + Dn].float(), og["p"][..., Wb:Wb + Dn], "dC", rtol=8e-3, atol_scale=1e-3)
        _close(g["xz"][..., Dn:].float(), og["xz"][..., Dn:], "dz", rtol=8e-3, atol_scale=1e-3)
    assert torch.equal(res[False][0], res[True][0])
    for k in res[False][2]:
        assert torch.equal(res[False][2][k], res[True][2][k]), k


@pytest.mark.parametrize("kind,batch,heads", [("staged", 32, 11), ("staged", 44, 11), ("lean", 44, 11), ("lookback", 44, 11),
                                              ("lookback", 32, 11), ("staged", 16, 4), ("lookback", 16, 4)])
def test_scan_gate_single_pass_stress_full_size(dev, kind, batch, heads):
    """Every form of the fused scan at the bench's full per-GPU sizes - config 4 (B=44, the batch the bench line is quoted
    on, and the earlier B=32; L=4096, Dn=176) and config 3 (B=16, L=4096, Dn=64: four sequences side by side in a wave of the
    look-back form) - thousands of forward and backward work items, far more than are resident at once, run back to back on
    one workspace: the ticket order of the staged and look-back forms must keep every wait on an already started
    work-group (no time-out in the error word), the alternating ticket counters - which the two forms SHARE - must hand over
    cleanly from launch to launch, and every repetition must give the same bits."""
    from apertis_llm_amd import ops
    with _kind(kind):
        _stress(dev, ops, batch, heads)


def _stress(dev, ops, batch, heads):
    i = _inputs(batch, 4096, heads, 16, seed=9, dtype=torch.bfloat16)
    Dn, Wb = i["Dn"], i["Wb"]
    lv = {k: i[k].to(dev).requires_grad_(True) for k in ("p", "xz", "xc", "logits", "A_log", "D")}
    dout = i["dout"].to(dev)
    first = None
    for rep in range(6):
        for v in lv.values():
            v.grad = None
        Btp, Cp, _dt = ops.split_cols(lv["p"], (Wb, Wb, lv["p"].shape[-1] - 2 * Wb))
        _xp, z = ops.split_cols(lv["xz"], (Dn, Dn))
        out = ops.scan_gate(lv["logits"], lv["A_log"], Btp, Cp, lv["xc"], z, lv["D"], delta_softplus=True)
        out.backward(dout)
        got = (out.detach().clone(), lv["p"].grad.clone(), lv["xz"].grad.clone(), lv["A_log"].grad.clone(), lv["D"].grad.clone())
        if first is None:
            first = got
        else:
            for a, b in zip(first, got):
                assert torch.equal(a, b)
    assert ops.scan_gate_error() == 0
    assert torch.isfinite(first[0].float()).all() and torch.isfinite(first[1].float()).all()
    # size-independent property: the first half of the sequence does not depend on the second (causality), bit for bit
    half = 2048
    Btp, Cp, _dt = ops.split_cols(lv["p"].detach()[:, :half].contiguous(), (Wb, Wb, lv["p"].shape[-1] - 2 * Wb))
    out_h = ops.scan_gate(lv["logits"].detach()[:, :half].contiguous(), lv["A_log"].detach(), Btp, Cp,
                          lv["xc"].detach()[:, :half].contiguous(), lv["xz"].detach()[:, :half, Dn:], lv["D"].detach(),
                          delta_softplus=True)
    assert torch.equal(out_h, first[0][:, :half])


@pytest.mark.parametrize("kind", ["lean", "lookback"])
def test_scan_forms_agree_on_gradients(dev, kind):
    """The lean (three launches) and look-back (one launch) forms against the staged kernels at (4, 4096, 11, 16) bf16: the
    same recurrence, different chunk-carry compositions - outputs within one bf16 rounding of each other, every gradient
    within 4e-3 relative RMS (bf16 outputs) resp. 1e-4 (the fp32 ones), pad columns exactly zero.  (Round 4 checked this with
    tools/scan_lean_check.py only.)"""
    i = _inputs(4, 4096, 11, 16, seed=13, dtype=torch.bfloat16)
    Dn, Wb = i["Dn"], i["Wb"]
    a_out, a_hl, ag = _fused(dev, i, False, True, "staged")
    b_out, b_hl, bg = _fused(dev, i, False, True, kind)
    d = (a_out.float() - b_out.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(a_out.float().abs().max()), float(d.max())
    _close(b_hl, a_hl, "h_last", rtol=1e-4)
    for k in ag:
        ga, gb = ag[k].float(), bg[k].float()
        rel = float((ga - gb).pow(2).mean().sqrt() / (ga.pow(2).mean().sqrt() + 1e-30))
        assert rel < (4e-3 if ag[k].dtype == torch.bfloat16 else 1e-4), (k, rel)
    assert float(bg["p"][..., Dn:Wb].abs().max()) == 0.0 and float(bg["p"][..., Wb + Dn:2 * Wb].abs().max()) == 0.0


@pytest.mark.parametrize("B,L,h", [(3, 257, 11), (5, 100, 4), (1, 64, 16), (7, 1000, 2), (3, 130, 1), (2, 2245, 11), (9, 333, 8), (2, 640, 6)])
@pytest.mark.parametrize("use_h0", [False, True])
def test_scan_lookback_bf16_vs_oracle_shapes(dev, B, L, h, use_h0):
    """The look-back form (apertis_scan_lookback_fwd / _bwd) against the oracle on bf16-rounded inputs over the geometries it
    takes: ragged last chunks and waves past the end, one to sixteen sequences side by side in a wave (Dn = 16 ... 256), a
    batch that does not fill the last wave, an initial state and the final state (the prefill-with-cache form), Dn = 16 whose
    backward declines (pad columns wider than the row's lanes) and falls through to the staged kernels from ckpt16[:, ::4]."""
    from apertis_llm_amd import ops
    i = _inputs(B, L, h, 16, seed=17 * L + h, dtype=torch.bfloat16)
    Dn, Wb = i["Dn"], i["Wb"]
    o_out, o_hl, og = _oracle(i, use_h0)
    out, hl, g = _fused(dev, i, use_h0, True, "lookback")
    _close(out.float(), o_out, "out", rtol=8e-3, atol_scale=1e-3)
    _close(hl, o_hl, "h_last", rtol=1e-4, atol_scale=1e-5)
    _close(g["logits"], og["logits"], "d_logits", rtol=2e-3, atol_scale=2e-4)
    _close(g["A_log"], og["A_log"], "dA_log", rtol=2e-3, atol_scale=2e-4)
    _close(g["D"], og["D"], "dD", rtol=2e-3, atol_scale=2e-4)
    _close(g["xc"].float(), og["xc"], "dxc", rtol=8e-3, atol_scale=1e-3)
    _close(g["p"][..., :Dn].float(), og["p"][..., :Dn], "dBt", rtol=8e-3, atol_scale=1e-3)
    _close(g["p"][..., Wb:Wb + Dn].float(), og["p"][..., Wb:Wb + Dn], "dC", rtol=8e-3, atol_scale=1e-3)
    _close(g["xz"][..., Dn:].float(), og["xz"][..., Dn:], "dz", rtol=8e-3, atol_scale=1e-3)
    assert float(g["p"][..., Dn:Wb].abs().max() if Wb > Dn else 0.0) == 0.0
    assert float(g["p"][..., Wb + Dn:2 * Wb].abs().max() if Wb > Dn else 0.0) == 0.0


def test_ssm_layer_golden_through_fused_path(dev):
    """The module on the fused kernel against the SSM layer captured from the reference (forward), with the achieved
    relative error reported; then the module's two paths (fused / scan + gate with output_attentions) against each
    other."""
    import apertis_llm_amd as A
    g = load_golden("ssm_layer")
    cfg = A.ApertisConfig(hidden_size=48, num_attention_heads=3, ssm_d_state=16, attention_type="selective_ssm")
    mod = A.SelectiveLinearAttention(cfg)
    mod.load_state_dict(g["sd"])
    mod = mod.to(dev).eval()
    with torch.no_grad():
        out_f, y_f, cache_f = mod(g["x"].to(dev), use_cache=True)                            # fused
        out_t, y_t, cache_t = mod(g["x"].to(dev), output_attentions=True, use_cache=True)    # scan + gate
    assert y_f is None and y_t is not None
    rel_error_report("ssm_layer (fused scan+gate) out vs reference capture", out_f, g["out"])
    _close(cache_f[1].reshape(2, -1), g["ssm_state"], "ssm_state")
    _close(cache_f[0], g["conv_state"], "conv_state")
    _close(out_f, out_t, "fused vs scan + gate", rtol=1e-5)        # token segments of 8 vs 16: last bits of the states
    _close(cache_f[1], cache_t[1], "final state", rtol=1e-5)


def test_decode_kernels_match_the_chunk_path(dev):
    """Single-token steps through apertis_ssm_decode_conv / apertis_ssm_decode_state (the no-grad L = 1 path of the
    module) against the same steps through the general kernels (grad mode on selects them), which
    test_ssm_layer_cached_decode_matches_reference_semantics pins to the oracle: same cache contents, same outputs
    (fp32 rtol 1e-5: the arithmetic is the same, the GEMM tiles differ for one-row inputs)."""
    import apertis_llm_amd as A
    g = load_golden("ssm_layer")
    cfg = A.ApertisConfig(hidden_size=48, num_attention_heads=3, ssm_d_state=16, attention_type="selective_ssm")
    mod = A.SelectiveLinearAttention(cfg)
    mod.load_state_dict(g["sd"])
    mod = mod.to(dev).eval()
    x = g["x"].to(dev)
    with torch.no_grad():
        _, _, cache0 = mod(x[:, :30], use_cache=True)
    cache_a = cache_b = cache0
    for t in range(30, 37):
        with torch.no_grad():
            out_a, _, cache_a = mod(x[:, t:t + 1], past_key_value=cache_a, use_cache=True)      # decode kernels
        with torch.enable_grad():
            out_b, _, cache_b = mod(x[:, t:t + 1], past_key_value=cache_b, use_cache=True)      # chunk kernels
        _close(out_a, out_b, f"step {t} out", rtol=1e-5)
        _close(cache_a[0], cache_b[0], f"step {t} conv_state", rtol=0, atol_scale=0)
        _close(cache_a[1], cache_b[1], f"step {t} ssm_state", rtol=1e-6)


@pytest.mark.parametrize("k", [2, 4, 5, 6, 8, 16])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_decode_conv_window_shift_for_every_accepted_kernel_width(dev, k, dt):
    """apertis_ssm_decode_conv's cache update (reference core.py:369-373,398-400: the new window is the last k - 1 tokens of
    [conv_state | xp]) for every ssm_conv_kernel the entry point accepts, out of place and in place (the decode graph's form).
    Round 4's kernel shifted three entries only: right for the default k = 4, stale / uninitialised from k = 6 on."""
    from apertis_llm_amd import ops
    torch.manual_seed(k)
    B, Dn = 3, 80
    xp = torch.randn(B, Dn, device=dev).to(dt)
    cs = torch.randn(B, Dn, max(k - 1, 0), device=dev).to(dt)
    w = torch.randn(Dn, 1, k, device=dev)
    bias = torch.randn(Dn, device=dev)
    win = torch.cat([cs, xp.unsqueeze(-1)], -1).float()                 # [B, Dn, k]
    acc = w[:, 0, k - 1] * win[..., 0] + bias                            # (the front-slice quirk: only the last tap meets a value)
    ref_xc, ref_cs = torch.nn.functional.silu(acc), win[..., 1:].to(dt)
    xc, cs_out = ops.ssm_decode_step(xp, cs, w, bias)
    assert cs_out.data_ptr() != cs.data_ptr()
    assert torch.equal(cs_out, ref_cs)
    _close(xc, ref_xc, "xc", rtol=1e-5 if dt == torch.float32 else 1e-2)
    cs2 = cs.clone()
    xc2, cs_in = ops.ssm_decode_step(xp, cs2, w, bias, inplace=True)
    assert cs_in.data_ptr() == cs2.data_ptr() and torch.equal(cs2, ref_cs) and torch.equal(xc2, xc)


@pytest.mark.parametrize("B,L,h,N,R", [(3, 257, 11, 16, 44), (2, 700, 14, 16, 56), (2, 130, 12, 16, 48), (1, 200, 11, 16, 20)])
def test_scan_gate_dt_fused_equals_tiny_linear_then_scan_gate(dev, monkeypatch, B, L, h, N, R):
    """N4's pre-scan prologue: ops.scan_gate_dt (dt_proj_head inside the lean forward's state pass, core.py:382-396) against
    ops.tiny_linear followed by ops.scan_gate on the same padded projection output: logits, outputs, final state and every
    gradient (incl. dt_proj_head's weight / bias and the dt columns of p) must be the same BITS; and the fused kernel must be
    the one that ran (the call count of the stand-alone kernel's entry point does not move)."""
    from apertis_llm_amd import ops, _lib
    torch.manual_seed(5)
    Dn = h * N
    Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
    p0 = torch.randn(B, L, 2 * Wb + Wr, device=dev)
    p0[..., Dn:Wb] = 0
    p0[..., Wb + Dn:2 * Wb] = 0
    p0[..., 2 * Wb + R:] = 0
    W = (0.3 * torch.randn(h, R, device=dev)).requires_grad_(True)
    bias = (torch.randn(h, device=dev) - 4.0).requires_grad_(True)
    A_log = torch.empty(h, N, device=dev).uniform_(math.log(0.5), math.log(0.99)).requires_grad_(True)
    D = (1.0 + 0.2 * torch.randn(Dn, device=dev)).requires_grad_(True)
    xz0, xc0 = torch.randn(B, L, 2 * Dn, device=dev), torch.randn(B, L, Dn, device=dev)
    dout = torch.randn(B, L, Dn, device=dev).bfloat16()
    h0 = torch.randn(B, Dn, device=dev)
    res = {}
    lib = _lib.load()
    calls = {"n": 0}
    real = lib.apertis_tiny_linear_fwd

    def counting(*a):
        calls["n"] += 1
        return real(*a)

    monkeypatch.setattr(ops, "SCAN_DT_FUSED", True)      # (off by default: measured slower than the launch it replaces)
    monkeypatch.setattr(ops, "SCAN_LOOKBACK", False)     # (the fused prologue lives in the lean form's state pass)
    for mode in ("two_op", "fused"):
        p, xz, xc = (t.bfloat16().requires_grad_(True) for t in (p0, xz0, xc0))
        for t in (W, bias, A_log, D):
            t.grad = None
        Btp, Cp, dt_in = ops.split_cols(p, (Wb, Wb, R, Wr - R))[:3]
        _, z = ops.split_cols(xz, (Dn, Dn))
        lib.apertis_tiny_linear_fwd = counting
        try:
            n0 = calls["n"]
            if mode == "two_op":
                out, hl = ops.scan_gate(ops.tiny_linear(dt_in, W, bias), A_log, Btp, Cp, xc, z, D, h0=h0, delta_softplus=True,
                                        return_last=True)
            else:
                out, hl = ops.scan_gate_dt(dt_in, W, bias, A_log, Btp, Cp, xc, z, D, h0=h0, delta_softplus=True, return_last=True)
            ran_standalone = calls["n"] - n0
        finally:
            lib.apertis_tiny_linear_fwd = real
        out.backward(dout)
        res[mode] = dict(out=out.detach(), hl=hl.detach(), p=p.grad, xz=xz.grad, xc=xc.grad, W=W.grad.clone(), b=bias.grad.clone(),
                         A=A_log.grad.clone(), D=D.grad.clone(), standalone=ran_standalone)
    assert res["two_op"]["standalone"] == 1
    lean_shape = 128 < Dn <= 256
    assert res["fused"]["standalone"] == (0 if lean_shape and ops.SCAN_LEAN and ops.SCAN_DT_FUSED else 1)
    for k in ("out", "hl", "p", "xz", "xc", "W", "b", "A", "D"):
        assert torch.equal(res["two_op"][k], res["fused"][k]), f"{k}: fused dt_proj differs from the two-op form " \
            f"(max abs diff {float((res['two_op'][k].float() - res['fused'][k].float()).abs().max()):.3e})"
