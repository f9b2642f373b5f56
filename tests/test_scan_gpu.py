"""Parity of the HIP selective scan (through the C ABI) with the CPU oracle and with the golden
vectors captured from the reference (tests/golden/scan_*.npz, made by tools/gen_golden.py).

Tolerances: the reference target is fp32 results within 1e-4 rtol (BASELINE.json north_star);
fp32 checks below use rtol 1e-4 with a small atol scaled to the tensor magnitude.
"""
import math

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _close(got, ref, name, rtol=RTOL, atol_scale=2e-6):
    ref = torch.as_tensor(ref).to(torch.float64)
    got = got.detach().cpu().to(torch.float64)
    atol = atol_scale * float(ref.abs().max()) + 1e-30
    bad = (got - ref).abs() > atol + rtol * ref.abs()
    assert not bad.any(), f"{name}: {int(bad.sum())} / {bad.numel()} outside rtol {rtol}; max abs diff " \
                          f"{float((got - ref).abs().max()):.3e} (ref max {float(ref.abs().max()):.3e})"


def _run(dev, delta, A_log, Bt, C, dy, h0=None, **kw):
    from apertis_llm_amd import ops
    d = delta.float().to(dev).requires_grad_(True)
    a = A_log.float().to(dev).requires_grad_(True)
    b = Bt.to(dev).requires_grad_(True)
    c = C.to(dev).requires_grad_(True)
    res = ops.selective_scan(d, a, b, c, None if h0 is None else h0.float().to(dev), return_last=True, **kw)
    y, hl = res
    y.backward(dy.to(dev).to(y.dtype))
    return y, hl, d.grad, a.grad, b.grad, c.grad


@pytest.mark.parametrize("name", ["L1", "L7", "L64", "L257", "L2048", "bigdelta"])
def test_scan_golden_fp32(dev, name):
    g = load_golden("scan_" + name)
    y, hl, dd, da, db, dc = _run(dev, g["delta"], g["A_log"], g["Bt"], g["C"], g["dy"], g.get("h0"))
    _close(y, g["y"], "y")
    if "h_last" in g:
        _close(hl, g["h_last"], "h_last")
    _close(dd, g["d_delta"], "d_delta", atol_scale=1e-5)
    _close(da, g["dA_log"], "dA_log", atol_scale=1e-5)
    _close(db, g["dBt"], "dBt")
    _close(dc, g["dC"], "dC")
    if name == "bigdelta":
        assert int(g["parallel_nonfinite"]) > 0  # the reference's cumsum form blows up here; we must not
        assert torch.isfinite(y).all()


def test_scan_golden_f64_reference(dev):
    """fp32 kernel against the reference run in float64: bounds the kernel's own rounding."""
    g = load_golden("scan_L257_f64")
    y, hl, dd, da, db, dc = _run(dev, g["delta"], g["A_log"], g["Bt"].float(), g["C"].float(), g["dy"].float())
    _close(y, g["y"], "y", rtol=2e-5)
    _close(db, g["dBt"], "dBt", rtol=2e-5)
    _close(dd, g["d_delta"], "d_delta", rtol=2e-5, atol_scale=1e-5)


@pytest.mark.parametrize("B,L,h,N", [(2, 130, 11, 16), (1, 4096, 4, 16), (3, 64, 14, 16), (2, 100, 5, 8), (1, 77, 2, 64)])
@pytest.mark.parametrize("softplus", [False, True])
def test_scan_vs_oracle_strided_views(dev, B, L, h, N, softplus):
    """Bt/C taken as the strided column slices of one projection output (core.py:377-385),
    ragged L, channel tiles with tails (Dn=176), in-kernel softplus."""
    from oracle import ref_cpu
    torch.manual_seed(L * h)
    R = math.ceil(h * N / 4)
    Dn = h * N
    p = torch.randn(B, L, R + 2 * Dn)
    logits = torch.randn(B, L, h) - 4.0
    A_log = torch.empty(h, N).uniform_(math.log(0.5), math.log(0.99))
    dy = torch.randn(B, L, Dn)
    h0 = torch.randn(B, Dn)
    delta = torch.nn.functional.softplus(logits)
    Bt, C = p[..., R:R + Dn], p[..., R + Dn:]
    yo, hlo = ref_cpu.scan_recurrent(delta, A_log, Bt, C, h0)
    ddo, dao, dbo, dco = ref_cpu.scan_backward(delta, A_log, Bt, C, dy, h0)
    if softplus:
        ddo = ddo * torch.sigmoid(logits)

    from apertis_llm_amd import ops
    pd = p.to(dev).requires_grad_(True)
    dl = (logits if softplus else delta).to(dev).requires_grad_(True)
    al = A_log.to(dev).requires_grad_(True)
    y, hl = ops.selective_scan(dl, al, pd[..., R:R + Dn], pd[..., R + Dn:], h0.to(dev), delta_softplus=softplus,
                               return_last=True)
    y.backward(dy.to(dev))
    _close(y, yo, "y")
    _close(hl, hlo, "h_last")
    _close(dl.grad, ddo, "d_delta", atol_scale=2e-5)
    _close(al.grad, dao, "dA_log", atol_scale=2e-5)
    _close(pd.grad[..., R:R + Dn], dbo, "dBt")
    _close(pd.grad[..., R + Dn:], dco, "dC")
    assert float(pd.grad[..., :R].abs().max()) == 0.0


@pytest.mark.parametrize("y_dtype", [torch.float32, torch.bfloat16])
def test_scan_bf16_io(dev, y_dtype):
    """bf16 Bt/C (what bf16 autocast hands the op): compare with the oracle fed the SAME
    bf16-rounded inputs; the state stays fp32 so only output rounding differs."""
    from oracle import ref_cpu
    torch.manual_seed(3)
    B, L, h, N = 2, 300, 11, 16
    Dn = h * N
    R = 44
    p = torch.randn(B, L, R + 2 * Dn).bfloat16()
    delta = torch.nn.functional.softplus(torch.randn(B, L, h) - 4.0)
    A_log = torch.empty(h, N).uniform_(math.log(0.5), math.log(0.99))
    dy = torch.randn(B, L, Dn).to(y_dtype)
    Bt, C = p[..., R:R + Dn], p[..., R + Dn:]
    yo, _ = ref_cpu.scan_recurrent(delta, A_log, Bt.float(), C.float())
    ddo, dao, dbo, dco = ref_cpu.scan_backward(delta, A_log, Bt.float(), C.float(), dy.float())
    from apertis_llm_amd import ops
    pd = p.to(dev).requires_grad_(True)
    dl = delta.to(dev).requires_grad_(True)
    al = A_log.to(dev).requires_grad_(True)
    y = ops.selective_scan(dl, al, pd[..., R:R + Dn], pd[..., R + Dn:], y_dtype=y_dtype)
    assert y.dtype == y_dtype
    y.backward(dy.to(dev))
    out_tol = 1e-4 if y_dtype == torch.float32 else 8e-3   # bf16 has 8 significant bits
    _close(y, yo, "y", rtol=out_tol, atol_scale=out_tol)
    _close(dl.grad, ddo, "d_delta", atol_scale=2e-5)
    _close(al.grad, dao, "dA_log", atol_scale=2e-5)
    _close(pd.grad[..., R:R + Dn], dbo, "dBt", rtol=8e-3, atol_scale=8e-3)   # stored as bf16
    _close(pd.grad[..., R + Dn:], dco, "dC", rtol=8e-3, atol_scale=8e-3)


def test_scan_full_size_properties(dev):
    """BASELINE config sizes (seq 4096, 11 heads x 16): size-independent properties.
    (1) splitting the sequence and chaining the carried state reproduces the one-shot scan;
    (2) linearity in Bt;  (3) dBt of a scan with C=1, dy=1 equals the reverse cumulative
    product sum, checked through the adjoint identity <dy, y(Bt)> == <dBt, Bt>."""
    from apertis_llm_amd import ops
    torch.manual_seed(0)
    B, L, h, N = 4, 4096, 11, 16
    Dn = h * N
    delta = torch.nn.functional.softplus(torch.randn(B, L, h, device=dev) - 5.5)
    A_log = torch.empty(h, N, device=dev).uniform_(math.log(0.5), math.log(0.99))
    Bt = torch.randn(B, L, Dn, device=dev)
    Bt2 = torch.randn(B, L, Dn, device=dev)
    C = torch.randn(B, L, Dn, device=dev)
    y, hl = ops.selective_scan(delta, A_log, Bt, C, return_last=True)
    ya, ha = ops.selective_scan(delta[:, :1500], A_log, Bt[:, :1500], C[:, :1500], return_last=True)
    yb, hb = ops.selective_scan(delta[:, 1500:].contiguous(), A_log, Bt[:, 1500:], C[:, 1500:], h0=ha, return_last=True)
    _close(torch.cat([ya, yb], 1), y.cpu(), "split/chain", rtol=2e-5, atol_scale=2e-6)
    _close(hb, hl.cpu(), "final state", rtol=2e-5)
    y2 = ops.selective_scan(delta, A_log, Bt2, C)
    y12 = ops.selective_scan(delta, A_log, Bt + Bt2, C)
    _close(y12, (y + y2).cpu(), "linearity", rtol=1e-4, atol_scale=1e-5)
    Btg = Bt.clone().requires_grad_(True)
    dy = torch.randn(B, L, Dn, device=dev)
    yg = ops.selective_scan(delta, A_log, Btg, C)
    yg.backward(dy)
    lhs = float((dy.double() * yg.double()).sum())
    rhs = float((Btg.grad.double() * Bt.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs) + 1e-3, (lhs, rhs)


def test_scan_rejects_bad_input(dev):
    from apertis_llm_amd import ops
    from apertis_llm_amd._lib import ApertisHipError
    d = torch.rand(1, 8, 2, device=dev)
    A = torch.zeros(2, 12, device=dev)   # d_state not a power of two
    x = torch.randn(1, 8, 24, device=dev)
    with pytest.raises(ApertisHipError):
        ops.selective_scan(d, A, x, x)
    with pytest.raises(ApertisHipError):
        ops.selective_scan(d.cpu(), A.cpu(), x.cpu(), x.cpu())


# ------------------------------------------------------------------ dt_proj_head / column split
@pytest.mark.parametrize("dt,B,L,K,N,ld", [(torch.float32, 2, 300, 22, 11, 64), (torch.bfloat16, 3, 257, 22, 11, 368),
                                            (torch.float32, 1, 5, 1, 1, 1), (torch.bfloat16, 2, 1000, 32, 7, 32),
                                            (torch.float32, 1, 70000, 8, 16, 24), (torch.bfloat16, 2, 700, 44, 11, 400),
                                            (torch.float32, 1, 130, 64, 16, 64)])
def test_tiny_linear_on_a_column_slice(dev, dt, B, L, K, N, ld):
    """ops.tiny_linear (dt_proj_head, reference core.py:382) on p[..., :K] read in place vs F.linear in fp32."""
    import torch.nn.functional as F
    from apertis_llm_amd import ops
    torch.manual_seed(K * 100 + N)
    p_full = torch.randn(B, L, ld).to(dt)
    W, b = torch.randn(N, K) * 0.3, torch.randn(N)
    gy = torch.randn(B, L, N)
    xr = p_full[..., :K].float().clone().requires_grad_(True)
    Wr, br = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    (F.linear(xr, Wr, br) * gy).sum().backward()
    pg = p_full.to(dev).requires_grad_(True)
    Wg, bg = W.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    x = pg[..., :K]
    assert ops.tiny_linear_supported(x, K, N)
    y = ops.tiny_linear(x, Wg, bg)
    assert y.dtype == torch.float32 and y.shape == (B, L, N)
    (y * gy.to(dev)).sum().backward()
    ref_y = F.linear(p_full[..., :K].float(), W, b)
    assert torch.allclose(y.cpu(), ref_y, rtol=1e-5, atol=1e-5)
    tol = 1e-5 if dt == torch.float32 else 1e-2
    assert torch.allclose(pg.grad[..., :K].float().cpu(), xr.grad, rtol=tol, atol=tol)
    assert float(pg.grad[..., K:].abs().max()) == 0.0 if ld > K else True
    scale = (B * L) ** 0.5
    assert torch.allclose(Wg.grad.cpu(), Wr.grad, rtol=1e-4, atol=1e-4 * scale)
    assert torch.allclose(bg.grad.cpu(), br.grad, rtol=1e-4, atol=1e-4 * scale)


def test_split_cols_backward_is_one_concatenation(dev):
    from apertis_llm_amd import ops
    x = torch.randn(2, 7, 20, device=dev, requires_grad=True)
    a, b, c, d = ops.split_cols(x, (3, 8, 8, 1))
    assert a.shape[-1] == 3 and d.shape[-1] == 1 and a.data_ptr() == x.data_ptr()
    (a.sum() * 2 + (c * c).sum()).backward()          # b and d get no gradient at all
    ref = torch.zeros_like(x)
    ref[..., :3] = 2
    ref[..., 11:19] = 2 * x.detach()[..., 11:19]
    assert torch.equal(x.grad, ref)


def test_split_cols_shared_gradient_buffer(dev):
    """Consumers from this module write their input gradients into one buffer (no concatenation); a view used
    twice, a view used by a stock op and an unused view still give the right total gradient."""
    import torch.nn.functional as F
    from apertis_llm_amd import ops
    torch.manual_seed(3)
    B, L, Dn, R = 2, 70, 16, 6
    base = torch.randn(B, L, R + 2 * Dn + 2)
    W, bb = torch.randn(5, R) * 0.3, torch.randn(5)
    wc, bc = torch.randn(Dn, 1, 3) * 0.3, torch.randn(Dn) * 0.1

    def loss(p, split):
        a, b, c, d = split(p)
        y1 = ops.tiny_linear(a, W.to(p.device), bb.to(p.device))            # protocol consumer
        y2 = ops.dwconv_silu(b, wc.to(p.device), bc.to(p.device))           # protocol consumer ...
        y3 = ops.dwconv_silu(b, wc.to(p.device) * 0.5, bc.to(p.device))     # ... of the same view, a second time
        return y1.sum() * 0.7 + (y2 * y2).sum() + y3.sum() + torch.tanh(c).sum()   # c: stock op; d: unused
    ref = base.clone().requires_grad_(True)
    a, b, c, d = ref[..., :R], ref[..., R:R + Dn], ref[..., R + Dn:R + 2 * Dn], ref[..., R + 2 * Dn:]
    y1 = F.linear(a, W, bb)
    def conv(x, w, bias):
        return F.silu(F.conv1d(x.transpose(1, 2), w, bias, groups=Dn, padding=2)[..., :L].transpose(1, 2))
    (y1.sum() * 0.7 + (conv(b, wc, bc) ** 2).sum() + conv(b, wc * 0.5, bc).sum() + torch.tanh(c).sum()).backward()
    p = base.to(dev).requires_grad_(True)
    loss(p, lambda t: ops.split_cols(t, (R, Dn, Dn, 2))).backward()
    assert torch.allclose(p.grad.cpu(), ref.grad, rtol=1e-4, atol=1e-5), float((p.grad.cpu() - ref.grad).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("dt,B,L,Dn", [(torch.bfloat16, 3, 777, 176), (torch.float32, 2, 300, 64), (torch.bfloat16, 2, 4096, 224)])
def test_dwconv_pair_adds_its_two_gradients_in_the_backward_kernel(dev, dt, B, L, Dn):
    """ops.dwconv_silu_pair hands the conv output out twice (one view per consumer) so that the two gradients reach the
    backward kernel separately and are added where the rows are read (apertis_dwconv_silu_bwd2): the same bits as the single
    output followed by autograd's add, also when only one of the two views is used."""
    from apertis_llm_amd import ops
    torch.manual_seed(5)
    x = torch.randn(B, L, Dn).to(dt)
    w, b = torch.randn(Dn, 1, 4) * 0.3, torch.randn(Dn) * 0.1
    g1, g2 = torch.randn(B, L, Dn).to(dt), torch.randn(B, L, Dn).to(dt)

    def run(pair, both=True):
        L_ = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
        if pair:
            ya, yb = ops.dwconv_silu_pair(*L_)
        else:
            ya = yb = ops.dwconv_silu(*L_)
        loss = (ya.float() * g1.to(dev).float()).sum()
        if both:
            loss = loss + (yb.float() * g2.to(dev).float()).sum()
        loss.backward()
        torch.cuda.synchronize()
        return ya.detach(), [t.grad for t in L_]

    for both in (True, False):
        y0, gr0 = run(False, both)
        y1, gr1 = run(True, both)
        assert torch.equal(y0, y1)
        for name, a, c in zip(("dx", "dw", "db"), gr0, gr1):
            assert torch.equal(a, c), (both, name, float((a.float() - c.float()).abs().max()))
