"""CPU ORACLE — test infrastructure only.

A functional (state-dict in, tensors out) restatement of the reference's hot path in plain
PyTorch-CPU / numpy.  It exists to CHECK the HIP path; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  The product package (apertis_llm_amd) never does.

Pinned against the reference itself: tools/gen_golden.py imports /root/reference, runs the
reference modules on seeded inputs and commits the input/output vectors under tests/golden/;
tests/test_host_cpu.py (test_oracle_* at its top) replays them through this file (the reference's own tests hold no
numeric vectors for this path — SURVEY.md §4).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _up(t):
    """The reference's `.float()` promotions (core.py:340, :482): low precision -> fp32.  An fp64 run of this oracle (the
    yardstick the bf16 tests measure both mixed-precision flows against) stays fp64."""
    return t if t.dtype in (torch.float32, torch.float64) else t.float()


# ----------------------------------------------------------------------------------------------
# S2 — selective scan, sequential recurrence (src/model/core.py:337-353)
# ----------------------------------------------------------------------------------------------
def scan_recurrent(delta, A_log, Bt, C, h0=None):
    """delta [B,L,h] (post-softplus), A_log [h,N], Bt/C [B,L,h*N] token-major.
    Returns y [B,L,h*N], h_last [B,h*N].  core.py:339-349 with the [B,h,L,N] transposes undone."""
    B, L, h = delta.shape
    N = A_log.shape[1]
    A = -torch.exp(A_log.to(delta.dtype)).reshape(1, h * N)                      # :339
    a = torch.exp(delta.repeat_interleave(N, dim=2) * A.unsqueeze(0))           # :340-341
    s = torch.zeros(B, h * N, dtype=delta.dtype) if h0 is None else h0.clone()  # :342-345
    ys = []
    for t in range(L):                                                           # :347
        s = a[:, t] * s + Bt[:, t].to(delta.dtype)                              # :348
        ys.append(C[:, t].to(delta.dtype) * s)                                  # :349
    return torch.stack(ys, dim=1), s


def scan_backward(delta, A_log, Bt, C, dy, h0=None):
    """Analytic adjoint of scan_recurrent (SURVEY.md §8a row S-bwd; the reference relies on
    autograd through core.py:347-349).  Returns d_delta [B,L,h], dA_log [h,N], dBt, dC."""
    B, L, h = delta.shape
    N = A_log.shape[1]
    dt = delta.dtype
    A = -torch.exp(A_log.to(dt)).reshape(1, h * N)
    dex = delta.repeat_interleave(N, dim=2)
    a = torch.exp(dex * A.unsqueeze(0))
    s = torch.zeros(B, h * N, dtype=dt) if h0 is None else h0.clone()
    states = [s]
    for t in range(L):
        s = a[:, t] * s + Bt[:, t].to(dt)
        states.append(s)
    dBt = torch.zeros(B, L, h * N, dtype=dt)
    dC = torch.zeros_like(dBt)
    da = torch.zeros_like(dBt)
    lam_next = torch.zeros(B, h * N, dtype=dt)
    for t in range(L - 1, -1, -1):
        g = dy[:, t].to(dt)
        dC[:, t] = g * states[t + 1]
        lam = g * C[:, t].to(dt)
        if t + 1 < L:
            lam = lam + a[:, t + 1] * lam_next
        dBt[:, t] = lam
        da[:, t] = lam * states[t]
        lam_next = lam
    q = da * a * A.unsqueeze(0)                       # da_t * a_t * A
    d_delta = q.reshape(B, L, h, N).sum(-1)
    dA_log = (q * dex).sum(dim=(0, 1)).reshape(h, N)
    return d_delta, dA_log, dBt, dC


def scan_chunked_vectorised(delta, A_log, Bt, C, chunk=64, h0=None):
    """The reference's VECTORISED scan (core.py:324-335: h_t = P_t * cumsum(B_s / P_s), P = exp(cumsum(log a))) applied per
    chunk of `chunk` tokens with the state carried across chunks: L/chunk Python iterations of whole-tensor ops instead of L.
    Over a whole sequence the reference form divides by P -> 0 and goes non-finite (golden `scan_bigdelta` counts them); inside
    64 tokens P >= exp(-64*delta*|A|) stays far from fp32's floor for the model's delta range, so this is the fair vectorised
    CPU timing of BASELINE.md section 3 and agrees with the recurrence to ~1e-5 (tests/test_host_cpu.py)."""
    B, L, h = delta.shape
    N = A_log.shape[1]
    dt = delta.dtype
    A = -torch.exp(A_log.to(dt)).reshape(1, 1, h * N)                                    # :326
    loga = delta.repeat_interleave(N, dim=2) * A                                         # log(Ab) of :327-328, exactly
    ys = []
    s = torch.zeros(B, h * N, dtype=dt) if h0 is None else h0.clone()
    for t0 in range(0, L, chunk):
        P = torch.exp(torch.cumsum(loga[:, t0:t0 + chunk], dim=1))                       # :329-330
        bc = Bt[:, t0:t0 + chunk].to(dt)
        st = P * (torch.cumsum(bc / P, dim=1) + s.unsqueeze(1))                          # :331-333 + the carried-in state
        ys.append(C[:, t0:t0 + chunk].to(dt) * st)                                       # :334
        s = st[:, -1]
    return torch.cat(ys, dim=1), s


# ----------------------------------------------------------------------------------------------
# S1/S4 — SelectiveLinearAttention.forward, prefill (src/model/core.py:355-401)
# ----------------------------------------------------------------------------------------------
def dwconv_silu(xp, w, b):
    """xp [B,L,Dn]; w [Dn,1,k]; causal depthwise conv (pad k-1, keep first L) + SiLU
    (core.py:368-375)."""
    k = w.shape[-1]
    xt = xp.transpose(1, 2)
    xc = F.conv1d(xt, w, b, padding=k - 1, groups=xt.shape[1])[:, :, :xp.shape[1]]
    return F.silu(xc.transpose(1, 2))


def ssm_layer(sd, pre, x, n_heads, d_state, dt_rank, h0=None, return_parts=False, scan=None):
    """sd: state dict, pre: '...attention_mechanism_impl.'; x [B,L,H] (already pre-normed).  `scan`: the scan restatement to
    use (default scan_recurrent = what the reference executes; cpu_baseline also times scan_chunked_vectorised)."""
    Dn = n_heads * d_state
    xp = F.linear(x, sd[pre + "in_proj_x.weight"])                      # :366
    z = F.linear(x, sd[pre + "in_proj_z.weight"])                       # :367
    xc = dwconv_silu(xp, sd[pre + "conv1d.weight"], sd[pre + "conv1d.bias"])  # :368-375
    p = F.linear(xc, sd[pre + "x_param_proj.weight"])                   # :376
    dtf, Bt, C = torch.split(p, [dt_rank, Dn, Dn], dim=-1)               # :377-381
    delta = F.softplus(F.linear(dtf, sd[pre + "dt_proj_head.weight"], sd[pre + "dt_proj_head.bias"]))  # :382-383
    y, h_last = (scan or scan_recurrent)(_up(delta), _up(sd[pre + "A_log"]), Bt, C, h0=h0)   # :391
    ys = y + sd[pre + "D"].view(1, 1, -1) * xc                           # :395
    g = ys * F.silu(z)                                                   # :396
    out = F.linear(g, sd[pre + "out_proj.weight"])                       # :397
    if return_parts:
        return out, dict(xp=xp, z=z, xc=xc, p=p, delta=delta, y=y, gated=g, h_last=h_last)
    return out


# ----------------------------------------------------------------------------------------------
# M1-M5 — AdaptiveExpertSystem.forward (src/model/core.py:470-607)
# ----------------------------------------------------------------------------------------------
def topk_lowest_index_first(g, K):
    """torch.topk(g, K) with a defined tie rule (lowest expert index first)."""
    order = torch.argsort(-g, dim=-1, stable=True)[:, :K]
    return torch.gather(g, 1, order), order


def router(sd, pre, x_flat, K, eps, noise=None):
    """x_flat [S,H] -> logits, gates [S,E], idx [S,K], w [S,K].  core.py:481-492,529."""
    H = x_flat.shape[1]
    xn = F.layer_norm(x_flat, (H,), sd[pre + "router_norm.weight"], sd[pre + "router_norm.bias"], eps)  # :481
    logits = _up(F.linear(xn, sd[pre + "router.weight"], sd[pre + "router.bias"]))                     # :482
    if noise is not None:
        logits = logits + noise                                                                         # :486-488
    gates = F.softmax(logits, dim=-1)                                                                   # :491
    p, idx = topk_lowest_index_first(gates, K)                                                          # :492
    w = p / (p.sum(-1, keepdim=True) + 1e-6)                                                            # :529
    return logits, gates, idx, w


def aux_losses(logits, gates, idx, E, lb_coef, z_coef):
    """core.py:499-505 (load balance) and :524-526 (router z-loss); train mode only."""
    P = gates.mean(0)
    onehot = torch.zeros_like(gates).scatter_(1, idx, 1.0)
    f = onehot.mean(0)
    lb = lb_coef * E * torch.sum(f * P)
    rz = z_coef * torch.mean(torch.logsumexp(logits, dim=-1) ** 2)
    return lb, rz


def expert_capacity(S, E, factor):
    """core.py:508-511 (train mode with use_expert_capacity_limit)."""
    return max(1, math.floor((S / E) * factor)) if S > 0 else 0


def dispatch_plan(idx, w, E, capacity=None, active=None):
    """Integer restatement of the K x E loop core.py:547-591, in canonical expert-major order.

    idx [S,K] int, w [S,K] float (numpy).  capacity None = no limit (eval, :508).
    For expert e the reference consumes capacity k-major (:547,:551,:565-576) and, when a
    (k,e) slot overflows, keeps the `n` candidates with the largest gate weight (:578-582);
    ties -> lowest token first (torch.topk leaves it unspecified; fixtures avoid ties).
    Returns expert_offsets [E+1], row_token [A], row_k [A], slot_of [S,K] (-1 = dropped).
    Rows inside one (e,k) segment are in ascending token order."""
    idx = np.asarray(idx)
    w = np.asarray(w)
    S, K = idx.shape
    offsets = np.zeros(E + 1, dtype=np.int32)
    row_token, row_k = [], []
    slot_of = -np.ones((S, K), dtype=np.int32)
    for e in range(E):
        load = 0
        if active is None or bool(active[e]):                         # :552
            for k in range(K):
                cand = np.nonzero(idx[:, k] == e)[0]                  # :556-561
                n = cand.shape[0]
                if capacity is not None:
                    remaining = capacity - load                       # :568
                    if remaining <= 0:                                # :570
                        continue
                    n = min(n, remaining)                             # :576
                if n < cand.shape[0]:                                 # :578
                    order = np.lexsort((cand, -w[cand, k].astype(np.float64)))[:n]
                    kept = np.sort(cand[order])                       # :580-582 (set), canonical order
                else:
                    kept = cand                                       # :584
                load += kept.shape[0]                                 # :590
                for tkn in kept:
                    slot_of[tkn, k] = len(row_token)
                    row_token.append(int(tkn))
                    row_k.append(k)
        offsets[e + 1] = len(row_token)
    return offsets, np.asarray(row_token, dtype=np.int32), np.asarray(row_k, dtype=np.int32), slot_of


def activation(name):
    if name == "gelu":
        return F.gelu
    if name == "relu":
        return F.relu
    if name in ("silu", "swish"):
        return F.silu
    return F.gelu                                                     # core.py:467-468


def expert_mlp(sd, pre, e, x, act, eps):
    """experts.{e}: LayerNorm -> Linear -> act -> (Dropout) -> Linear.  core.py:434-442."""
    p = f"{pre}experts.{e}."
    H = x.shape[1]
    xn = F.layer_norm(x, (H,), sd[p + "0.weight"], sd[p + "0.bias"], eps)
    u = F.linear(xn, sd[p + "1.weight"], sd[p + "1.bias"])
    return F.linear(activation(act)(u), sd[p + "4.weight"], sd[p + "4.bias"])


def moe_layer(sd, pre, x, E, K, act, eps, training=False, capacity_factor=1.25,
              lb_coef=0.01, z_coef=0.001, noise=None, active=None):
    """Whole AdaptiveExpertSystem.forward in expert-major order (proved equivalent to the
    reference's k-major loop: capacity is per expert).  Dropout is not modelled (eval / p=0)."""
    B, L, H = x.shape
    S = B * L
    xf = x.reshape(S, H)
    logits, gates, idx, w = router(sd, pre, xf, K, eps, noise)
    zero = torch.zeros((), dtype=x.dtype)
    lb, rz = (aux_losses(logits, gates, idx, E, lb_coef, z_coef) if training else (zero, zero))
    cap = expert_capacity(S, E, capacity_factor) if training else None
    offs, row_token, row_k, slot_of = dispatch_plan(idx.numpy(), w.detach().numpy(), E, cap, active)
    out = torch.zeros_like(xf)
    contrib = torch.zeros(S, K, H, dtype=xf.dtype)
    for e in range(E):
        r0, r1 = int(offs[e]), int(offs[e + 1])
        if r1 == r0:
            continue
        tok = torch.from_numpy(row_token[r0:r1]).long()
        kk = torch.from_numpy(row_k[r0:r1]).long()
        ye = expert_mlp(sd, pre, e, xf[tok], act, eps)                 # :593-596
        # (:594,:605: under autocast the expert output is 16-bit and the gate weight fp32 - the product promotes to fp32)
        contrib[tok, kk] = (ye * w[tok, kk].unsqueeze(1)).to(contrib.dtype)
    for k in range(K):                                                 # k-ascending sum == index_add_ order
        out = out + contrib[:, k]
    return out.reshape(B, L, H), lb, rz, dict(logits=logits, gates=gates, idx=idx, w=w, offsets=offs,
                                              row_token=row_token, row_k=row_k, slot_of=slot_of)


# ----------------------------------------------------------------------------------------------
# V1-V3 — UnifiedMultimodalEncoder.forward (src/multimodal/module.py:89-119) and the fusion
# ----------------------------------------------------------------------------------------------
def im2col_patches(pixel_values, p):
    """[B,3,Hi,Wi] -> [B*(Hi/p)*(Wi/p), 3*p*p] with K index c*p*p + ky*p + kx (stride == kernel,
    so Conv2d(3,Dv,p,p) == this matrix times weight.view(Dv,-1).T; module.py:102-103)."""
    B, Cc, Hi, Wi = pixel_values.shape
    gh, gw = Hi // p, Wi // p
    x = pixel_values.reshape(B, Cc, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B * gh * gw, Cc * p * p)


def patch_embed(sd, pre, pixel_values, p):
    B = pixel_values.shape[0]
    W = sd[pre + "patch_embed.weight"]
    cols = im2col_patches(pixel_values, p)
    pe = F.linear(cols, W.reshape(W.shape[0], -1), sd[pre + "patch_embed.bias"]).reshape(B, -1, W.shape[0])
    cls = sd[pre + "cls_token"].expand(B, -1, -1)                       # module.py:106-107
    return torch.cat([cls, pe], dim=1) + sd[pre + "vision_pos_embed"]   # :110


def vit_layer(sd, p, x, n_heads):
    """nn.TransformerEncoderLayer(norm_first=True, activation='gelu', batch_first=True), eval.
    module.py:57-68."""
    B, T, D = x.shape
    h = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    qkv = F.linear(h, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    sh = lambda t: t.reshape(B, T, n_heads, D // n_heads).transpose(1, 2)
    att = torch.softmax(sh(q) @ sh(k).transpose(-1, -2) / math.sqrt(D // n_heads), dim=-1) @ sh(v)
    att = att.transpose(1, 2).reshape(B, T, D)
    x = x + F.linear(att, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    h = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    h = F.linear(F.gelu(F.linear(h, sd[p + "linear1.weight"], sd[p + "linear1.bias"])),
                 sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    return x + h


def vision_encoder(sd, pre, pixel_values, patch, n_layers, n_heads):
    x = patch_embed(sd, pre, pixel_values, patch)
    for i in range(n_layers):
        x = vit_layer(sd, f"{pre}vision_layers.{i}.", x, n_heads)        # module.py:113-114
    D = x.shape[-1]
    return F.layer_norm(x, (D,), sd[pre + "vision_ln.weight"], sd[pre + "vision_ln.bias"], 1e-5)  # :117


# ----------------------------------------------------------------------------------------------
# G1/G2 — ApertisModel / ApertisForCausalLM forward, eval mode (core.py:1142-1307, 1361-1472)
# ----------------------------------------------------------------------------------------------
def model_forward(sd, cfg, input_ids, pixel_values=None, labels=None, aux_out=None, inputs_embeds=None):
    """cfg: dict with the ApertisConfig fields.  selective_ssm attention, LayerNorm, dense-FFN or
    MoE feed-forward; eval mode (no dropout / noise / capacity).  Returns (loss, logits).  aux_out: optional list that
    receives each MoE layer's routing record (gates, idx, ...), for tests that assert a minimum top-k gap.
    inputs_embeds: used instead of the embedding lookup (core.py:1155-1158) - lets a test take the input gradient."""
    H = cfg["hidden_size"]
    eps = cfg["layer_norm_eps"]
    x = inputs_embeds if inputs_embeds is not None else F.embedding(input_ids, sd["model.token_embeddings.weight"])  # :1158
    if cfg.get("multimodal") and pixel_values is not None:                              # :1207-1212
        img = vision_encoder(sd, "model.multimodal_encoder.", pixel_values, cfg["vision_patch_size"],
                             cfg["vision_layers"], cfg["vision_heads"])
        if cfg["vision_embed_dim"] != H:
            img = F.linear(img, sd["model.vision_projection.weight"], sd["model.vision_projection.bias"])
        x = torch.cat([img, x], dim=1)
    lb_tot = torch.zeros((), dtype=x.dtype)
    rz_tot = torch.zeros((), dtype=x.dtype)
    for i in range(cfg["num_hidden_layers"]):
        lp = f"model.layers.{i}."
        h = F.layer_norm(x, (H,), sd[lp + "attention.pre_norm.weight"], sd[lp + "attention.pre_norm.bias"], eps)  # :695
        x = x + ssm_layer(sd, lp + "attention.attention_mechanism_impl.", h, cfg["num_attention_heads"],
                          cfg["ssm_d_state"], cfg["ssm_dt_rank"])                        # :699-704,:836-837
        h = F.layer_norm(x, (H,), sd[lp + "feed_forward.pre_norm.weight"], sd[lp + "feed_forward.pre_norm.bias"], eps)  # :888
        if cfg.get("use_expert_system") and cfg.get("num_experts", 0) > 0:
            f, lb, rz, aux = moe_layer(sd, lp + "feed_forward.ffn.", h, cfg["num_experts"], cfg["experts_per_token"],
                                       cfg["hidden_act"], eps, training=False)
            if aux_out is not None:
                aux_out.append(aux)
        else:
            f = F.linear(activation(cfg["hidden_act"])(F.linear(h, sd[lp + "feed_forward.ffn.0.weight"],
                                                                sd[lp + "feed_forward.ffn.0.bias"])),
                         sd[lp + "feed_forward.ffn.3.weight"], sd[lp + "feed_forward.ffn.3.bias"])  # :870-875
        x = x + f                                                                        # :918-919
    x = F.layer_norm(x, (H,), sd["model.final_post_norm.weight"], sd["model.final_post_norm.bias"], eps)  # :1294
    if cfg.get("multimodal") and pixel_values is not None:
        x = x[:, x.shape[1] - input_ids.shape[1]:]                                       # :1399-1406
    lm_w = sd.get("lm_head.weight", sd["model.token_embeddings.weight"])
    logits = F.linear(x, lm_w)                                                           # :1412
    loss = None
    if labels is not None:                                                               # :1417-1450
        loss = F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1),
                               ignore_index=-100)
        if cfg.get("use_expert_system"):
            loss = loss + lb_tot + rz_tot                                                # :1457-1460 (zeros in eval)
    return loss, logits
