"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box; the .npz data
fixtures do).  Import recipe per SURVEY.md §8(c): stub torchvision / wandb in sys.modules, then
`from src.model.core import ...`.  Every fixture is inputs + the reference's outputs; nothing
of the reference's source is stored.

    python tools/gen_golden.py            # writes tests/golden/, prints oracle-vs-reference diffs
"""
import importlib.machinery
import json
import math
import os
import sys
import types
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def _stub(name, classes=()):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for c in classes:
        setattr(m, c, type(c, (), {"__init__": lambda self, *a, **k: None}))
    sys.modules[name] = m
    return m


def import_reference():
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms", ["Compose", "Resize", "ToTensor", "Normalize"])
    _stub("wandb")
    sys.path.insert(0, "/root/reference")
    import src.model.core as core
    return core


def npz(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KB)")


def sd_arrays(sd, prefix="sd::"):
    return {prefix + k: v for k, v in sd.items()}


def gen_scan(core):
    """S2: (delta, A_log, Bt, C[, h0]) -> y, h_L and dy -> d_delta, dA_log, dBt, dC."""
    from oracle import ref_cpu
    cases = [  # name, B, L, h, N, delta_mean, with_h0, dtype
        ("L1", 2, 1, 2, 16, -5.5, False, torch.float32),
        ("L7", 2, 7, 3, 16, -5.5, True, torch.float32),
        ("L64", 1, 64, 2, 16, -5.5, False, torch.float32),
        ("L257", 2, 257, 2, 16, -3.0, True, torch.float32),
        ("L2048", 1, 2048, 1, 4, -5.5, False, torch.float32),
        ("L257_f64", 1, 257, 1, 16, -4.0, False, torch.float64),
        ("bigdelta", 1, 512, 1, 16, -1.0, False, torch.float32),  # sum(delta*|A|) >> 87: the parallel form NaNs
    ]
    for name, B, L, h, N, dmean, with_h0, dt in cases:
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()) % (2 ** 31))
        cfg = core.ApertisConfig(hidden_size=16 * h, num_attention_heads=h, ssm_d_state=N,
                                 attention_type="selective_ssm")
        mod = core.SelectiveLinearAttention(cfg).to(dt)
        delta = torch.nn.functional.softplus(torch.randn(B, L, h, generator=g, dtype=dt) + dmean)
        if name == "bigdelta":
            delta = delta + 0.3
        A_log = (torch.rand(h, N, generator=g, dtype=dt) * (math.log(0.99) - math.log(0.5)) + math.log(0.5))
        Bt = torch.randn(B, L, h * N, generator=g, dtype=dt)
        C = torch.randn(B, L, h * N, generator=g, dtype=dt)
        dy = torch.randn(B, L, h * N, generator=g, dtype=dt)
        h0 = torch.randn(B, h * N, generator=g, dtype=dt) if with_h0 else None
        leaves = [t.clone().requires_grad_(True) for t in (delta, A_log, Bt, C)]
        d_, a_, b_, c_ = leaves
        # reference layouts: delta (B,h,L,1); Bt/C/u (B,h,L,N)  (core.py:383-386)
        to_bhln = lambda t: t.view(B, L, h, N).transpose(1, 2)
        mod.use_cache = with_h0
        res = mod._ssm_pytorch_scan_recurrent(to_bhln(b_), d_.transpose(1, 2).unsqueeze(-1), a_, to_bhln(b_),
                                              to_bhln(c_), None if h0 is None else h0.view(B, h, N))
        y_ref, hl_ref = res if with_h0 else (res, None)
        y_tm = y_ref.transpose(1, 2).contiguous().view(B, L, h * N)        # core.py:394
        y_tm.backward(dy)
        # oracle cross-check
        yo, hlo = ref_cpu.scan_recurrent(delta, A_log, Bt, C, h0)
        go = ref_cpu.scan_backward(delta, A_log, Bt, C, dy, h0)
        errs = [float((yo - y_tm).abs().max())] + [float((o - l.grad).abs().max() / (l.grad.abs().max() + 1e-30))
                                                   for o, l in zip(go, leaves)]
        print(f"  scan_{name}: oracle-vs-reference y abs {errs[0]:.2e}; grad rel (d_delta,dA_log,dBt,dC) "
              + " ".join(f"{e:.1e}" for e in errs[1:]))
        par = mod._ssm_scan_parallel(to_bhln(Bt), delta.transpose(1, 2).unsqueeze(-1), A_log, to_bhln(Bt), to_bhln(C))
        arrs = dict(delta=delta, A_log=A_log, Bt=Bt, C=C, dy=dy, y=y_tm, d_delta=d_.grad, dA_log=a_.grad,
                    dBt=b_.grad, dC=c_.grad, parallel_nonfinite=int((~torch.isfinite(par)).sum()))
        if with_h0:
            arrs.update(h0=h0, h_last=hl_ref.reshape(B, h * N))
        npz("scan_" + name, **arrs)


def gen_ssm_layer(core):
    from oracle import ref_cpu
    torch.manual_seed(11)
    cfg = core.ApertisConfig(hidden_size=48, num_attention_heads=3, ssm_d_state=16, attention_type="selective_ssm")
    mod = core.SelectiveLinearAttention(cfg).eval()
    with torch.no_grad():  # move the SSM parameters off their init so the test is not trivial
        mod.A_log.add_(0.1 * torch.randn_like(mod.A_log))
        mod.D.add_(0.1 * torch.randn_like(mod.D))
    x = torch.randn(2, 37, 48)
    with torch.no_grad():
        out, y_ssm, cache = mod(x, output_attentions=True, use_cache=True)
    sd = {k: v.detach() for k, v in mod.state_dict().items()}
    oo, parts = ref_cpu.ssm_layer(sd, "", x, 3, 16, cfg.ssm_dt_rank, return_parts=True)
    print(f"  ssm_layer: oracle-vs-reference out {float((oo - out).abs().max()):.2e}  "
          f"y {float((parts['y'] - y_ssm).abs().max()):.2e}")
    npz("ssm_layer", x=x, out=out, y_ssm=y_ssm, conv_state=cache[0], ssm_state=cache[1].reshape(2, -1),
        dt_rank=cfg.ssm_dt_rank, **sd_arrays(sd))


def _capture_dispatch(mod, x):
    """Run AdaptiveExpertSystem and record, per expert call, which tokens it received."""
    flat = x.reshape(-1, x.shape[-1])
    calls = []

    def mk(j):
        def hook(_m, inp):
            rows = inp[0]
            eq = (rows.unsqueeze(1) == flat.unsqueeze(0)).all(-1)     # rows are unique random vectors
            calls.append((j, eq.float().argmax(1).tolist()))
        return hook
    hs = [mod.experts[j].register_forward_pre_hook(mk(j)) for j in range(len(mod.experts))]
    out = mod(x)
    for h_ in hs:
        h_.remove()
    return out, calls


def gen_moe(core):
    from oracle import ref_cpu
    for name, training, S_shape, E, K, H, I in [("eval", False, (2, 40), 8, 2, 32, 64),
                                                 ("train_overflow", True, (2, 64), 4, 2, 32, 64),
                                                 ("eval_k3", False, (1, 48), 8, 3, 32, 64)]:
        torch.manual_seed(zlib.crc32(name.encode()) % 1000)
        cfg = core.ApertisConfig(hidden_size=H, intermediate_size=I, num_attention_heads=2, use_expert_system=True,
                                 num_experts=E, experts_per_token=K, hidden_dropout_prob=0.0,
                                 use_noisy_top_k_routing=False, use_expert_dropout=False)
        mod = core.AdaptiveExpertSystem(cfg, activation_function_override="gelu")
        with torch.no_grad():
            for p in mod.parameters():      # default init is tiny/zero: randomise so routing is non-trivial
                p.copy_(torch.randn_like(p) * (0.5 if p.dim() > 1 else 0.2))
            for j in range(E):
                mod.experts[j][0].weight.add_(1.0)
            mod.router_norm.weight.add_(1.0)
            if training:                    # skew the router so that capacity overflows deterministically
                mod.router.bias[0] += 2.0
        mod.train(training)
        x = torch.randn(*S_shape, H)
        with torch.no_grad():
            (out, lb, rz), calls = _capture_dispatch(mod, x)
        sd = {k: v.detach() for k, v in mod.state_dict().items()}
        o_out, o_lb, o_rz, aux = ref_cpu.moe_layer(sd, "", x, E, K, "gelu", cfg.layer_norm_eps, training=training)
        # reference kept sets per (k, e): calls come k-major, experts ascending
        kept = {}
        k_cur, last_e = 0, -1
        for j, toks in calls:
            if j <= last_e:
                k_cur += 1
            last_e = j
            kept[(k_cur, j)] = sorted(toks)
        ref_rows = []
        for e in range(E):
            for k in range(K):
                ref_rows += [(t, k, e) for t in kept.get((k, e), [])]
        offs, row_token, row_k = aux["offsets"], aux["row_token"], aux["row_k"]
        ora_rows = [(int(row_token[r]), int(row_k[r]), e) for e in range(E) for r in range(offs[e], offs[e + 1])]
        gates = aux["gates"]
        srt = torch.sort(gates, dim=-1, descending=True).values
        min_gap = float((srt[:, :K] - srt[:, 1:K + 1]).min())
        print(f"  moe_{name}: out diff {float((o_out - out).abs().max()):.2e} lb {float(abs(o_lb - lb)):.1e} "
              f"rz {float(abs(o_rz - rz)):.1e} rows equal {ora_rows == ref_rows} kept {len(ref_rows)}/{x.shape[0] * x.shape[1] * K} "
              f"min top-k prob gap {min_gap:.2e}")
        assert ora_rows == ref_rows
        npz("moe_" + name, x=x, out=out, lb=lb, rz=rz, logits=aux["logits"], gates=gates, idx=aux["idx"], w=aux["w"],
            kept_rows=np.asarray(ref_rows, dtype=np.int32), expert_offsets=offs, min_gap=min_gap,
            E=E, K=K, training=int(training), capacity=(ref_cpu.expert_capacity(x.shape[0] * x.shape[1], E, 1.25) if training else -1),
            eps=cfg.layer_norm_eps, **sd_arrays(sd))


def gen_vision(core):
    from oracle import ref_cpu
    torch.manual_seed(5)
    cfg = core.ApertisConfig(hidden_size=48, num_attention_heads=3, multimodal=True, image_size=32, vision_embed_dim=32,
                             vision_patch_size=8, vision_layers=2, vision_heads=2)
    from src.multimodal.module import UnifiedMultimodalEncoder
    enc = UnifiedMultimodalEncoder(cfg).eval()
    proj = torch.nn.Linear(32, 48)
    px = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        pe = enc.patch_embed(px).flatten(2).transpose(1, 2)
        feats = enc(px)
        pr = proj(feats)
    sd = {k: v.detach() for k, v in enc.state_dict().items()}
    o_pe = ref_cpu.patch_embed(sd, "", px, 8)[:, 1:] - sd["vision_pos_embed"][:, 1:]
    o_feats = ref_cpu.vision_encoder(sd, "", px, 8, 2, 2)
    print(f"  vision: patch-embed diff {float((o_pe - pe).abs().max()):.2e} encoder diff {float((o_feats - feats).abs().max()):.2e}")
    npz("vision", pixel_values=px, patch_embeds=pe, features=feats, projected=pr, proj_weight=proj.weight, proj_bias=proj.bias,
        **sd_arrays(sd))


def gen_models(core):
    from oracle import ref_cpu
    specs = {
        "model_ssm_dense": dict(use_expert_system=False, multimodal=False),
        "model_ssm_moe": dict(use_expert_system=True, num_experts=4, experts_per_token=2, multimodal=False),
        "model_ssm_moe_mm": dict(use_expert_system=True, num_experts=4, experts_per_token=2, multimodal=True, image_size=32,
                                 vision_embed_dim=24, vision_patch_size=8, vision_layers=1, vision_heads=2),
    }
    for name, extra in specs.items():
        torch.manual_seed(zlib.crc32(name.encode()) % 1000)
        cfg = core.ApertisConfig(vocab_size=96, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                 intermediate_size=64, attention_type="selective_ssm", **extra)
        model = core.ApertisForCausalLM(cfg).eval()
        with torch.no_grad():  # widen the init so logits are not ~0
            for n_, p in model.named_parameters():
                if p.dim() > 1 and "token_embeddings" not in n_:
                    p.mul_(8.0)
        ids = torch.randint(4, 96, (2, 12))
        labels = ids.clone()
        labels[0, :3] = -100
        px = torch.randn(2, 3, 32, 32) if extra.get("multimodal") else None
        with torch.no_grad():
            out = model(input_ids=ids, pixel_values=px, labels=labels, use_cache=False)
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        o_loss, o_logits = ref_cpu.model_forward(sd, dict(cfg.to_dict()), ids, px, labels)
        rel = float(((o_logits - out[1]).abs() / (out[1].abs() + 1e-3)).max())
        print(f"  {name}: loss ref {float(out[0]):.6f} oracle {float(o_loss):.6f}; logits max rel diff {rel:.2e}")
        arrs = dict(input_ids=ids, labels=labels, loss=out[0], logits=out[1], config_json=json.dumps(cfg.to_dict()),
                    **sd_arrays(sd))
        if px is not None:
            arrs["pixel_values"] = px
        npz(name, **arrs)


def gen_dims(core):
    table = {}
    for target, moe in [("125M", False), ("350M", True), ("1.5B", True), ("10M", False), ("7B", False), ("3B", True)]:
        d = core.calculate_model_dimensions(target, 32000, use_expert_system=moe)
        cfg = core.ApertisConfig(vocab_size=32000, hidden_size=d["hidden_size"], num_hidden_layers=d["num_hidden_layers"],
                                 num_attention_heads=d["num_attention_heads"], intermediate_size=d["intermediate_size"],
                                 use_expert_system=moe)
        table[f"{target}|{int(moe)}"] = dict(dims={k: v for k, v in d.items()}, estimate=core.estimate_model_parameters(cfg))
    parse = {s: core.parse_param_count(s) for s in ["10M", "1.5B", "350m", "70B", "125000000", "2.5k"]}
    defaults = core.ApertisConfig().to_dict()
    ssm_cfg = core.ApertisConfig(attention_type="selective_ssm", hidden_size=704, num_attention_heads=11,
                                 use_expert_system=True).to_dict()
    with open(os.path.join(OUT, "config_and_dims.json"), "w") as f:
        json.dump(dict(dims=table, parse=parse, config_defaults=defaults, config_ssm_moe=ssm_cfg), f, indent=1, sort_keys=True)
    print("  wrote config_and_dims.json")


# ------------------------------------------------------------------------------------------------------------
# §8(f) N3 / N2: on-disk formats and the trainer.  The input files are authored here (they are data, written into the
# fixture verbatim); the expected outputs come from the reference's own dataset / trainer classes.
VOCAB_FLAT = {"<pad>": 0, "<bos>": 1, "<eos>": 2, "<unk>": 3, "the": 4, "cat": 5, "sat": 6, "on": 7, "mat": 8, "User:": 9,
              "Assistant:": 10, "what": 11, "is": 12, "a": 13, "dog": 14, "it": 15, "barks": 16, "Q:": 17, "A:": 18,
              "big": 40, "huge": 41}
VOCAB_LIST = {"tokens": ["<pad>", "<bos>", "<eos>", "<unk>", "alpha", "beta", "gamma", "beta"]}
PRETRAIN_LINES = [
    '{"text": "the cat sat on the mat"}',
    '{"text": "the dog barks   and the cat sat"}',
    'this line is not json',
    '{"no_text_here": 1}',
    '{"text": ["the", 5, "zebra", 41, 6]}',
    '{"text": "big huge cat on a mat the cat sat on the mat the cat"}',
    '{"text": ""}',
    '   {"text": "a"}   ',
    '{"text": 17}',
]
FINETUNE_LINES = [
    '{"instruction": "what is a cat", "output": "it sat on the mat"}',
    '{"instruction": "what is a dog", "output": ""}',
    '{"instruction": "only instruction"}',
    'garbage',
    '{"instruction": "what is the big cat on the mat the cat sat on", "output": "it barks it barks it barks"}',
    '{"instruction": "", "output": "the cat"}',
]


HF_VOCAB = {"<pad>": 0, "<bos>": 1, "<eos>": 2, "<unk>": 3, "the": 4, "cat": 5, "sat": 6, "on": 7, "mat": 8, "User": 9,
            "Assistant": 10, "what": 11, "is": 12, "a": 13, "dog": 14, "it": 15, "barks": 16, "Q": 17, "A": 18, ":": 19}


def hf_tokenizer_from_spec(vocab, spec):
    """PreTrainedTokenizerFast over a word-level vocabulary: `bos` = a post-processor that prepends <bos> (and, with
    `eos_in_template`, appends <eos>) when add_special_tokens=True; `pad` = whether a pad token is defined."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.WordLevel(dict(vocab), unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    if spec["bos"]:
        single = "<bos> $A <eos>" if spec["eos_in_template"] else "<bos> $A"
        tok.post_processor = processors.TemplateProcessing(single=single, special_tokens=[("<bos>", vocab["<bos>"]),
                                                                                          ("<eos>", vocab["<eos>"])])
    kw = dict(bos_token="<bos>", eos_token="<eos>", unk_token="<unk>")
    if spec["pad"]:
        kw["pad_token"] = "<pad>"
    return PreTrainedTokenizerFast(tokenizer_object=tok, **kw)


def _items(ds):
    out = []
    for i in range(len(ds)):
        it = ds[i]
        out.append({k: np.asarray(v).tolist() for k, v in it.items()})
    return out


def gen_data_formats(pipe):
    import tempfile
    cases = {}
    with tempfile.TemporaryDirectory() as d:
        def write(name, text):
            path = os.path.join(d, name)
            with open(path, "w", encoding="utf-8") as f:
                f.write(text)
            return path
        vf, vl = write("vocab_flat.json", json.dumps(VOCAB_FLAT)), write("vocab_list.json", json.dumps(VOCAB_LIST))
        pre, ft = write("pre.jsonl", "\n".join(PRETRAIN_LINES) + "\n"), write("ft.jsonl", "\n".join(FINETUNE_LINES) + "\n")
        vocab_cases = {}
        for name, text in [("flat", json.dumps(VOCAB_FLAT)), ("list", json.dumps(VOCAB_LIST)), ("empty_dict", "{}"),
                           ("empty_list", '{"tokens": []}'), ("dup_id", '{"a": 1, "b": 1}'), ("neg_id", '{"a": 0, "b": -2}'),
                           ("dup_then_bad", '{"a": 1, "b": 1, "c": "x"}'), ("bad_then_dup", '{"a": 1.5, "b": 2, "c": 2}'),
                           ("json_list", '["a", "b"]'), ("tokens_not_list", '{"tokens": 5, "x": 6}')]:
            path = write("v_" + name + ".json", text)
            try:
                v, n = pipe._load_vocabulary_and_get_size(path)
                vocab_cases[name] = {"text": text, "vocab": v, "size": n}
            except Exception as e:
                vocab_cases[name] = {"text": text, "error": type(e).__name__, "message_head": str(e).split(" in /")[0][:60]}
        cases["vocab"] = vocab_cases
        flat, n_flat = pipe._load_vocabulary_and_get_size(vf)
        pre_cases = []
        for kw in [dict(model_config_vocab_size=n_flat, max_length=8), dict(model_config_vocab_size=20, max_length=5),
                   dict(model_config_vocab_size=n_flat, max_length=16, pad_token_id_from_config=2, unk_token_id_from_config=1)]:
            ds = pipe.ApertisPretrainDataset(pre, flat, **kw)
            pre_cases.append({"kwargs": kw, "items": _items(ds)})
        cases["pretrain"] = {"lines": PRETRAIN_LINES, "vocab": VOCAB_FLAT, "cases": pre_cases}
        ft_cases = []
        for kw in [dict(max_length=16), dict(max_length=8), dict(max_length=12, prompt_template="Q: {instruction} A: {output}"),
                   dict(max_length=12, prompt_template="no placeholders"),
                   dict(max_length=10, model_config_pad_token_id=2)]:
            full = dict(model_config_vocab_size=n_flat, model_config_eos_token_id=2, model_config_pad_token_id=0,
                        model_config_unk_token_id=3, model_config_bos_token_id=1)
            full.update(kw)
            ds = pipe.ApertisFineTuneDataset(ft, flat, is_hf_tokenizer=False, **full)
            ft_cases.append({"kwargs": full, "items": _items(ds)})
        cases["finetune"] = {"lines": FINETUNE_LINES, "vocab": VOCAB_FLAT, "cases": ft_cases}
        # the Hugging Face tokenizer branch, with tokenizers built offline from a word-level vocabulary (spec stored in the
        # fixture; tests rebuild the same object with hf_tokenizer_from_spec)
        hf_cases = []
        for spec in [dict(bos=True, pad=True, eos_in_template=False), dict(bos=False, pad=True, eos_in_template=False),
                     dict(bos=True, pad=False, eos_in_template=False), dict(bos=True, pad=True, eos_in_template=True)]:
            hf = hf_tokenizer_from_spec(HF_VOCAB, spec)
            for kw in [dict(max_length=16), dict(max_length=7), dict(max_length=12, prompt_template="Q : {instruction} A : {output}")]:
                ds = pipe.ApertisFineTuneDataset(ft, hf, is_hf_tokenizer=True, **kw)
                hf_cases.append({"spec": spec, "kwargs": kw, "items": _items(ds)})
        cases["finetune_hf"] = {"lines": FINETUNE_LINES, "vocab": HF_VOCAB, "cases": hf_cases}
    with open(os.path.join(OUT, "data_formats.json"), "w") as f:
        json.dump(cases, f, indent=0, sort_keys=True)
    print(f"  wrote data_formats.json ({os.path.getsize(os.path.join(OUT, 'data_formats.json')) / 1024:.1f} KB)")


def gen_trainer_run(core, pipe):
    """A whole reference training run on CPU (fp32, dropout / routing noise off so that it is deterministic): the loss
    of every optimizer step, the learning rates, validation losses, checkpoint directory listing and the saved
    config.json key set.  DataLoader shuffling is switched off for the capture (the trainer under test gets
    shuffle=False) - the order of a shuffled epoch depends on how much RNG the model construction consumed."""
    import tempfile
    import torch.utils.data as tud
    rng = np.random.RandomState(7)
    words = [w for w in VOCAB_FLAT if not w.startswith("<")]
    # every text fills max_length: a padded position is the all-zero embedding row (padding_idx), and LayerNorm with the
    # model's eps = 1e-12 turns such a row's aux-loss gradient into 1e13..1e16 (in the reference as well) - the clipped
    # update is then rounding noise, which no two implementations share
    train_lines = [json.dumps({"text": " ".join(rng.choice(words, size=rng.randint(12, 17)))}) for _ in range(10)]
    val_lines = [json.dumps({"text": " ".join(rng.choice(words, size=rng.randint(12, 17)))}) for _ in range(3)]
    cfg_kw = dict(vocab_size=max(VOCAB_FLAT.values()) + 1, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                  intermediate_size=64, attention_type="selective_ssm", use_expert_system=True, num_experts=4,
                  experts_per_token=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                  use_noisy_top_k_routing=False, use_expert_dropout=False, max_position_embeddings=64,
                  # no capacity overflow in this run: identical token rows (same word, same causal history) tie exactly in
                  # the overflow top-n, and torch.topk's pick among equals is not a defined order (the kernels keep the
                  # lowest token index: DESIGN.md section 7); overflow itself is pinned by the moe_train_overflow fixture
                  expert_capacity_factor=4.0)
    real_loader = tud.DataLoader

    class NoShuffle(real_loader):
        def __init__(self, *a, **k):
            k["shuffle"] = False
            k["num_workers"] = 0
            k["pin_memory"] = False
            super().__init__(*a, **k)
    with tempfile.TemporaryDirectory() as d:
        vpath = os.path.join(d, "vocab.json")
        json.dump(VOCAB_FLAT, open(vpath, "w"))
        for name, lines in (("train.jsonl", train_lines), ("val.jsonl", val_lines)):
            open(os.path.join(d, name), "w").write("\n".join(lines) + "\n")
        torch.manual_seed(11)
        model = core.ApertisForCausalLM(core.ApertisConfig(**cfg_kw))
        init_sd = {k: v.clone() for k, v in model.state_dict().items()}
        vocab, n = pipe._load_vocabulary_and_get_size(vpath)
        tr = pipe.ApertisPretrainDataset(os.path.join(d, "train.jsonl"), vocab, n, max_length=12)
        va = pipe.ApertisPretrainDataset(os.path.join(d, "val.jsonl"), vocab, n, max_length=12)
        losses, lrs = [], []
        pipe.DataLoader = NoShuffle
        try:
            trainer = pipe.ApertisTrainer(model, tr, va, output_dir=os.path.join(d, "out"), batch_size=2, learning_rate=1e-3,
                                          num_epochs=2, gradient_accumulation_steps=2, fp16=False, device="cpu",
                                          checkpoint_steps=2, iteration_checkpoint_steps=4, use_gradient_checkpointing=False,
                                          original_manual_vocab_path_for_ft=vpath)
            sched_step = trainer.scheduler.step

            def spy():
                sched_step()
                lrs.append(trainer.scheduler.get_last_lr()[0])
            trainer.scheduler.step = spy
            real_set_postfix = pipe.tqdm.set_postfix if hasattr(pipe.tqdm, "set_postfix") else None

            class Bar:
                def __init__(self, *a, **k): pass
                def set_postfix(self, dct): losses.append(float(dct["loss"]))
                def update(self, n=1): pass
                def close(self): pass
            pipe.tqdm = Bar
            val_losses = []
            real_eval = trainer.evaluate

            def eval_spy():
                v = real_eval()
                val_losses.append(v)
                return v
            trainer.evaluate = eval_spy
            trainer.train()
        finally:
            pipe.DataLoader = real_loader
        out = os.path.join(d, "out")
        listing = {name: sorted(os.listdir(os.path.join(out, name))) for name in sorted(os.listdir(out))}
        cfg_keys = sorted(json.load(open(os.path.join(out, "final", "config.json"))).keys())
        final_sd = torch.load(os.path.join(out, "final", "pytorch_model.bin"), map_location="cpu", weights_only=True)
    meta = dict(cfg=cfg_kw, train_lines=train_lines, val_lines=val_lines, vocab=VOCAB_FLAT, losses_4dp=losses, lrs=lrs,
                val_losses=val_losses, listing=listing, config_keys=cfg_keys,
                trainer=dict(batch_size=2, learning_rate=1e-3, num_epochs=2, gradient_accumulation_steps=2, fp16=False,
                             checkpoint_steps=2, iteration_checkpoint_steps=4, max_length=12))
    arrs = sd_arrays(init_sd, "init::")
    arrs.update(sd_arrays(final_sd, "final::"))
    npz("trainer_run", meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrs)
    print("    optimizer steps:", len(lrs), "losses:", losses, "val:", val_losses, "dirs:", list(listing))


def gen_config1(core):
    """BASELINE config 1 at its stated size: `create-model --target-params 125M` (H896 / 10 layers / 14 heads / I3584,
    vocab 32000), B=2, L=512, forward + loss in eval mode, for BOTH attention types: standard_mha (what the CLI
    builds) and selective_ssm (the north-star path).  The 100 M weights are not stored: both sides rebuild them from
    oracle/seeded.py (one generator per state-dict key).  Stored: the token ids, the loss, and a strided sample of
    the logits."""
    from oracle import ref_cpu, seeded
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(4, 32000, (2, 512), generator=g)
    arrs = dict(input_ids=ids)
    for att in ("standard_mha", "selective_ssm"):
        torch.manual_seed(0)
        m = core.create_apertis_model("125M", vocab_size_override=32000, attention_type_override=att)
        cfg = m.config
        assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size) == (896, 10, 14, 3584)
        sd = seeded.fill_state_dict(m.state_dict())
        m.load_state_dict(sd)
        m.eval()
        with torch.no_grad():
            out = m(input_ids=ids, attention_mask=torch.ones_like(ids), labels=ids)
        arrs[att + "::loss"] = out[0]
        arrs[att + "::logits_sample"] = out[1][:, ::37, ::251].contiguous()
        arrs[att + "::logits_absmax"] = out[1].abs().max()
        arrs[att + "::config_json"] = json.dumps(cfg.to_dict())
        arrs[att + "::n_params"] = sum(p.numel() for p in m.parameters())
        msg = f"  config1 {att}: loss {float(out[0]):.6f}"
        if att == "selective_ssm":
            o_loss, o_logits = ref_cpu.model_forward(sd, dict(cfg.to_dict()), ids, None, ids)
            rel = float(((o_logits - out[1]).abs() / (out[1].abs() + 1e-3)).max())
            msg += f"; oracle {float(o_loss):.6f}, logits max rel diff {rel:.2e}"
        print(msg)
    npz("config1_125m", **arrs)


def gen_generate(core):
    """N1: the reference's generate() (core.py:1520-1644) on toy selective-SSM models, greedy, use_cache=True - the path
    `apertis chat` takes (prefill, then one token per step through the (conv window, SSM state) cache, with the
    reference's front-slice of the cached conv window, core.py:369-373).  Stored: the prompt, the generated token
    sequence, and the last-position logits of every step (captured by wrapping forward)."""
    specs = {"generate_ssm_dense": dict(use_expert_system=False),
             "generate_ssm_moe": dict(use_expert_system=True, num_experts=4, experts_per_token=2)}
    for name, extra in specs.items():
        torch.manual_seed(zlib.crc32(name.encode()) % 1000)
        cfg = core.ApertisConfig(vocab_size=96, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                 intermediate_size=64, attention_type="selective_ssm", **extra)
        model = core.ApertisForCausalLM(cfg).eval()
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if p.dim() > 1 and "token_embeddings" not in n_:
                    p.mul_(8.0)
        prompt = torch.randint(4, 96, (2, 9))
        steps = []
        fwd = model.forward

        def spy(*a, **k):
            out = fwd(*a, **k)
            steps.append(out[1][:, -1, :].detach().clone())
            return out
        model.forward = spy
        with torch.no_grad():
            toks = model.generate(input_ids=prompt, max_new_tokens=16, do_sample=False, use_cache=True, eos_token_id=95)
        model.forward = fwd
        logits = torch.stack(steps, dim=1)                       # [B, steps, V]
        top2 = torch.topk(logits, 2, dim=-1).values
        gap = float((top2[..., 0] - top2[..., 1]).min())
        print(f"  {name}: {toks.shape[1] - prompt.shape[1]} new tokens, {len(steps)} forward calls, min top-2 logit gap {gap:.3e}")
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        npz(name, prompt=prompt, tokens=toks, step_logits=logits, min_gap=gap, config_json=json.dumps(cfg.to_dict()),
            **sd_arrays(sd))


def gen_generate_long(core):
    """N1, the long capture: 56 new tokens at B = 2 with ONE sequence reaching eos mid-way - enough steps
    for the product's captured-graph tail, the stacked-cache pre-pass and the small-batch decode kernels to engage (they need
    >= 24 tokens left), pinned against the reference's own generate() (core.py:1520-1644) instead of against the eager form.
    The eos id is chosen from a first, eos-free pass of the reference: a token sequence 0 emits for the first time at a step in
    [14, 40) and sequence 1 never emits (so sequence 1 decodes to the end while sequence 0 is padded, core.py:1612-1628)."""
    specs = {"generate_ssm_dense_long": dict(use_expert_system=False),
             "generate_ssm_moe_long": dict(use_expert_system=True, num_experts=4, experts_per_token=2)}
    NEW = 56
    for name, extra in specs.items():
      for bump in range(16):     # (the first seed whose capture has no near-tie among the live greedy choices: gap > 1e-3)
        torch.manual_seed(zlib.crc32(name.encode()) % 1000 + bump)
        cfg = core.ApertisConfig(vocab_size=96, hidden_size=32, num_hidden_layers=2, num_attention_heads=2,
                                 intermediate_size=64, attention_type="selective_ssm", **extra)
        model = core.ApertisForCausalLM(cfg).eval()
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if p.dim() > 1 and "token_embeddings" not in n_:
                    p.mul_(8.0)
        prompt = torch.randint(4, 96, (2, 9))
        with torch.no_grad():
            free = model.generate(input_ids=prompt, max_new_tokens=NEW, do_sample=False, use_cache=True, eos_token_id=10 ** 6,
                                  pad_token_id=0)
        new = free[:, prompt.shape[1]:]
        eos, who = None, None
        for b in (0, 1):
            mine, other = new[b].tolist(), set(new[1 - b].tolist())
            for s_ in range(14, 40):
                if mine[s_] not in mine[:s_] and mine[s_] not in other and mine[s_] != 0:
                    eos, who = mine[s_], (b, s_)
                    break
            if eos is not None:
                break
        if eos is None:
            continue
        steps = []
        fwd = model.forward

        def spy(*a, **k):
            out = fwd(*a, **k)
            steps.append(out[1][:, -1, :].detach().clone())
            return out
        model.forward = spy
        with torch.no_grad():
            toks = model.generate(input_ids=prompt, max_new_tokens=NEW, do_sample=False, use_cache=True, eos_token_id=eos,
                                  pad_token_id=0)
        model.forward = fwd
        logits = torch.stack(steps, dim=1)                       # [B, steps, V]
        # the greedy gap over the steps whose choice matters (a finished sequence is padded whatever its logits say)
        newt = toks[:, prompt.shape[1]:]
        live = torch.ones_like(newt, dtype=torch.bool)
        fin = (newt[who[0]] == eos).nonzero()[0, 0].item()
        live[who[0], fin + 1:] = False
        top2 = torch.topk(logits, 2, dim=-1).values
        gap = float((top2[..., 0] - top2[..., 1])[live].min())
        if gap <= 1e-3:
            continue
        print(f"  {name} (seed bump {bump}): {newt.shape[1]} new tokens, {len(steps)} forward calls, eos {eos} ends sequence {who[0]} at step {fin}, "
              f"min live top-2 gap {gap:.3e}")
        assert newt.shape[1] == NEW and fin == who[1] and (newt[who[0], fin + 1:] == 0).all() and (newt[1 - who[0]] != eos).all()
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        npz(name, prompt=prompt, tokens=toks, step_logits=logits, min_gap=gap, eos=eos, eos_seq=who[0], eos_step=fin,
            live=live, config_json=json.dumps(cfg.to_dict()), **sd_arrays(sd))
        break
      else:
        raise SystemExit(f"{name}: no seed in 16 fits the eos rule with a clear greedy gap")


GENERATORS = ["scan", "ssm_layer", "moe", "vision", "models", "dims", "data_formats", "trainer_run", "config1", "generate",
              "generate_long"]


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(4)
    only = [a for a in sys.argv[1:] if not a.startswith("-")] or GENERATORS     # e.g. `gen_golden.py config1`
    unknown = [a for a in only if a not in GENERATORS]
    if unknown:
        raise SystemExit(f"unknown fixture group(s) {unknown}; choose from {GENERATORS}")
    core = import_reference()
    print("reference imported from /root/reference")
    import src.training.pipeline as pipe
    for name in only:
        fn = globals()["gen_" + name]
        if name == "data_formats":
            fn(pipe)
        elif name == "trainer_run":
            fn(core, pipe)
        else:
            fn(core)
