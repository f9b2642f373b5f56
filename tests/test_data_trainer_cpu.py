"""§8(f) N3 / N2 on the CPU: the on-disk formats (vocabulary, pre-training and fine-tuning JSONL) against outputs captured
from the reference's own dataset classes (tests/golden/data_formats.json, tools/gen_golden.py), and the trainer's step /
checkpoint / stop-event schedule on a stand-in model (the real model needs the GPU: tests/test_model_gpu.py holds the
whole-run comparison with the reference)."""
import json
import math
import os
import threading

import numpy as np
import pytest
import torch
import torch.nn as nn

from apertis_llm_amd import data as D
from apertis_llm_amd.trainer import ApertisTrainer

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "data_formats.json")))


def _write(tmp_path, name, text):
    p = tmp_path / name
    p.write_text(text, encoding="utf-8")
    return str(p)


@pytest.mark.parametrize("name", sorted(GOLD["vocab"]))
def test_vocabulary_file_formats(tmp_path, name):
    case = GOLD["vocab"][name]
    path = _write(tmp_path, "v.json", case["text"])
    if "error" in case:
        with pytest.raises(Exception) as ei:
            D.load_vocabulary(path)
        assert type(ei.value).__name__ == case["error"]
        assert str(ei.value).startswith(case["message_head"][:25]), (str(ei.value), case["message_head"])
    else:
        vocab, size = D.load_vocabulary(path)
        assert vocab == case["vocab"] and size == case["size"]


def _same_items(ds, items):
    assert len(ds) == len(items)
    for i, exp in enumerate(items):
        got = ds[i]
        assert sorted(got) == sorted(exp)
        for k, v in exp.items():
            assert got[k].dtype == torch.int64 and got[k].tolist() == v, (i, k, got[k].tolist(), v)


@pytest.mark.parametrize("ci", range(len(GOLD["pretrain"]["cases"])))
def test_pretrain_dataset_matches_reference(tmp_path, ci):
    g = GOLD["pretrain"]
    path = _write(tmp_path, "pre.jsonl", "\n".join(g["lines"]) + "\n")
    case = g["cases"][ci]
    _same_items(D.ApertisPretrainDataset(path, g["vocab"], **case["kwargs"]), case["items"])


@pytest.mark.parametrize("ci", range(len(GOLD["finetune"]["cases"])))
def test_finetune_dataset_matches_reference(tmp_path, ci):
    g = GOLD["finetune"]
    path = _write(tmp_path, "ft.jsonl", "\n".join(g["lines"]) + "\n")
    case = g["cases"][ci]
    _same_items(D.ApertisFineTuneDataset(path, g["vocab"], is_hf_tokenizer=False, **case["kwargs"]), case["items"])


def test_dataset_errors_and_images(tmp_path):
    with pytest.raises(FileNotFoundError):
        D.ApertisPretrainDataset(str(tmp_path / "missing.jsonl"), {}, 10)
    with pytest.raises(ValueError):
        D.ApertisFineTuneDataset(_write(tmp_path, "f.jsonl", ""), {"a": 1}, is_hf_tokenizer=False)   # ids missing
    from PIL import Image
    Image.fromarray((np.arange(30 * 20 * 3) % 255).astype(np.uint8).reshape(30, 20, 3)).save(tmp_path / "im.png")
    path = _write(tmp_path, "mm.jsonl", '{"text": "a", "image": "im.png"}\n{"text": "a", "image": "nope.png"}\n{"text": "a"}\n')
    ds = D.ApertisPretrainDataset(path, {"a": 4}, 10, max_length=4, multimodal=True, image_dir=str(tmp_path), image_size=8)
    px = ds[0]["pixel_values"]
    assert px.shape == (3, 8, 8) and px.dtype == torch.float32 and float(px.abs().max()) < 3.0 and float(px.std()) > 0
    assert torch.equal(ds[1]["pixel_values"], torch.zeros(3, 8, 8)) and "pixel_values" not in ds[2]


class _Toy(nn.Module):
    """7-tuple-style output with the loss first, like ApertisForCausalLM."""

    def __init__(self, vocab=16):
        super().__init__()
        self.emb = nn.Embedding(vocab, 8)
        self.LayerNorm = nn.LayerNorm(8)
        self.out = nn.Linear(8, vocab)

    def forward(self, input_ids, attention_mask=None, labels=None):
        logits = self.out(self.LayerNorm(self.emb(input_ids)))
        loss = nn.functional.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]), labels[:, 1:].reshape(-1),
                                           ignore_index=-100)
        return (loss, logits)


def _toy_data(tmp_path, n):
    rng = np.random.RandomState(0)
    lines = [json.dumps({"text": " ".join(rng.choice(list("abcdefgh"), size=6))}) for _ in range(n)]
    vocab = {c: i + 4 for i, c in enumerate("abcdefgh")}
    return D.ApertisPretrainDataset(_write(tmp_path, f"d{n}.jsonl", "\n".join(lines)), vocab, 16, max_length=8), vocab


def test_trainer_schedule_checkpoints_and_stop(tmp_path):
    torch.manual_seed(0)
    train, vocab = _toy_data(tmp_path, 10)
    val, _ = _toy_data(tmp_path, 3)
    vpath = _write(tmp_path, "vocab.json", json.dumps(vocab))
    out = str(tmp_path / "out")
    t = ApertisTrainer(_Toy(), train, val, output_dir=out, batch_size=2, learning_rate=1e-2, num_epochs=2,
                       gradient_accumulation_steps=2, fp16=False, device="cpu", checkpoint_steps=2, iteration_checkpoint_steps=4,
                       original_manual_vocab_path_for_ft=vpath, shuffle=False, num_workers=0)
    # 5 batches per epoch, accumulation 2 -> optimizer steps after batches 2, 4 and 5 (the epoch's last): 3 per epoch
    assert t.scheduler.total_steps == 6
    groups = t.optimizer.param_groups
    assert groups[0]["weight_decay"] == 0.01 and groups[1]["weight_decay"] == 0.0
    assert sum(p.numel() for p in groups[1]["params"]) == 16 + 8 + 8      # biases and LayerNorm.* only
    t.train()
    assert len(t.history["loss"]) == 6 and len(t.history["val_loss"]) == 2
    ref_opt = torch.optim.SGD([nn.Parameter(torch.zeros(1))], lr=1.0)
    ref = torch.optim.lr_scheduler.OneCycleLR(ref_opt, max_lr=1e-2, total_steps=6, pct_start=0.1, anneal_strategy="cos",
                                              div_factor=25.0, final_div_factor=10000.0)
    want = []
    for _ in range(6):
        ref_opt.step(); ref.step(); want.append(ref.get_last_lr()[0])
    assert np.allclose(t.history["lr"], want, rtol=1e-12)
    assert t.history["checkpoints"] == ["step-2", "epoch1-iter4", "best_model", "epoch-1", "step-4", "epoch2-iter4", "step-6",
                                        "best_model", "epoch-2", "final"] or \
        t.history["checkpoints"] == ["step-2", "epoch1-iter4", "best_model", "epoch-1", "step-4", "epoch2-iter4", "step-6",
                                     "epoch-2", "final"]
    for name in ("step-2", "epoch1-iter4", "best_model", "epoch-2", "final"):
        assert sorted(os.listdir(os.path.join(out, name))) == ["pytorch_model.bin", "vocab.json"]   # the toy has no config
    sd = torch.load(os.path.join(out, "final", "pytorch_model.bin"), weights_only=True)
    assert sorted(sd) == sorted(t.model.state_dict())
    assert t.history["loss"][-1] < t.history["loss"][0]

    # a stop request between batches: no further optimizer steps, no 'final'
    ev = threading.Event()
    out2 = str(tmp_path / "out2")
    t2 = ApertisTrainer(_Toy(), train, None, output_dir=out2, batch_size=2, num_epochs=3, gradient_accumulation_steps=1,
                        fp16=False, device="cpu", checkpoint_steps=0, shuffle=False, num_workers=0, stop_event=ev)
    real = t2._optimizer_step

    def step_then_stop():
        real()
        if len(t2.history["loss"]) + 1 == 2:
            ev.set()
    t2._optimizer_step = step_then_stop
    t2.train()
    assert len(t2.history["loss"]) == 2 and not os.path.exists(os.path.join(out2, "final"))
    assert t2.evaluate() == float("inf")


def test_build_from_config_pretrain_and_finetune_resize(tmp_path):
    """The reference's training-config schema (pipeline.py:708-980): the tokenizer decides vocab size and special ids; a
    fine-tune from a checkpoint with a different vocabulary keeps the overlapping embedding rows.  (Construction only:
    running the model needs the GPU.)"""
    from apertis_llm_amd.trainer import build_from_config
    vocab = {"<pad>": 0, "<bos>": 5, "<eos>": 6, "<unk>": 7, "a": 8, "b": 9, "c": 10}
    vpath = _write(tmp_path, "vocab.json", json.dumps(vocab))
    train = _write(tmp_path, "train.jsonl", "\n".join(json.dumps({"text": "a b c a"}) for _ in range(4)))
    cfg = {"data_config": {"train_data_path": train, "val_data_path": train, "tokenizer_path": vpath, "max_length": 6},
           "model_config": {"target_param_count": "1M", "attention_type": "selective_ssm", "use_expert_system": True,
                            "num_experts": 4, "experts_per_token": 2},
           "training_config": {"output_dir": str(tmp_path / "out"), "batch_size": 2, "num_epochs": 1, "device": "cpu",
                               "gradient_accumulation_steps": 1, "fp16": False}}
    t = build_from_config(cfg, num_workers=0)
    c = t.model.config
    assert (c.vocab_size, c.pad_token_id, c.bos_token_id, c.eos_token_id, c.unk_token_id) == (11, 0, 5, 6, 7)
    assert c.attention_type == "selective_ssm" and c.use_expert_system and c.num_experts == 4
    assert isinstance(t.train_dataset, D.ApertisPretrainDataset) and len(t.val_dataset) == 4
    assert t.original_manual_vocab_path_for_ft == vpath and not t.is_fine_tuning
    assert t.train_dataset[0]["input_ids"].tolist() == [8, 9, 10, 8, 0, 0]
    t.save_checkpoint("base")
    base = os.path.join(str(tmp_path / "out"), "base")
    assert sorted(os.listdir(base)) == ["config.json", "pytorch_model.bin", "vocab.json"]

    big = dict(vocab, d=11, e=12, f=13)
    vpath2 = _write(tmp_path, "vocab2.json", json.dumps(big))
    ft = _write(tmp_path, "ft.jsonl", json.dumps({"instruction": "a b", "output": "c d"}) + "\n")
    cfg2 = {"data_config": {"train_data_path": ft, "tokenizer_path": vpath2, "max_length": 8},
            "model_config": {},
            "training_config": {"task_type": "finetune", "pretrained_model_path_for_finetune": base, "device": "cpu",
                                "output_dir": str(tmp_path / "out2"), "fp16": False}}
    t2 = build_from_config(cfg2, num_workers=0)
    assert t2.is_fine_tuning and t2.model.config.vocab_size == 14 and isinstance(t2.train_dataset, D.ApertisFineTuneDataset)
    old, new = t.model.state_dict(), t2.model.state_dict()
    assert torch.equal(new["model.token_embeddings.weight"][:11], old["model.token_embeddings.weight"])
    k = "model.layers.0.attention.attention_mechanism_impl.A_log"
    assert torch.equal(new[k], old[k])
    item = t2.train_dataset[0]
    assert item["labels"].tolist()[:4] == [-100] * 4 and item["labels"].tolist()[4:7] == [10, 11, 6]


def _hf_from_spec(vocab, spec):
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


@pytest.mark.parametrize("ci", range(len(GOLD["finetune_hf"]["cases"])))
def test_finetune_dataset_hf_tokenizer_matches_reference(tmp_path, ci):
    """The Hugging Face tokenizer branch (prompt-length bookkeeping with BOS / EOS special tokens, truncation, pad falling
    back to eos) with word-level tokenizers built offline from the fixture's spec."""
    g = GOLD["finetune_hf"]
    case = g["cases"][ci]
    path = _write(tmp_path, "ft.jsonl", "\n".join(g["lines"]) + "\n")
    hf = _hf_from_spec(g["vocab"], case["spec"])
    _same_items(D.ApertisFineTuneDataset(path, hf, is_hf_tokenizer=True, **case["kwargs"]), case["items"])
