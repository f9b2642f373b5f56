"""Trainer around the hot path: the caller of the data-parallel training step.

Restates the behaviour of the reference's `ApertisTrainer` / `train_from_config`
(src/training/pipeline.py:387-699, :708-990) with the same constructor keywords, config-file schema, step schedule and
checkpoint layout, so that a reference training configuration runs unchanged and its checkpoints load on either side:

  * AdamW, two parameter groups (no decay on names containing 'bias' / 'LayerNorm.bias' / 'LayerNorm.weight');
    OneCycleLR(max_lr=lr, total = ceil(len(loader)/accum) * epochs, pct_start .1, cos, div 25, final_div 1e4);
  * micro-batch loss / accum; an optimizer step (clip to max_grad_norm, step, schedule, zero) when
    (step+1) % accum == 0 or on the epoch's last batch;
  * checkpoints: `step-N` every `checkpoint_steps` optimizer steps, `epoch{e}-iter{i}` every
    `iteration_checkpoint_steps` micro-batches, `best_model` on a new best validation loss, `epoch-N` after every
    epoch, `final` unless stopped; each holds `pytorch_model.bin` (state dict under the reference's key names),
    `config.json` and, for a manual vocabulary, `vocab.json`;
  * `stop_event` is polled before every epoch / batch / evaluation (the reference's UI sets it from another thread);
  * evaluation = mean of the per-batch losses under no_grad; an out-of-memory error halves the batch size and restarts
    the epoch when `dynamic_batch_sizing`.
Deviations: autocast is bf16 without a GradScaler (`fp16=True` selects it); data parallelism is
`BucketedDataParallel` (RCCL, one reduction per optimizer step) instead of DDP; wandb logging is not wired.
`history` records what the reference only logs (loss per optimizer step, learning rates, validation losses,
checkpoint names) - it is what tests/ compare with the reference run captured in tests/golden/trainer_run.npz.
"""
import json
import logging
import math
import os
import shutil
import threading
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from . import ops
from .data import ApertisFineTuneDataset, ApertisPretrainDataset, load_vocabulary
from .model import ApertisConfig, ApertisForCausalLM, create_apertis_model
from .parallel import BucketedDataParallel
from .training import build_optimizer, build_train_prep, clip_and_step, prepared_step

logger = logging.getLogger(__name__)


class ApertisTrainer:
    def __init__(self, model: ApertisForCausalLM, train_dataset, val_dataset=None, output_dir: str = "output",
                 batch_size: int = 4, learning_rate: float = 5e-5, weight_decay: float = 0.01, num_epochs: int = 3,
                 warmup_steps: int = 0, gradient_accumulation_steps: int = 4, max_grad_norm: float = 1.0,
                 use_wandb: bool = False, wandb_project: str = "apertis", wandb_run_name: Optional[str] = None,
                 fp16: bool = True, device: Optional[str] = None, checkpoint_steps: int = 1000,
                 iteration_checkpoint_steps: int = 0, gpu_memory_fraction: float = 0.7,
                 use_gradient_checkpointing: bool = True, eval_every_n_epochs: int = 1, dynamic_batch_sizing: bool = True,
                 gpu_ids: Optional[List[int]] = None, distributed_training: bool = False, local_rank: int = -1,
                 stop_event: Optional[threading.Event] = None, is_fine_tuning: bool = False,
                 original_tokenizer_path_for_ft_hf: Optional[str] = None,
                 original_manual_vocab_path_for_ft: Optional[str] = None, *, shuffle: bool = True, num_workers: int = 4):
        self.model, self.train_dataset, self.val_dataset = model, train_dataset, val_dataset
        self.output_dir, self.batch_size, self.learning_rate = output_dir, batch_size, learning_rate
        self.weight_decay, self.num_epochs, self.warmup_steps = weight_decay, num_epochs, warmup_steps
        self.gradient_accumulation_steps, self.max_grad_norm = gradient_accumulation_steps, max_grad_norm
        self.fp16, self.checkpoint_steps = fp16, checkpoint_steps
        self.iteration_checkpoint_steps, self.eval_every_n_epochs = iteration_checkpoint_steps, eval_every_n_epochs
        self.dynamic_batch_sizing, self.gpu_ids = dynamic_batch_sizing, gpu_ids
        self.distributed_training, self.local_rank = distributed_training, local_rank
        self.stop_event = stop_event if stop_event is not None else threading.Event()
        self.is_fine_tuning = is_fine_tuning
        self.original_tokenizer_path_for_ft_hf = original_tokenizer_path_for_ft_hf
        self.original_manual_vocab_path_for_ft = original_manual_vocab_path_for_ft
        self.use_gradient_checkpointing = use_gradient_checkpointing
        self._shuffle, self._num_workers = shuffle, num_workers
        if use_wandb:
            logger.warning("wandb logging is not wired in this trainer; metrics are kept in trainer.history")
        os.makedirs(output_dir, exist_ok=True)

        self.world_size, self.is_main_process = 1, True
        if distributed_training:
            if self.local_rank == -1:
                self.local_rank = int(os.environ.get("LOCAL_RANK", 0))
            self.device = torch.device(f"cuda:{self.local_rank}" if torch.cuda.is_available() else "cpu")
        elif gpu_ids and torch.cuda.is_available():
            self.device = torch.device(f"cuda:{gpu_ids[0]}")
        elif device is not None:
            self.device = torch.device(device)
        else:
            self.device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
        if self.device.type == "cuda":
            # the HIP kernels launch on the CURRENT device and stream (stock torch ops guard themselves, which is why
            # the reference's trainer gets away without this call, pipeline.py:447-460)
            torch.cuda.set_device(self.device)
        if distributed_training:
            if not dist.is_initialized():
                if self.device.type == "cuda":
                    dist.init_process_group(backend="nccl", device_id=self.device)
                else:
                    dist.init_process_group(backend="gloo")
            self.world_size = dist.get_world_size()
            self.is_main_process = self.local_rank == 0
        if use_gradient_checkpointing and hasattr(model, "gradient_checkpointing_enable"):
            model.gradient_checkpointing_enable()
        model.to(self.device)
        if hasattr(model, "fused_lm_head_loss"):
            model.fused_lm_head_loss = True      # this loop reads only outputs[0]: LM head + CE without the [B, L, V] logits
        self.dp = BucketedDataParallel(model) if (distributed_training and self.world_size > 1) else None

        self._create_dataloaders()
        self.optimizer = build_optimizer(model, learning_rate, weight_decay)
        n_batches = len(self.train_dataloader)
        if n_batches > 0 and gradient_accumulation_steps > 0:
            total = math.ceil(n_batches / gradient_accumulation_steps) * num_epochs
        else:
            total = 1
        self.scheduler = torch.optim.lr_scheduler.OneCycleLR(self.optimizer, max_lr=learning_rate, total_steps=total,
                                                             pct_start=0.1, anneal_strategy="cos", div_factor=25.0,
                                                             final_div_factor=10000.0)
        self._params = [p for p in model.parameters() if p.requires_grad]
        self.history: Dict[str, list] = {"loss": [], "lr": [], "val_loss": [], "checkpoints": []}

    # --------------------------------------------------------------------------------------------------------
    def _create_dataloaders(self):
        pin = self.device.type == "cuda"
        sampler = (DistributedSampler(self.train_dataset, num_replicas=self.world_size, rank=self.local_rank, shuffle=True)
                   if self.distributed_training else None)
        self.train_dataloader = DataLoader(self.train_dataset, batch_size=self.batch_size,
                                           shuffle=(sampler is None and self._shuffle), sampler=sampler,
                                           num_workers=self._num_workers, pin_memory=pin, drop_last=self.distributed_training)
        self.val_dataloader = None
        if self.val_dataset:
            vs = (DistributedSampler(self.val_dataset, num_replicas=self.world_size, rank=self.local_rank, shuffle=False)
                  if self.distributed_training else None)
            self.val_dataloader = DataLoader(self.val_dataset, batch_size=self.batch_size, shuffle=False, sampler=vs,
                                             num_workers=self._num_workers, pin_memory=pin, drop_last=False)

    def _train_prep(self):
        """The one-launch weight preparation of a training micro-step (ops.TrainPrep): built at the first step, bf16 autocast on
        the GPU only."""
        if not hasattr(self, "_prep"):
            self._prep = build_train_prep(self.model) if (self.fp16 and self.device.type == "cuda") else None
        return self._prep

    def _loss(self, batch):
        batch = {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}
        with torch.autocast(self.device.type, dtype=torch.bfloat16, enabled=self.fp16):
            out = self.model(**batch)
        return out[0] if isinstance(out, tuple) else out.loss

    def _optimizer_step(self):
        if self.dp is not None:
            self.dp.finish()
        clip_and_step(self.optimizer, self._params, self.max_grad_norm)
        self.scheduler.step()
        if self.dp is not None:
            self.dp.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    # --------------------------------------------------------------------------------------------------------
    def train(self):
        best_val = float("inf")
        global_step = 0
        accum = self.gradient_accumulation_steps
        epoch = 0
        while epoch < self.num_epochs:
            if self.stop_event.is_set():
                break
            sampler = getattr(self.train_dataloader, "sampler", None)
            if self.distributed_training and hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            self.model.train()
            window_loss, window_n = 0.0, 0
            n_batches = len(self.train_dataloader)
            for step, batch in enumerate(self.train_dataloader):
                if self.stop_event.is_set():
                    break
                try:
                    boundary = (step + 1) % accum == 0 or (step + 1) == n_batches
                    sync = self.dp.no_sync() if (self.dp is not None and not boundary) else _nullcontext()
                    with sync, prepared_step(self._train_prep()):
                        loss = self._loss(batch)
                        if loss is None:
                            continue
                        loss = loss / accum
                        loss.backward()
                    window_loss += loss.item()
                    window_n += 1
                    if loss.is_cuda:      # the host has just waited for this micro-step: a timed-out scan look-back stops the run
                        ops.scan_gate_raise_on_error(loss.device)
                    if boundary:
                        self._optimizer_step()
                        global_step += 1
                        self.history["loss"].append(window_loss * accum / window_n if window_n else 0.0)
                        self.history["lr"].append(self.scheduler.get_last_lr()[0])
                        window_loss, window_n = 0.0, 0
                        if self.checkpoint_steps > 0 and global_step % self.checkpoint_steps == 0 and self.is_main_process:
                            self.save_checkpoint(f"step-{global_step}")
                    if (self.iteration_checkpoint_steps > 0 and (step + 1) % self.iteration_checkpoint_steps == 0
                            and self.is_main_process):
                        self.save_checkpoint(f"epoch{epoch + 1}-iter{step + 1}")
                except Exception as exc:
                    oom = "out of memory" in str(exc).lower()
                    if self.dynamic_batch_sizing and oom and self.batch_size > 1:
                        self.batch_size = max(1, self.batch_size // 2)
                        # as in the reference: the rest of this epoch is abandoned, the end-of-epoch work still runs and
                        # the next epoch uses the smaller batches (its log line says "restarting", its code does this)
                        logger.warning("OOM: reducing batch size to %d", self.batch_size)
                        if torch.cuda.is_available():
                            torch.cuda.empty_cache()
                        self._create_dataloaders()
                        break
                    raise
            if self.stop_event.is_set():
                break
            if self.val_dataloader and self.eval_every_n_epochs > 0 and (epoch + 1) % self.eval_every_n_epochs == 0:
                val = self.evaluate()
                if not math.isinf(val):
                    self.history["val_loss"].append(val)
                    if val < best_val and self.is_main_process:
                        best_val = val
                        self.save_checkpoint("best_model")
            if self.is_main_process:
                self.save_checkpoint(f"epoch-{epoch + 1}")
            epoch += 1
        if self.is_main_process and not self.stop_event.is_set():
            self.save_checkpoint("final")

    def evaluate(self) -> float:
        if not self.val_dataloader:
            return float("inf")
        self.model.eval()
        total, n = 0.0, 0
        with torch.no_grad(), ops.prep_cache_scope():     # (the weights do not change during a validation pass)
            for batch in self.val_dataloader:
                if self.stop_event.is_set():
                    return float("inf")
                try:
                    loss = self._loss(batch)
                except Exception as exc:          # the reference logs and skips a failing validation batch
                    logger.error("Error during validation batch: %s", exc)
                    continue
                if loss is not None:
                    total += loss.item()
                    n += 1
                    if loss.is_cuda:
                        ops.scan_gate_raise_on_error(loss.device)
        self.model.train()
        return total / n if n else float("inf")

    def save_checkpoint(self, name: str):
        ckpt = os.path.join(self.output_dir, name)
        os.makedirs(ckpt, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(ckpt, "pytorch_model.bin"))
        if hasattr(self.model, "config") and hasattr(self.model.config, "save_pretrained"):
            self.model.config.save_pretrained(ckpt)
        tok = getattr(self.train_dataset, "tokenizer", None)
        if self.is_fine_tuning and getattr(self.train_dataset, "is_hf_tokenizer", False) and hasattr(tok, "save_pretrained"):
            tok.save_pretrained(ckpt)
        elif self.original_manual_vocab_path_for_ft and os.path.exists(self.original_manual_vocab_path_for_ft):
            shutil.copy(self.original_manual_vocab_path_for_ft, os.path.join(ckpt, "vocab.json"))
        self.history["checkpoints"].append(name)


class _nullcontext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_SPECIALS = (("<pad>", "pad_token_id"), ("<bos>", "bos_token_id"), ("<eos>", "eos_token_id"), ("<unk>", "unk_token_id"))


def build_from_config(config: Dict[str, Any], stop_event: Optional[threading.Event] = None, **trainer_kwargs) -> ApertisTrainer:
    """Model, datasets and trainer from a reference training configuration (the dict of its JSON file: `data_config`,
    `model_config`, `training_config`; pipeline.py:708-980).  The tokenizer is the source of truth for the vocabulary
    size and the special-token ids; fine-tuning loads `pretrained_model_path_for_finetune` (a checkpoint directory or
    weights file) and carries the overlapping token embeddings over when the vocabulary size changed."""
    data_cfg, model_cfg = config.get("data_config", {}), config.get("model_config", {})
    train_cfg = config.get("training_config", {})
    finetune = train_cfg.get("task_type", "pretrain") == "finetune"
    tok_path = data_cfg.get("tokenizer_path")
    if not tok_path:
        raise ValueError("tokenizer_path is missing in data_config")
    ids = {"pad_token_id": 0, "bos_token_id": 1, "eos_token_id": 2, "unk_token_id": 3}
    hf_tok, vocab = None, None
    if finetune and data_cfg.get("use_hf_tokenizer_for_finetune", False):
        from transformers import AutoTokenizer
        hf_tok = AutoTokenizer.from_pretrained(tok_path)
        vocab_size = hf_tok.vocab_size
        for k in ids:
            if getattr(hf_tok, k) is not None:
                ids[k] = getattr(hf_tok, k)
    else:
        vocab, vocab_size = load_vocabulary(tok_path)
        for tok, k in _SPECIALS:
            if tok in vocab:
                ids[k] = vocab[tok]

    base = train_cfg.get("pretrained_model_path_for_finetune") if finetune else None
    if base:
        cfg_dir = os.path.dirname(base) if os.path.isfile(base) else base
        base_cfg = ApertisConfig.from_pretrained(cfg_dir)
        old_vocab = base_cfg.vocab_size
        merged = base_cfg.to_dict()
        merged.update(model_cfg or {})
        merged.update(ids, vocab_size=vocab_size)
        model = ApertisForCausalLM(ApertisConfig.from_dict(merged))
        weights = base
        if os.path.isdir(base):
            weights = next((os.path.join(base, f) for f in ("pytorch_model.bin", "model.pt")
                            if os.path.exists(os.path.join(base, f))), None)
            if weights is None:
                raise FileNotFoundError(f"No model weights found in dir: {base}")
        sd = torch.load(weights, map_location="cpu", weights_only=True)
        if old_vocab != vocab_size:
            n = min(old_vocab, vocab_size)
            emb = sd.pop("model.token_embeddings.weight", None)
            head = sd.pop("lm_head.weight", None)
            with torch.no_grad():
                if emb is not None:
                    model.model.token_embeddings.weight[:n] = emb[:n]
                if not model.config.tie_word_embeddings and head is not None:
                    model.lm_head.weight[:n] = head[:n]
            model.load_state_dict(sd, strict=False)
        else:
            model.load_state_dict(sd, strict=True)
    else:
        overrides = dict(model_cfg.get("config_overrides", {}))
        overrides.update(ids)
        model = create_apertis_model(
            target_param_count=model_cfg.get("target_param_count", "125M"), vocab_size_override=vocab_size,
            attention_type_override=model_cfg.get("attention_type"), multimodal=model_cfg.get("multimodal", False),
            use_expert_system=model_cfg.get("use_expert_system", False),
            num_experts_target_override=model_cfg.get("num_experts"),
            experts_per_token_target_override=model_cfg.get("experts_per_token"),
            use_flash_attention=model_cfg.get("use_flash_attention", False), ssm_d_inner=model_cfg.get("ssm_d_inner"),
            ssm_d_state=model_cfg.get("ssm_d_state", 16), ssm_dt_rank=model_cfg.get("ssm_dt_rank", "auto"),
            ssm_conv_kernel=model_cfg.get("ssm_conv_kernel", 4), config_overrides=overrides)

    c = model.config
    max_len = data_cfg.get("max_length", 512)

    def dataset(path):
        if finetune:
            return ApertisFineTuneDataset(
                data_path=path, tokenizer=hf_tok if hf_tok is not None else vocab, max_length=max_len,
                prompt_template=data_cfg.get("prompt_template", "User: {instruction}\nAssistant: {output}"),
                is_hf_tokenizer=hf_tok is not None, model_config_vocab_size=c.vocab_size,
                model_config_eos_token_id=c.eos_token_id, model_config_pad_token_id=c.pad_token_id,
                model_config_unk_token_id=c.unk_token_id, model_config_bos_token_id=c.bos_token_id)
        if vocab is None or c.vocab_size == 0:
            raise ValueError("Manual vocabulary must be provided and valid for pre-training.")
        return ApertisPretrainDataset(path, vocab, c.vocab_size, max_len, c.multimodal, data_cfg.get("image_dir"),
                                      c.image_size, pad_token_id_from_config=c.pad_token_id,
                                      unk_token_id_from_config=c.unk_token_id, bos_token_id_from_config=c.bos_token_id,
                                      eos_token_id_from_config=c.eos_token_id)

    train_ds = dataset(data_cfg.get("train_data_path"))
    val_ds = dataset(data_cfg["val_data_path"]) if data_cfg.get("val_data_path") else None
    g = train_cfg.get
    return ApertisTrainer(
        model, train_ds, val_ds, g("output_dir", "output"), g("batch_size", 4), g("learning_rate", 5e-5),
        g("weight_decay", 0.01), g("num_epochs", 3), g("warmup_steps", 0), g("gradient_accumulation_steps", 4),
        g("max_grad_norm", 1.0), g("use_wandb", False), g("wandb_project", "apertis"), g("wandb_run_name"), g("fp16", True),
        g("device"), g("checkpoint_steps", 0), g("iteration_checkpoint_steps", 0), g("gpu_memory_fraction", 0.7),
        g("use_gradient_checkpointing", True), g("eval_every_n_epochs", 1), g("dynamic_batch_sizing", True), g("gpu_ids"),
        g("distributed_training", False), g("local_rank", -1), stop_event=stop_event, is_fine_tuning=finetune,
        original_tokenizer_path_for_ft_hf=tok_path if (finetune and hf_tok is not None) else None,
        original_manual_vocab_path_for_ft=tok_path if vocab is not None else None, **trainer_kwargs)


def train_from_config(config_path: str, stop_event: Optional[threading.Event] = None):
    """The reference's entry point (pipeline.py:708): errors are logged, not raised, so a UI thread survives them."""
    try:
        with open(config_path, "r", encoding="utf-8") as fh:
            config = json.load(fh)
        trainer = build_from_config(config, stop_event)
    except Exception as exc:
        logger.error("Failed to set up training from %s: %s", config_path, exc, exc_info=True)
        return None
    try:
        trainer.train()
    except Exception as exc:
        logger.error("Error during training: %s", exc, exc_info=True)
    return trainer
