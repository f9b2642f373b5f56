"""Host-side mirror of the reference's model API for the hot path (Python, as the reference is).

Same importable names, constructor keywords, config fields, state-dict keys and forward
contracts as /root/reference/src/model/core.py, so a checkpoint or a `config.json` written by
either side loads in the other and `apertis train` / `apertis chat` callers need no change.
What differs is underneath: the selective scan, the depthwise conv, the gate, the SSM block's
dense projections, LayerNorm and the sub-block boundaries, the whole MoE dispatch, the expert
GEMMs, the dense FFN, the cross-entropy and the optimizer step run as HIP kernels through
libapertis_hip.so (apertis_llm_amd.ops); stock torch (hipBLASLt) keeps the embedding, the LM head's
GEMMs, the ViT body and the `standard_mha` fallback.  There is no eager fallback for the kernel
paths: off-GPU they raise ApertisHipError.
"""
import functools
import inspect
import json
import logging
import math
import os
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .multimodal import UnifiedMultimodalEncoder

logger = logging.getLogger(__name__)


def _on_input_device(fn):
    """Run a module's forward with its first tensor argument's device current.  Stock torch ops guard themselves; the
    C-ABI kernels launch on the CURRENT HIP device and stream, so a model placed on cuda:1 without
    torch.cuda.set_device(1) (the reference's trainer never calls it, pipeline.py:447-460) would otherwise launch its
    kernels on GPU 0's stream.  No-op when the device is already current or the input is not on a GPU."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        t = next((a for a in (*args, *kwargs.values()) if isinstance(a, torch.Tensor)), None)
        if t is None or not t.is_cuda or t.device.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with ops.device_guard(t):
            return fn(self, *args, **kwargs)
    return wrapper


# ----------------------------------------------------------------------------------------------
# ApertisConfig  (reference core.py:67-256)
# ----------------------------------------------------------------------------------------------
class ApertisConfig:
    """Flat config object; `to_dict()` is the full attribute dict and round-trips through
    `config.json` exactly like the reference's (core.py:212-256)."""

    def __init__(self, vocab_size: int = 32000, hidden_size: int = 768, num_hidden_layers: int = 12,
                 num_attention_heads: int = 12, intermediate_size: int = 3072, hidden_act: str = "gelu",
                 hidden_dropout_prob: float = 0.1, attention_probs_dropout_prob: float = 0.1,
                 max_position_embeddings: int = 2048, type_vocab_size: int = 2, initializer_range: float = 0.02,
                 layer_norm_eps: float = 1e-12, pad_token_id: int = 0, bos_token_id: int = 1, eos_token_id: int = 2,
                 unk_token_id: int = 3, position_embedding_type: str = "rotary", use_cache: bool = True,
                 classifier_dropout: float = None, model_type: str = "apertis", tie_word_embeddings: bool = True,
                 rope_theta: float = 10000.0, sliding_window: Optional[int] = None,
                 attention_type: str = "standard_mha", ssm_d_inner: Optional[int] = None, ssm_d_state: int = 16,
                 ssm_dt_rank: Union[int, str] = "auto", ssm_conv_kernel: int = 4, use_flash_attention: bool = False,
                 use_expert_system: bool = False, num_experts: int = 8, experts_per_token: int = 2,
                 multimodal: bool = False, image_size: int = 224, vision_embed_dim: int = 768,
                 vision_patch_size: int = 16, vision_layers: int = 12, vision_heads: int = 12,
                 output_attentions: bool = False, output_hidden_states: bool = False,
                 load_balancing_loss_coef: float = 0.01, expert_capacity_factor: float = 1.25,
                 noisy_routing_alpha: float = 0.1, expert_dropout_prob: float = 0.1,
                 router_z_loss_coef: float = 0.001, expert_output_gating: bool = False,
                 use_noisy_top_k_routing: bool = True, use_expert_capacity_limit: bool = True,
                 use_expert_dropout: bool = True, use_router_z_loss: bool = True,
                 use_load_balancing_loss: bool = True, use_rmsnorm: bool = False, use_swiglu: bool = False,
                 **kwargs):
        given = dict(locals())
        # attribute order follows the reference so a dumped config.json lists keys identically
        for name in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "hidden_act",
                     "intermediate_size", "hidden_dropout_prob", "attention_probs_dropout_prob",
                     "max_position_embeddings", "type_vocab_size", "initializer_range", "layer_norm_eps",
                     "pad_token_id", "bos_token_id", "eos_token_id", "unk_token_id", "position_embedding_type",
                     "use_cache", "classifier_dropout", "model_type", "tie_word_embeddings", "rope_theta",
                     "sliding_window", "attention_type", "ssm_d_state"):
            setattr(self, name, given[name])
        derived = num_attention_heads * ssm_d_state
        if attention_type == "selective_ssm":                         # core.py:153-157
            if ssm_d_inner is not None and ssm_d_inner != derived:
                logger.warning("ssm_d_inner=%s overridden by num_attention_heads*ssm_d_state=%s", ssm_d_inner, derived)
            self.ssm_d_inner = derived
        else:
            self.ssm_d_inner = 2 * hidden_size if ssm_d_inner is None else ssm_d_inner
        self.ssm_dt_rank = math.ceil(hidden_size / 16) if ssm_dt_rank == "auto" else int(ssm_dt_rank)
        for name in ("ssm_conv_kernel", "use_flash_attention", "use_expert_system", "num_experts",
                     "experts_per_token", "multimodal", "image_size", "vision_embed_dim", "vision_patch_size",
                     "vision_layers", "vision_heads", "output_attentions", "output_hidden_states",
                     "load_balancing_loss_coef", "expert_capacity_factor", "noisy_routing_alpha",
                     "expert_dropout_prob", "router_z_loss_coef", "expert_output_gating", "use_noisy_top_k_routing",
                     "use_expert_capacity_limit", "use_expert_dropout", "use_router_z_loss",
                     "use_load_balancing_loss", "use_rmsnorm", "use_swiglu"):
            setattr(self, name, given[name])
        if not use_expert_system:                                     # core.py:200-204
            self.num_experts = 0
            self.experts_per_token = 0
        else:
            self.experts_per_token = min(num_experts, experts_per_token) if num_experts > 0 else 0
        for key, value in kwargs.items():
            if not hasattr(self, key):
                logger.warning("Ignoring unknown config parameter: %s=%s", key, value)

    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]):
        params = inspect.signature(cls.__init__).parameters
        keep = {k: v for k, v in config_dict.items() if (k in params and k != "self") or k == "kwargs"}
        if keep.get("ssm_dt_rank") == "auto":
            keep["ssm_dt_rank"] = math.ceil(keep.get("hidden_size", 768) / 16)
        return cls(**keep)

    def to_dict(self) -> Dict[str, Any]:
        return self.__dict__

    @classmethod
    def from_pretrained(cls, model_name_or_path: str):
        if os.path.isfile(model_name_or_path) and model_name_or_path.endswith(".json"):
            path = model_name_or_path
        else:
            path = os.path.join(model_name_or_path, "config.json")
            if not os.path.exists(path) and os.path.isdir(model_name_or_path):
                parent = os.path.join(Path(model_name_or_path).parent, "config.json")
                if os.path.exists(parent):
                    path = parent
        if not os.path.exists(path):
            raise FileNotFoundError(f"Config file not found. Looked for: '{path}' based on input '{model_name_or_path}'")
        with open(path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))

    def save_pretrained(self, save_directory: str):
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w", encoding="utf-8") as f:
            json.dump(self.to_dict(), f, indent=2)


def _activation_module(name: str) -> nn.Module:
    if name == "relu":
        return nn.ReLU()
    if name in ("silu", "swish"):
        return nn.SiLU()
    if name != "gelu":
        logger.warning("Unsupported activation %s; using GELU", name)
    return nn.GELU()


def _activation_name(name: str) -> str:
    return name if name in ("gelu", "relu", "silu", "swish") else "gelu"


def _compute_dtype(t: torch.Tensor) -> torch.dtype:
    """bf16 under torch.autocast (the reference trains under fp16 autocast, pipeline.py:533; the
    MI355X build maps that to bf16 and drops the GradScaler), else the tensor's own dtype."""
    if torch.is_autocast_enabled():
        dt = torch.get_autocast_dtype("cuda")
        return torch.bfloat16 if dt in (torch.bfloat16, torch.float16) else dt
    return t.dtype if t.dtype in (torch.float32, torch.bfloat16) else torch.float32


class HipLayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / checkpoint keys) whose GPU forward+backward are the HIP row
    kernels; under autocast it emits the compute dtype directly instead of fp32 + a cast kernel.
    Off-GPU (only the stock-torch standard_mha fallback gets there) it is plain F.layer_norm."""

    def forward(self, x):
        H = x.shape[-1]
        if x.is_cuda and H % 4 == 0 and H <= 4096 and x.dtype in (torch.float32, torch.bfloat16):
            return ops.layer_norm(x, self.weight, self.bias, self.eps, out_dtype=_compute_dtype(x))
        return super().forward(x)

    def forward_pass(self, x):
        """(LayerNorm(x), x) for a pre-norm residual block: adding the residual through the returned x lets
        the backward kernel fold the residual gradient into dx (ops.layer_norm_pass)."""
        H = x.shape[-1]
        if x.is_cuda and H % 4 == 0 and H <= 4096 and x.dtype in (torch.float32, torch.bfloat16):
            return ops.layer_norm_pass(x, self.weight, self.bias, self.eps, out_dtype=_compute_dtype(x))
        return self.forward(x), x


def _norm_pass(norm, x):
    return norm.forward_pass(x) if isinstance(norm, HipLayerNorm) else (norm(x), x)


class _LazyCombine:
    """The MoE block's output before its combine (reference core.py:594,605): expert rows yr [rows,H], gate
    weights w [S,K] and the dispatch plan.  `_enter_block` forms the combine inside the boundary kernel;
    `materialise()` is the stand-alone combine."""
    __slots__ = ("yr", "w", "plan", "shape", "dtype")

    def __init__(self, yr, w, plan, shape, dtype):
        self.yr, self.w, self.plan, self.shape, self.dtype = yr, w, plan, shape, dtype

    def materialise(self):
        return ops.moe_combine(self.yr, self.w, self.plan, out_dtype=self.dtype).reshape(self.shape)


class _StackedPast(list):
    """generate()'s static cache of an all-SSM model as ONE tensor per kind - conv windows [NL,B,Dn,k-1], states [NL,B,Dn] fp32 -
    with the usual per-layer (conv_state, ssm_state) tuples as views into them: the model then runs the cache-only half of a
    token step for every layer at once (ApertisModel._decode_prepass)."""

    def __init__(self, conv_all, state_all, heads, d_state):
        NL, B = conv_all.shape[0], conv_all.shape[1]
        super().__init__((conv_all[l], state_all[l].view(B, heads, d_state)) for l in range(NL))
        self.conv_all, self.state_all = conv_all, state_all
        self._offs = None

    def offsets(self, NL, B):
        if self._offs is None:
            self._offs = (torch.arange(NL + 1, device=self.conv_all.device, dtype=torch.int32) * B).contiguous()
        return self._offs


class _Pending:
    """A sub-block's output that has not been added to the residual stream yet: the consumer either resolves
    it (`residual + dropout(out)`) or folds the add into its own pre-norm (`_enter_block`)."""
    __slots__ = ("out", "res", "drop")

    def __init__(self, out, res, drop):
        self.out, self.res, self.drop = out, res, drop

    def resolve(self):
        out = self.out.materialise() if isinstance(self.out, _LazyCombine) else self.out
        return _dropout_add(self.drop, out, self.res)


def _enter_block(norm, h, router=None):
    """(normalised input, residual) of a pre-norm sub-block.  When `h` is a _Pending boundary and the norm is a
    HipLayerNorm on the GPU, the residual add, its dropout and the norm run as ONE kernel in each direction.
    `router` = (router_norm, router Linear) of the MoE feed-forward this block is: its logits are then formed in the same
    forward pass and ride on the returned activation (`_apertis_router_logits`; AdaptiveExpertSystem.forward picks them up)."""
    if isinstance(h, _Pending):
        out, res = h.out, h.res
        H = res.shape[-1]
        cd = _compute_dtype(res)
        lazy = out if isinstance(out, _LazyCombine) else None
        if (isinstance(norm, HipLayerNorm) and res.is_cuda and tuple(out.shape) == tuple(res.shape) and H % 4 == 0 and
                H <= 4096 and res.dtype in (torch.float32, torch.bfloat16) and
                (res.dtype == torch.float32 or cd == res.dtype) and (lazy is None or lazy.dtype == lazy.yr.dtype == cd)):
            if (router is not None and lazy is None and isinstance(router[0], nn.LayerNorm) and H <= 1024 and
                    router[1].weight.shape[0] in (2, 4, 8) and cd in (torch.float32, torch.bfloat16)):
                rn, rl = router
                y, xn, logits = ops.dropout_add_layer_norm_router(out, res, norm.weight, norm.bias, norm.eps, h.drop.p,
                                                                  h.drop.training, rn.weight, rn.bias, rn.eps, rl.weight,
                                                                  rl.bias, out_dtype=cd)
                xn._apertis_router_logits = logits
                return xn, y
            if lazy is not None:
                y, xn = ops.dropout_add_layer_norm(lazy.yr, res, norm.weight, norm.bias, norm.eps, h.drop.p,
                                                   h.drop.training, out_dtype=cd, combine=(lazy.w, lazy.plan))
            else:
                y, xn = ops.dropout_add_layer_norm(out, res, norm.weight, norm.bias, norm.eps, h.drop.p,
                                                   h.drop.training, out_dtype=cd)
            return xn, y
        h = h.resolve()
    return _norm_pass(norm, h)


def _dropout_add(drop: nn.Dropout, out, residual):
    """residual + dropout(out): one HIP kernel on the GPU, stock torch elsewhere."""
    if out.is_cuda and out.numel() % 4 == 0 and out.shape == residual.shape and \
            out.dtype in (torch.float32, torch.bfloat16) and residual.dtype in (torch.float32, torch.bfloat16) and \
            (residual.dtype == torch.float32 or out.dtype == residual.dtype):
        return ops.dropout_add(out, residual, drop.p, drop.training)
    return drop(out) + residual


# generate(): from this many remaining greedy tokens on, the single-token steps of an SSM model are replayed from a captured HIP
# graph (APERTIS_DECODE_GRAPH=0: always eager).  Capture costs about three eager steps.
DECODE_GRAPH = os.environ.get("APERTIS_DECODE_GRAPH", "1") == "1"
DECODE_GRAPH_MIN_STEPS = 24
# APERTIS_DECODE_PREPASS=0: every layer runs its whole single-token SSM step itself (conv, x_param_proj, state update inside the layer loop)
DECODE_PREPASS = os.environ.get("APERTIS_DECODE_PREPASS", "1") == "1"


def _mfma_linear(x, weight, bias=None):
    """x @ W.T (+b) on the MFMA GEMM tile when the shapes allow 16-byte rows, else stock F.linear
    (both run on the GPU; this is a provider choice, not a fallback to the host)."""
    K = weight.shape[1]
    if x.is_cuda and K % 8 == 0 and weight.shape[0] % 8 == 0:
        return ops.linear_mfma(x, weight, bias, compute_dtype=_compute_dtype(x))
    if getattr(weight, "_apertis_prep", None) is not None:
        # (a TrainPrep placeholder carries shape and gradient route only: its values are uninitialised memory)
        raise ops.ApertisHipError(f"prepared-weight placeholder {tuple(weight.shape)} on the stock linear path")
    return F.linear(x, weight, bias)


class RMSNorm(nn.Module):
    """x / (||x||_2 / sqrt(D) + eps) * scale  (reference core.py:30-59)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.hidden_size, self.eps = hidden_size, eps
        self.scale = nn.Parameter(torch.ones(hidden_size))

    def forward(self, x):
        rms = x.norm(p=2, dim=-1, keepdim=True) * (self.hidden_size ** -0.5)
        return self.scale * (x / (rms + self.eps))


class RotaryEmbedding(nn.Module):
    """Interleaved-pair rotary embedding over the full projection width (reference core.py:258-293)."""

    def __init__(self, dim: int, max_position_embeddings: int = 2048, base: float = 10000):
        super().__init__()
        if dim % 2:
            raise ValueError(f"RoPE dimension must be even, got {dim}")
        self.dim, self.max_position_embeddings, self.base = dim, max_position_embeddings, base
        inv = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
        ang = torch.outer(torch.arange(max_position_embeddings, dtype=torch.float32), inv)
        self.register_buffer("inv_freq", inv, persistent=False)
        self.register_buffer("cos_cached", ang.cos(), persistent=False)
        self.register_buffer("sin_cached", ang.sin(), persistent=False)

    def forward(self, x, position_ids=None):
        B, L, _ = x.shape
        if position_ids is None:
            position_ids = torch.arange(L, device=x.device).unsqueeze(0)
        cos, sin = self.cos_cached[position_ids], self.sin_cached[position_ids]
        pairs = x.float().reshape(B, L, -1, 2)
        a, b = pairs[..., 0], pairs[..., 1]
        return torch.stack((a * cos - b * sin, a * sin + b * cos), dim=-1).reshape(B, L, self.dim).type_as(x)


# ----------------------------------------------------------------------------------------------
# SelectiveLinearAttention  (reference core.py:295-401) on HIP kernels
# ----------------------------------------------------------------------------------------------
class SelectiveLinearAttention(nn.Module):
    """in_proj_x/z -> depthwise causal conv + SiLU -> x_param_proj -> (dt feats, Bt, C) ->
    softplus(dt_proj_head) -> selective scan -> +D*x -> *SiLU(z) -> out_proj.

    One code path for training, prefill and cached decode: the chunked HIP scan reproduces the
    sequential recurrence (core.py:337-353), which is what the reference's trainer actually
    executes (use_cache defaults to True).  The overflow-prone cumsum form (core.py:324-335) is
    not emulated."""

    def __init__(self, config: ApertisConfig):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.d_state = config.ssm_d_state
        self.d_inner = self.num_heads * self.d_state
        self.dt_rank = config.ssm_dt_rank
        self.conv_kernel_size = config.ssm_conv_kernel
        self.in_proj_x = nn.Linear(self.hidden_size, self.d_inner, bias=False)
        self.in_proj_z = nn.Linear(self.hidden_size, self.d_inner, bias=False)
        # kept as nn.Conv1d so parameter names/shapes/initialisation match the checkpoint format
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, self.conv_kernel_size, groups=self.d_inner,
                                padding=self.conv_kernel_size - 1)
        self.x_param_proj = nn.Linear(self.d_inner, self.dt_rank + 2 * self.d_inner, bias=False)
        self.dt_proj_head = nn.Linear(self.dt_rank, self.num_heads, bias=True)
        nn.init.uniform_(self.dt_proj_head.bias, a=math.log(1e-3), b=math.log(1e-2))
        self.A_log = nn.Parameter(torch.empty(self.num_heads, self.d_state).uniform_(math.log(0.5), math.log(0.99)))
        self.D = nn.Parameter(torch.ones(self.d_inner))
        self.out_proj = nn.Linear(self.d_inner, self.hidden_size, bias=False)
        self.use_cache = False
        self._pad_idx = None
        self._inplace_cache = False     # generate()'s graph replay: the single-token step updates the SSM state it is handed
        self._decode_pre = None         # this layer's  C s + D xc  of the current token step, when the model ran it ahead

    def _pad_index(self, device):
        """Destination row of every row of x_param_proj.weight ([dt | Bt | C]) in the padded layout [Bt | 0 | C | 0 | dt | 0]."""
        Dn, R = self.d_inner, self.dt_rank
        Wb = -(-Dn // 64) * 64
        idx = self._pad_idx
        if idx is None or idx.device != device:
            i = torch.arange(R + 2 * Dn)
            dst = torch.where(i < R, 2 * Wb + i, torch.where(i < R + Dn, i - R, Wb + i - R - Dn))
            idx = self._pad_idx = dst.to(device)
        return idx

    def register_train_prep(self, prep):
        """This block's GEMM weights into a TrainPrep (ops/prep.py): stacked in_proj, padded x_param_proj, out_proj."""
        Dn, R = self.d_inner, self.dt_rank
        Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
        prep.add_stack(("in_proj_xz", id(self)), (self.in_proj_x.weight, self.in_proj_z.weight))
        prep.add_rowmap(("x_param_padded", id(self)), self.x_param_proj.weight, self._pad_index(self.x_param_proj.weight.device),
                        2 * Wb + Wr)
        prep.add_plain(self.out_proj.weight)

    def _padded_param_weight(self):
        """x_param_proj.weight with its output columns re-ordered and zero-padded so that, in the GEMM output p, the Bt
        and C column blocks each start on a 128-byte boundary and every row does too (core.py:376-381 reads them as
        [dt | Bt | C]; the scan reads whole 128-byte row segments): rows [Bt | 0 | C | 0 | dt | 0], block widths
        (Wb, Wb, Wr) multiples of 64 columns.  The GEMM tiles are 128 columns wide either way, so the pad costs no MFMA
        work.  Returns (weight [2*Wb + Wr, Dn], Wb, Wr)."""
        w, Dn, R = self.x_param_proj.weight, self.d_inner, self.dt_rank
        Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
        if w.is_cuda:
            # source row i of [dt | Bt | C] lands at: dt -> 2*Wb + i, Bt -> i - R, C -> Wb + i - R - Dn
            return ops.scatter_rows(w, self._pad_index(w.device), 2 * Wb + Wr), Wb, Wr
        zb = w.new_zeros(Wb - Dn, Dn)
        zr = w.new_zeros(Wr - R, Dn)
        return torch.cat([w[R:R + Dn], zb, w[R + Dn:], zb, w[:R], zr], dim=0), Wb, Wr

    def _stacked_in_proj(self):
        """in_proj_x | in_proj_z as one [2 Dn, H] weight (core.py:366-367 share their input): prepared per step / per generate()."""
        w_xz = ops.prepared_weight(("in_proj_xz", id(self)), (self.in_proj_x.weight, self.in_proj_z.weight))
        if w_xz is None:
            w_xz = ops.cached_prep("in_proj_xz", (self.in_proj_x.weight, self.in_proj_z.weight),
                                   lambda: torch.cat([self.in_proj_x.weight, self.in_proj_z.weight], dim=0))
        return w_xz

    def decode_finish(self, gated, past_key_value):
        """out_proj of a single-token step whose gate already ran (ops.decode_inproj's epilogue; `_decode_pre` is consumed)."""
        self._decode_pre = None
        B = gated.shape[0]
        out = ops.decode_dense_gemv(gated, self.out_proj.weight, self.out_proj.bias)
        if out is None:
            out = _mfma_linear(gated.reshape(B, 1, self.d_inner), self.out_proj.weight, self.out_proj.bias)
        return out.reshape(B, 1, -1), None, tuple(past_key_value)

    @_on_input_device
    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions: bool = False, use_cache: bool = False):
        self.use_cache = use_cache
        B, L, _ = hidden_states.shape
        Dn, R, kw = self.d_inner, self.dt_rank, self.conv_kernel_size
        conv_prev = ssm_prev = None
        if past_key_value is not None:
            conv_prev, ssm_prev = past_key_value
        have_window = (conv_prev is not None and use_cache and conv_prev.shape[1] == Dn and conv_prev.shape[2] == kw - 1)
        # in_proj_x and in_proj_z share their input: one GEMM with the stacked weight, outputs are
        # column views (core.py:366-367)
        # (under no_grad - generate() - the stacked and the padded weight are prepared once, not per token)
        # (inside a training step the stacked and the padded weight come prepared - ops.TrainPrep: one launch per step for the
        # whole model - and these are placeholders that carry shape and gradient route only)
        if (self._decode_pre is not None and L == 1 and have_window and hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16
                and not torch.is_grad_enabled() and not output_attentions):
            # single-token step whose cache-only half ran ahead (ApertisModel._decode_prepass): in_proj with the gate and the window
            # push in its epilogue - xz is never written
            gated = ops.decode_inproj(self._stacked_in_proj(), self._decode_pre, conv_prev, xn=hidden_states)
            if gated is not None:
                return self.decode_finish(gated, past_key_value)
        xz = _mfma_linear(hidden_states, self._stacked_in_proj())
        if self._decode_pre is not None and not (L == 1 and have_window and ssm_prev is not None and not output_attentions
                                                  and not torch.is_grad_enabled() and hidden_states.is_cuda and kw > 1):
            self._decode_pre = None
            raise ops.ApertisHipError("SelectiveLinearAttention: the model ran this step's cache-only half ahead (_decode_prepass: the "
                                      "SSM state is already updated) but the layer is not on its single-token path")
        Wb, Wr = -(-Dn // 64) * 64, -(-R // 64) * 64
        wp = ops.prepared_weight(("x_param_padded", id(self)), (self.x_param_proj.weight,))
        if wp is None:
            wp, Wb, Wr = ops.cached_prep("x_param_padded", (self.x_param_proj.weight,), self._padded_param_weight)
        if (L == 1 and have_window and ssm_prev is not None and not output_attentions and not torch.is_grad_enabled()
                and hidden_states.is_cuda and kw > 1):
            # single-token decode step (generate(), core.py:1578-1603): two small kernels around the projections
            # instead of the chunk machinery
            xz2 = xz.reshape(B, 2 * Dn)
            pre, self._decode_pre = self._decode_pre, None
            if pre is not None:
                # the step's first half - conv, x_param_proj, dt, state update: functions of the caches alone - ran for every
                # layer at the start of the token step (ApertisModel._decode_prepass); the gate and the window push are left
                gated = ops.decode_post(pre, xz2, conv_prev)
                out = _mfma_linear(gated.reshape(B, 1, Dn), self.out_proj.weight, self.out_proj.bias)
                return out, None, (conv_prev, ssm_prev)
            xc, conv_state = ops.ssm_decode_step(xz2[:, :Dn], conv_prev, self.conv1d.weight, self.conv1d.bias,
                                                 inplace=self._inplace_cache)
            p = _mfma_linear(xc, wp)                                                   # [B, 2*Wb + Wr]
            dt_in = p[:, 2 * Wb:2 * Wb + R]
            state = ssm_prev.reshape(B, Dn).float()
            if not (self._inplace_cache and state.data_ptr() == ssm_prev.data_ptr()):
                state = state.clone()            # (the captured decode graph owns its cache buffers and updates them in place)
            if ops.tiny_linear_supported(dt_in, R, self.num_heads):
                # dt_proj_head inside the state kernel (one launch less per layer of a token step; the same bits)
                gated = ops.ssm_decode_state_dt(dt_in, self.dt_proj_head.weight, self.dt_proj_head.bias, self.A_log, p[:, :Dn],
                                                p[:, Wb:Wb + Dn], xc, xz2[:, Dn:], self.D, state)
            else:
                dt_logits = self.dt_proj_head(dt_in).float()
                gated = ops.ssm_decode_state(dt_logits, self.A_log, p[:, :Dn], p[:, Wb:Wb + Dn], xc, xz2[:, Dn:], self.D, state)
            out = _mfma_linear(gated.reshape(B, 1, Dn), self.out_proj.weight)
            return out, None, (conv_state, state.reshape(B, self.num_heads, self.d_state))
        xp, z = ops.split_cols(xz, (Dn, Dn))
        conv_in = xp
        if have_window:
            # the reference prepends the cached window and keeps the FIRST L conv outputs
            # (core.py:369-373); reproduced as is so cached decode matches `apertis chat`
            conv_in = torch.cat([conv_prev.transpose(1, 2).to(xp.dtype), xp], dim=1)
        conv_state = conv_in[:, -(kw - 1):].transpose(1, 2).detach() if use_cache else None
        if hidden_states.is_cuda and ops.DWCONV_PAIR:
            # (two views of the one conv output, one per consumer - x_param_proj and the scan: the conv backward adds their
            #  gradients where it reads the rows)
            xc_p, xc = ops.dwconv_silu_pair(conv_in, self.conv1d.weight, self.conv1d.bias)  # core.py:373-375
        else:
            xc_p = xc = ops.dwconv_silu(conv_in, self.conv1d.weight, self.conv1d.bias)
        if have_window:
            xc_p, xc = xc_p[:, :L], xc[:, :L]
        p = _mfma_linear(xc_p, wp)                                           # x_param_proj (core.py:376), padded layout
        # Bt / C / dt are column slices of p taken in place (core.py:382-385); the Bt and C slices keep their zero pad
        Btp, Cp, dt_in = ops.split_cols(p, (Wb, Wb, R, Wr - R))[:3]
        h0 = ssm_prev.reshape(B, Dn) if (use_cache and ssm_prev is not None) else None
        two_op = output_attentions or not hidden_states.is_cuda
        tiny = ops.tiny_linear_supported(dt_in, R, self.num_heads)
        if tiny and not two_op:
            # dt_proj_head, softplus, recurrence, skip and gate as ONE op (core.py:382-396): the logits come from the
            # stand-alone kernel or - APERTIS_SCAN_DT_FUSED=1 - from inside the scan's state pass (the same bits)
            res = ops.scan_gate_dt(dt_in, self.dt_proj_head.weight, self.dt_proj_head.bias, self.A_log, Btp, Cp, xc, z, self.D,
                                   h0=h0, delta_softplus=True, return_last=use_cache)
            (gated, h_last), y = (res if use_cache else (res, None)), None
            out = _mfma_linear(gated, self.out_proj.weight)                 # core.py:397
            cache = (conv_state, h_last.reshape(B, self.num_heads, self.d_state)) if use_cache else None
            return out, None, cache
        if tiny:
            dt_logits = ops.tiny_linear(dt_in, self.dt_proj_head.weight, self.dt_proj_head.bias)   # [B,L,h] fp32
        else:
            dt_logits = self.dt_proj_head(dt_in).float()
        # softplus (core.py:383) is applied inside the scan kernel
        if two_op:
            # the caller wants y_ssm itself (core.py:401): scan and gate as two ops, y in fp32
            res = ops.selective_scan(dt_logits, self.A_log, Btp[..., :Dn], Cp[..., :Dn], h0=h0,
                                     delta_softplus=True, y_dtype=torch.float32, return_last=use_cache)
            y, h_last = res if use_cache else (res, None)
            gated = ops.ssm_gate(y, xc, z, self.D)                           # core.py:395-396
        else:
            # recurrence + skip + gate in one kernel per direction; y is never written (core.py:388-396)
            res = ops.scan_gate(dt_logits, self.A_log, Btp, Cp, xc, z, self.D, h0=h0, delta_softplus=True,
                                return_last=use_cache)
            (gated, h_last), y = (res if use_cache else (res, None)), None
        out = _mfma_linear(gated, self.out_proj.weight)                     # core.py:397
        cache = (conv_state, h_last.reshape(B, self.num_heads, self.d_state)) if use_cache else None
        return out, (y if output_attentions else None), cache


# ----------------------------------------------------------------------------------------------
# AdaptiveExpertSystem  (reference core.py:403-607) on HIP kernels
# ----------------------------------------------------------------------------------------------
_ZERO_SCALARS = {}


def _zero_scalar(device, dtype):
    """A shared 0-dim zero (never written in place): the auxiliary-loss slots of layers without an expert system."""
    key = (device, dtype)
    z = _ZERO_SCALARS.get(key)
    if z is None:
        z = _ZERO_SCALARS[key] = torch.zeros((), device=device, dtype=dtype)
    return z


class AdaptiveExpertSystem(nn.Module):
    """Top-K routed mixture of `LayerNorm -> Linear -> act -> Dropout -> Linear` experts.

    Parameters are held stacked ([E, ...]) for the grouped GEMMs; `state_dict()` /
    `load_state_dict()` expose the reference's per-expert names `experts.{e}.{0,1,4}.*`."""

    _STACKED = (("0.weight", "expert_ln_weight"), ("0.bias", "expert_ln_bias"), ("1.weight", "expert_w1"),
                ("1.bias", "expert_b1"), ("4.weight", "expert_w2"), ("4.bias", "expert_b2"))

    def register_train_prep(self, prep):
        """The stacked expert weights into a TrainPrep (ops/prep.py): bf16 and transposed bf16 copies, one launch per step."""
        prep.add_plain(self.expert_w1)
        prep.add_plain(self.expert_w2)

    def __init__(self, config: ApertisConfig, activation_function_override: Optional[str] = None):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.intermediate_size = config.intermediate_size
        self.num_experts = config.num_experts
        self.experts_per_token = config.experts_per_token
        self.load_balancing_loss_coef = config.load_balancing_loss_coef if self.num_experts > 0 else 0.0
        self.expert_capacity_factor = config.expert_capacity_factor
        self.router_z_loss_coef = config.router_z_loss_coef if self.num_experts > 0 else 0.0
        self.noisy_routing_alpha = config.noisy_routing_alpha if self.num_experts > 0 else 0.0
        self.expert_dropout_prob = config.expert_dropout_prob if self.num_experts > 0 else 0.0
        on = self.num_experts > 0
        self.use_noisy_top_k_routing = config.use_noisy_top_k_routing and on
        self.use_expert_capacity_limit = config.use_expert_capacity_limit and on
        self.use_expert_dropout = config.use_expert_dropout and on
        self.use_router_z_loss = config.use_router_z_loss and on
        self.use_load_balancing_loss = config.use_load_balancing_loss and on
        self.router = self.experts = self.w_noise = None
        if not on:
            return
        E, H, I = self.num_experts, self.hidden_size, self.intermediate_size
        self.activation = _activation_name(activation_function_override or config.hidden_act)
        self.hidden_dropout_prob = config.hidden_dropout_prob
        self.router_norm = HipLayerNorm(H, eps=config.layer_norm_eps)
        self.router = nn.Linear(H, E)
        self.experts = True  # marker: experts exist (the reference holds an nn.ModuleList here)
        k1, k2 = 1.0 / math.sqrt(H), 1.0 / math.sqrt(I)     # nn.Linear default init bounds
        self.expert_ln_weight = nn.Parameter(torch.ones(E, H))
        self.expert_ln_bias = nn.Parameter(torch.zeros(E, H))
        self.expert_w1 = nn.Parameter(torch.empty(E, I, H).uniform_(-k1, k1))
        self.expert_b1 = nn.Parameter(torch.empty(E, I).uniform_(-k1, k1))
        self.expert_w2 = nn.Parameter(torch.empty(E, H, I).uniform_(-k2, k2))
        self.expert_b2 = nn.Parameter(torch.empty(E, H).uniform_(-k2, k2))
        if config.use_noisy_top_k_routing:
            self.w_noise = nn.Parameter(torch.zeros(E))
        self._register_state_dict_hook(self._split_experts)
        self._register_load_state_dict_pre_hook(self._stack_experts)

    # -- checkpoint format: experts.{e}.{0,1,4}.{weight,bias} -----------------------------------
    @staticmethod
    def _split_experts(module, state_dict, prefix, local_metadata):
        for suffix, name in module._STACKED:
            stacked = state_dict.pop(prefix + name)
            for e in range(module.num_experts):
                state_dict[f"{prefix}experts.{e}.{suffix}"] = stacked[e]
        return state_dict

    def _stack_experts(self, state_dict, prefix, *args):
        for suffix, name in self._STACKED:
            keys = [f"{prefix}experts.{e}.{suffix}" for e in range(self.num_experts)]
            if all(k in state_dict for k in keys):
                state_dict[prefix + name] = torch.stack([state_dict.pop(k) for k in keys])

    def init_expert_weights(self, std: float):
        """ApertisModel._init_weights applied to the stacked experts (core.py:1045-1054)."""
        with torch.no_grad():
            self.expert_w1.normal_(0.0, std)
            self.expert_w2.normal_(0.0, std)
            self.expert_b1.zero_()
            self.expert_b2.zero_()
            self.expert_ln_weight.fill_(1.0)
            self.expert_ln_bias.zero_()

    @_on_input_device
    def forward(self, hidden_states, lazy_combine=False, aux_dtype=None):
        """`aux_dtype`: dtype of the two auxiliary losses (core.py:607 casts them to hidden_states.dtype; the layer passes
        the residual stream's dtype, which is what hidden_states has in the reference under autocast - here the
        pre-norm kernels hand over the bf16 activation, and bf16 losses would also cost two mixed-dtype adds per layer)."""
        aux_dtype = aux_dtype or hidden_states.dtype
        if self.num_experts <= 0 or self.router is None:
            zero = _zero_scalar(hidden_states.device, aux_dtype)
            return hidden_states, zero, zero
        B, L, H = hidden_states.shape
        S, E, K = B * L, self.num_experts, self.experts_per_token
        xf = hidden_states.reshape(S, H)
        pre_logits = getattr(hidden_states, "_apertis_router_logits", None)
        if pre_logits is not None:
            # the boundary kernel in front of this block already formed them (_enter_block); the gather op below hands its
            # gradient rows to that op through the link that rides on the activation
            logits = pre_logits.reshape(S, E)
            xf._apertis_rows_link = getattr(hidden_states, "_apertis_rows_link", None)
        elif ops.router_ln_linear_supported(xf, H, E):
            # core.py:481-482 in one pass over x; xf comes back as the pass-through the expert path reads
            logits, xf = ops.router_ln_linear(xf, self.router_norm.weight, self.router_norm.bias, self.router_norm.eps,
                                              self.router.weight, self.router.bias)
        else:
            xn = self.router_norm(xf)                                                     # core.py:481
            if xn.is_cuda and ops.skinny_linear_supported(H, E):
                logits = ops.skinny_linear(xn, self.router.weight, self.router.bias)      # core.py:482 (fp32 out)
            else:
                logits = self.router(xn).float()
        lb_loss = rz_loss = None
        lb_coef = self.load_balancing_loss_coef if (self.use_load_balancing_loss and self.training) else 0.0
        rz_coef = self.router_z_loss_coef if (self.use_router_z_loss and self.training) else 0.0
        noisy = self.use_noisy_top_k_routing and self.training
        cd = _compute_dtype(xf)
        if (not self.training and lb_coef == 0 and rz_coef == 0 and not noisy and logits.is_cuda
                and ops.moe_route_small_supported(logits, xf, K)):
            # a handful of rows under no_grad (the decode step, core.py:1578-1603): gate, plan and gather-LN in one launch
            _g, idx, w, plan, xg = ops.moe_route_small(logits, xf, self.expert_ln_weight, self.expert_ln_bias,
                                                       self.config.layer_norm_eps, K, out_dtype=cd)
            yr = ops.expert_mlp(xg, self.expert_w1, self.expert_b1, self.expert_w2, self.expert_b2, plan.offsets,
                                plan.max_rows, act=self.activation, drop_p=0.0, seed=0, compute_dtype=cd)
            out = _LazyCombine(yr, w, plan, (B, L, H), xf.dtype)
            if not lazy_combine:
                out = out.materialise()
            zero = _zero_scalar(hidden_states.device, aux_dtype)
            return out, zero, zero
        fused_gate = logits.is_cuda and S > 0 and (lb_coef > 0 or rz_coef > 0)
        if noisy and not fused_gate:                                                      # core.py:485-488
            logits = logits + torch.randn_like(logits) * (F.softplus(self.w_noise) * self.noisy_routing_alpha)
        if fused_gate:
            # noise + gate + both auxiliary losses in one pass (core.py:485-505, 524-529)
            nseed = int(torch.empty((), dtype=torch.int64).random_().item()) if noisy else 0
            idx, w, lb, rz = ops.moe_gate_topk_aux(logits, K, lb_coef, rz_coef, self.w_noise if noisy else None,
                                                   self.noisy_routing_alpha, nseed)
            lb_loss = lb if lb_coef > 0 else None
            rz_loss = rz if rz_coef > 0 else None
        else:
            gates, idx, w = ops.moe_gate_topk(logits, K)                                  # core.py:491-492,529
            if lb_coef > 0:                                                               # core.py:499-505
                frac = torch.zeros(E, device=xf.device).index_add_(0, idx.reshape(-1).long(),
                                                                   torch.ones(S * K, device=xf.device)) / S
                lb_loss = lb_coef * E * torch.sum(frac * gates.mean(dim=0))
            if rz_coef > 0:                                                               # core.py:524-526
                rz_loss = rz_coef * torch.mean(torch.logsumexp(logits, dim=-1) ** 2)
        capacity = None
        if self.use_expert_capacity_limit and self.training:                              # core.py:508-511
            capacity = max(1, math.floor((S / E) * self.expert_capacity_factor)) if S > 0 else 0
        active = None
        if self.use_expert_dropout and self.training and self.expert_dropout_prob > 0:    # core.py:514-521
            n_drop = min(math.floor(E * self.expert_dropout_prob), E - 1)
            if n_drop > 0:
                active = torch.ones(E, dtype=torch.bool)
                active[torch.randperm(E)[:n_drop]] = False
                active = active.to(xf.device)
        cd = _compute_dtype(xf)
        plan = ops.moe_plan(idx, w, E, capacity, active)                                  # core.py:547-591
        xg = ops.moe_gather_ln(xf, self.expert_ln_weight, self.expert_ln_bias, plan, self.config.layer_norm_eps,
                               out_dtype=cd)                                              # core.py:593 + :436
        p_drop = self.hidden_dropout_prob if self.training else 0.0
        seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p_drop > 0 else 0
        yr = ops.expert_mlp(xg, self.expert_w1, self.expert_b1, self.expert_w2, self.expert_b2, plan.offsets,
                            plan.max_rows, act=self.activation, drop_p=p_drop, seed=seed,
                            compute_dtype=cd)                                             # core.py:437-440
        out = _LazyCombine(yr, w, plan, (B, L, H), xf.dtype)                              # core.py:594,605
        if not lazy_combine:
            out = out.materialise()
        if lb_loss is None or rz_loss is None:
            zero = _zero_scalar(hidden_states.device, aux_dtype)
            lb_loss = zero if lb_loss is None else lb_loss
            rz_loss = zero if rz_loss is None else rz_loss
        return out, lb_loss.to(aux_dtype), rz_loss.to(aux_dtype)


class StateTrackingRecurrentCell(nn.Module):
    """Present in the reference (core.py:609-637) but never instantiated by any model path; kept
    as an importable name only."""

    def __init__(self, hidden_size: int):
        super().__init__()
        raise NotImplementedError("StateTrackingRecurrentCell is dead code in the reference and is not part of "
                                  "the MI355X hot path")


# ----------------------------------------------------------------------------------------------
# Attention / feed-forward / layer glue  (reference core.py:639-1018)
# ----------------------------------------------------------------------------------------------
class ApertisAttention(nn.Module):
    def __init__(self, config: ApertisConfig):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = self.hidden_size // self.num_attention_heads
        if config.attention_type in ("selective_ssm", "selective_linear"):
            self.attention_mechanism_impl = SelectiveLinearAttention(config)
            self._ssm = True
        else:
            if config.attention_type != "standard_mha":
                logger.error("Unsupported attention_type '%s'; using 'standard_mha'", config.attention_type)
                config.attention_type = "standard_mha"
            bias = config.attention_probs_dropout_prob == 0.0            # reference quirk, core.py:652
            self.q_proj = nn.Linear(self.hidden_size, self.hidden_size, bias=bias)
            self.k_proj = nn.Linear(self.hidden_size, self.hidden_size, bias=bias)
            self.v_proj = nn.Linear(self.hidden_size, self.hidden_size, bias=bias)
            self.out_proj = nn.Linear(self.hidden_size, self.hidden_size, bias=bias)
            self.attention_mechanism_impl = None
            self._ssm = False
        norm = RMSNorm if config.use_rmsnorm else HipLayerNorm
        self.pre_norm = norm(config.hidden_size, eps=config.layer_norm_eps)
        self.output_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.attention_dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.rope = None
        if config.position_embedding_type == "rotary" and not self._ssm:
            self.rope = RotaryEmbedding(config.hidden_size, config.max_position_embeddings, config.rope_theta)

    def _decode_entry(self, h, past_kv, use_c, output_att):
        """A single-token step whose cache-only half ran ahead (the SSM module's `_decode_pre` is set): the block boundary as the
        prologue of the in_proj product (ops.decode_inproj with `boundary`), whose epilogue gates and pushes the window.  None: the general path."""
        impl = self.attention_mechanism_impl
        if (impl._decode_pre is None or not isinstance(h, _Pending) or past_kv is None or not use_c or output_att
                or not isinstance(self.pre_norm, HipLayerNorm) or torch.is_grad_enabled() or self.training):
            return None
        out, res = h.out, h.res
        if res.shape[1] != 1 or not res.is_cuda:
            return None
        lazy = out if isinstance(out, _LazyCombine) else None
        conv_prev = past_kv[0]
        if lazy is not None:
            if not (lazy.yr.dtype == torch.bfloat16 == lazy.dtype):
                return None
            bnd = (lazy.yr, res, self.pre_norm.weight, self.pre_norm.bias, self.pre_norm.eps, (lazy.w, lazy.plan))
        elif isinstance(out, torch.Tensor) and tuple(out.shape) == tuple(res.shape) and out.dtype == torch.bfloat16:
            bnd = (out, res, self.pre_norm.weight, self.pre_norm.bias, self.pre_norm.eps, None)
        else:
            return None
        if res.shape[0] > 4 or conv_prev.dim() != 3:
            return None
        r = ops.decode_inproj(impl._stacked_in_proj(), impl._decode_pre, conv_prev, boundary=bnd)
        if r is None:
            return None
        y, gated = r
        return y, impl.decode_finish(gated, past_kv)

    def _heads(self, t):
        B, L, _ = t.shape
        return t.view(B, L, self.num_attention_heads, self.attention_head_size).transpose(1, 2)

    def forward(self, hidden_s, att_mask=None, pos_ids=None, past_kv=None, output_att=False, use_c=False, defer=False):
        """hidden_s: the residual stream, or the previous sub-block's _Pending output (its residual add is then
        folded into this block's pre-norm).  defer=True returns this block's output as a _Pending too."""
        fused = self._decode_entry(hidden_s, past_kv, use_c, output_att) if self._ssm else None
        if fused is not None:
            hidden_s, (out, proxy, cache) = fused
        else:
            x, hidden_s = _enter_block(self.pre_norm, hidden_s)
        if fused is not None:
            pass
        elif self._ssm:
            out, proxy, cache = self.attention_mechanism_impl(x, attention_mask=att_mask, position_ids=pos_ids,
                                                              past_key_value=past_kv, output_attentions=output_att,
                                                              use_cache=use_c)
        else:
            # standard_mha is outside the accelerated path: stock torch.
            q, k, v = self.q_proj(x), self.k_proj(x), self.v_proj(x)
            if self.rope is not None:
                q, k = self.rope(q, pos_ids), self.rope(k, pos_ids)
            if use_c and past_kv is not None:
                k, v = torch.cat([past_kv[0], k], dim=1), torch.cat([past_kv[1], v], dim=1)
            cache = (k, v) if use_c else None
            qh, kh, vh = self._heads(q), self._heads(k), self._heads(v)
            Lq, Lk = qh.shape[2], kh.shape[2]
            if att_mask is None and Lq > 1:
                i = torch.arange(Lq, device=x.device).unsqueeze(1) + (Lk - Lq)
                att_mask = torch.zeros(Lq, Lk, device=x.device, dtype=qh.dtype).masked_fill_(
                    i < torch.arange(Lk, device=x.device).unsqueeze(0), torch.finfo(qh.dtype).min)
            if output_att:
                scores = qh @ kh.transpose(-1, -2) / math.sqrt(self.attention_head_size)
                probs = F.softmax(scores + att_mask if att_mask is not None else scores, dim=-1)
                proxy = probs
                ctxv = self.attention_dropout(probs) @ vh
            else:
                proxy = None
                ctxv = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=att_mask,
                                                      dropout_p=self.attention_dropout.p if self.training else 0.0)
            out = self.out_proj(ctxv.transpose(1, 2).reshape(x.shape[0], Lq, self.hidden_size))
        if defer:
            return _Pending(out, hidden_s, self.output_dropout), proxy, cache
        return _dropout_add(self.output_dropout, out, hidden_s), proxy, cache


class SwiGLUFFN(nn.Module):
    """w_down(silu(w_gate x) * w_up x), width 2/3*I rounded up to 256 (reference core.py:925-993)."""

    def __init__(self, config: ApertisConfig, intermediate_size: Optional[int] = None):
        super().__init__()
        self.config = config
        self.hidden_size = config.hidden_size
        self.intermediate_size = intermediate_size if intermediate_size is not None else config.intermediate_size
        self.ffn_dim = max(256, -(-int(self.intermediate_size * 2 / 3) // 256) * 256)
        self.w_gate = nn.Linear(self.hidden_size, self.ffn_dim, bias=False)
        self.w_up = nn.Linear(self.hidden_size, self.ffn_dim, bias=False)
        self.w_down = nn.Linear(self.ffn_dim, self.hidden_size, bias=False)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, x):
        return self.dropout(self.w_down(F.silu(self.w_gate(x)) * self.w_up(x)))


class ApertisFeedForward(nn.Module):
    def __init__(self, config: ApertisConfig):
        super().__init__()
        self.config = config
        norm = RMSNorm if config.use_rmsnorm else HipLayerNorm
        self.pre_norm = norm(config.hidden_size, eps=config.layer_norm_eps)
        self.is_expert_system = False
        if config.use_swiglu:                                            # SwiGLU wins over MoE, core.py:849-859
            self.ffn = SwiGLUFFN(config)
        elif config.use_expert_system and config.num_experts > 0:
            self.ffn = AdaptiveExpertSystem(config, activation_function_override=config.hidden_act)
            self.is_expert_system = True
        else:
            self.ffn = nn.Sequential(nn.Linear(config.hidden_size, config.intermediate_size),
                                     _activation_module(config.hidden_act), nn.Dropout(config.hidden_dropout_prob),
                                     nn.Linear(config.intermediate_size, config.hidden_size))
        self.output_dropout = nn.Dropout(config.hidden_dropout_prob)

    def register_train_prep(self, prep):
        """The plain dense FFN's two weights into a TrainPrep (the expert system registers its own)."""
        if isinstance(self.ffn, nn.Sequential):
            prep.add_plain(self.ffn[0].weight)
            prep.add_plain(self.ffn[3].weight)

    def _dense_ffn(self, x):
        """Linear -> act -> Dropout -> Linear (core.py:861-866).  On the GPU the plain (non-MoE, non-SwiGLU) FFN runs on the
        expert-MLP kernels as ONE group: activation and dropout in the first GEMM's epilogue, their backward in the second data
        gradient's (ops.expert_mlp) - as nn.Sequential it was two library GEMMs plus an activation and a dropout kernel each
        way (13 % of the 125m configuration's step in those four elementwise kernels).  The dropout mask is the counter
        hash of the other fused kernels, not torch's Philox stream (train-mode RNG cannot match the reference bit-wise either way)."""
        if isinstance(self.ffn, nn.Sequential) and x.is_cuda and x.dtype in (torch.float32, torch.bfloat16):
            lin1, _act, drop, lin2 = self.ffn
            H = x.shape[-1]
            xf = x.reshape(-1, H)
            T = xf.shape[0]
            if T >= 1 and lin1.bias is not None and lin2.bias is not None:
                cd = _compute_dtype(xf)
                p_drop = drop.p if self.training else 0.0
                seed = int(torch.empty((), dtype=torch.int64).random_().item()) if p_drop > 0 else 0
                out = ops.expert_mlp(xf.to(cd), lin1.weight.unsqueeze(0), lin1.bias.unsqueeze(0), lin2.weight.unsqueeze(0),
                                     lin2.bias.unsqueeze(0), ops.dense_offsets(T, xf.device), T,
                                     act=_activation_name(self.config.hidden_act), drop_p=p_drop, seed=seed, compute_dtype=cd)
                return out.reshape(*x.shape[:-1], lin2.weight.shape[0])
        return self.ffn(x)

    def forward(self, hidden_s, defer=False):
        router = None
        if self.is_expert_system and self.ffn.router is not None and self.ffn.num_experts > 0:
            router = (self.ffn.router_norm, self.ffn.router)
            small = self._small_entry(hidden_s, defer)
            if small is not None:
                return small
        x, hidden_s = _enter_block(self.pre_norm, hidden_s, router=router)
        if self.is_expert_system:
            out, lb, rz = self.ffn(x, lazy_combine=defer and not os.environ.get("APERTIS_NO_LAZY_COMBINE"),
                                   aux_dtype=hidden_s.dtype)
        else:
            out = self._dense_ffn(x)
            lb = rz = _zero_scalar(hidden_s.device, hidden_s.dtype)
        if defer:
            return _Pending(out, hidden_s, self.output_dropout), lb, rz
        return _dropout_add(self.output_dropout, out, hidden_s), lb, rz


def _ffn_small_entry(self, h, defer):
    """A handful of rows under no_grad (the single-token decode step, core.py:1578-1603): the block boundary, the router, the
    gate, the dispatch plan and the per-expert LayerNorm as ONE launch (ops.moe_enter_small), then the two expert GEMMs.
    None when the shapes / modes are not the ones that kernel takes (the general path runs then)."""
    f = self.ffn
    if not (isinstance(h, _Pending) and not isinstance(h.out, _LazyCombine) and not self.training and not f.training
            and isinstance(self.pre_norm, HipLayerNorm) and isinstance(f.router_norm, nn.LayerNorm)
            and h.res.is_cuda and _compute_dtype(h.res) == h.out.dtype
            and ops.moe_enter_small_supported(h.out, h.res, f.num_experts, f.experts_per_token)):
        return None
    B, L, H = h.res.shape
    cd = h.out.dtype
    y, _logits, w, plan, xg = ops.moe_enter_small(h.out, h.res, self.pre_norm.weight, self.pre_norm.bias, self.pre_norm.eps,
                                                  f.router_norm.weight, f.router_norm.bias, f.router_norm.eps, f.router.weight,
                                                  f.router.bias, f.expert_ln_weight, f.expert_ln_bias, f.config.layer_norm_eps,
                                                  f.experts_per_token)
    yr = ops.expert_mlp(xg, f.expert_w1, f.expert_b1, f.expert_w2, f.expert_b2, plan.offsets, plan.max_rows, act=f.activation,
                        drop_p=0.0, seed=0, compute_dtype=cd)
    out = _LazyCombine(yr, w, plan, (B, L, H), cd)
    zero = _zero_scalar(y.device, y.dtype)
    if defer and not os.environ.get("APERTIS_NO_LAZY_COMBINE"):
        return _Pending(out, y, self.output_dropout), zero, zero
    out = out.materialise()
    if defer:
        return _Pending(out, y, self.output_dropout), zero, zero
    return _dropout_add(self.output_dropout, out, y), zero, zero


ApertisFeedForward._small_entry = _ffn_small_entry


class ApertisLayer(nn.Module):
    def __init__(self, config: ApertisConfig):
        super().__init__()
        self.config = config
        self.attention = ApertisAttention(config)
        self.feed_forward = ApertisFeedForward(config)

    def forward(self, hidden_s, att_mask=None, pos_ids=None, past_kv=None, output_att=False, use_c=False, defer=False):
        """defer=True (the model's layer loop): the layer accepts and returns a _Pending boundary, so every
        residual add of the stack is fused with the pre-norm that follows it (ops.dropout_add_layer_norm)."""
        x, att_w, cache = self.attention(hidden_s, att_mask, pos_ids, past_kv, output_att, use_c, defer=True)
        x, lb, rz = self.feed_forward(x, defer=defer)
        return x, att_w, cache, lb, rz


# ----------------------------------------------------------------------------------------------
# ApertisModel / ApertisForCausalLM  (reference core.py:1020-1648)
# ----------------------------------------------------------------------------------------------
class ApertisModel(nn.Module):
    def __init__(self, config: ApertisConfig):
        super().__init__()
        self.config = config
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.token_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.abs_pos_embeddings = None
        if config.position_embedding_type == "absolute":
            self.abs_pos_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.multimodal_encoder = None
        self.vision_projection = nn.Identity()
        if config.multimodal:
            self.multimodal_encoder = UnifiedMultimodalEncoder(config)
            if config.vision_embed_dim != config.hidden_size:
                self.vision_projection = nn.Linear(config.vision_embed_dim, config.hidden_size)
        self.layers = nn.ModuleList([ApertisLayer(config) for _ in range(config.num_hidden_layers)])
        norm = RMSNorm if config.use_rmsnorm else HipLayerNorm
        self.final_post_norm = norm(config.hidden_size, eps=config.layer_norm_eps)
        self.embed_dropout = nn.Dropout(config.hidden_dropout_prob)
        self.apply(self._init_weights)
        self.gradient_checkpointing = False

    def _init_weights(self, module):
        """normal(0, initializer_range) for Linear/Embedding, zero biases, unit norms; dt bias
        re-drawn in [ln 1e-3, ln 1e-2] (reference core.py:1045-1062)."""
        std = self.config.initializer_range
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(0.0, std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(0.0, std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.LayerNorm):
            if module.bias is not None:
                module.bias.data.zero_()
            if module.weight is not None:
                module.weight.data.fill_(1.0)
        elif isinstance(module, RMSNorm):
            module.scale.data.fill_(1.0)
        elif isinstance(module, AdaptiveExpertSystem) and module.router is not None:
            module.init_expert_weights(std)
        if isinstance(module, SelectiveLinearAttention):
            nn.init.uniform_(module.dt_proj_head.bias, a=math.log(1e-3), b=math.log(1e-2))

    def gradient_checkpointing_enable(self):
        self.gradient_checkpointing = True

    def get_input_embeddings(self):
        return self.token_embeddings

    def set_input_embeddings(self, embs):
        self.token_embeddings = embs

    def resize_token_embeddings(self, new_num_tokens: int) -> nn.Embedding:
        old = self.token_embeddings
        new = nn.Embedding(new_num_tokens, self.config.hidden_size, self.padding_idx, device=old.weight.device,
                           dtype=old.weight.dtype)
        self._init_weights(new)
        n = min(old.num_embeddings, new_num_tokens)
        new.weight.data[:n] = old.weight.data[:n]
        self.token_embeddings = new
        self.config.vocab_size = self.vocab_size = new_num_tokens
        if self.padding_idx is not None and self.padding_idx >= new_num_tokens:
            self.padding_idx = 0
            self.token_embeddings.padding_idx = 0
        return self.token_embeddings

    def _prepare_decoder_attention_mask(self, attention_mask, input_shape, inputs_embeds, past_len):
        """Additive causal+padding mask for the stock-attention fallback, or None when nothing is
        padded (reference core.py:1088-1139).  Ignored by the selective-SSM path."""
        if attention_mask is None or bool(torch.all(attention_mask.bool())):
            return None
        B, Lq = input_shape
        allow = None
        if Lq > 1:
            Lk = past_len + Lq
            causal = torch.ones(Lk, Lk, dtype=torch.bool, device=inputs_embeds.device).tril()[past_len:]
            allow = causal[None, None] & attention_mask[:, None, None, :].bool()
        elif past_len > 0:
            allow = attention_mask[:, None, None, :].bool()
        if allow is None:
            return None
        return (1.0 - allow.to(inputs_embeds.dtype)) * torch.finfo(inputs_embeds.dtype).min

    @_on_input_device
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, pixel_values=None, use_cache=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        cfg = self.config
        use_c = cfg.use_cache if use_cache is None else use_cache
        out_att = cfg.output_attentions if output_attentions is None else output_attentions
        out_hs = cfg.output_hidden_states if output_hidden_states is None else output_hidden_states
        if input_ids is not None and inputs_embeds is not None:
            raise ValueError("Specify one of input_ids or inputs_embeds.")
        if inputs_embeds is None:
            if input_ids is None:
                raise ValueError("input_ids required if inputs_embeds is None.")
            inputs_embeds = self.token_embeddings(input_ids)
        B, Lq = inputs_embeds.shape[:2]
        ssm = cfg.attention_type != "standard_mha"
        past_len = 0
        if past_key_values is not None and past_key_values[0] is not None and not ssm:
            past_len = past_key_values[0][0].shape[1]
        pos = position_ids
        if pos is None:
            pos = torch.arange(past_len, past_len + Lq, device=inputs_embeds.device).unsqueeze(0).expand(B, -1)
        if cfg.position_embedding_type == "absolute" and self.abs_pos_embeddings is not None:
            inputs_embeds = inputs_embeds + self.abs_pos_embeddings(pos)
        x, pos_layers = inputs_embeds, pos
        if cfg.multimodal and pixel_values is not None and past_len == 0:              # core.py:1207-1227
            img = self.vision_projection_forward(self.multimodal_encoder(pixel_values))
            n_img = img.shape[1]
            x = torch.cat([img.to(inputs_embeds.dtype), inputs_embeds], dim=1)
            img_pos = torch.arange(n_img, device=x.device).unsqueeze(0).expand(B, -1)
            pos_layers = torch.cat([img_pos, pos + n_img], dim=1)
            if attention_mask is not None and attention_mask.shape[1] == Lq:
                attention_mask = torch.cat([attention_mask.new_ones(B, n_img), attention_mask], dim=1)
            elif attention_mask is None:
                attention_mask = torch.ones(B, n_img + Lq, dtype=torch.long, device=x.device)
        elif cfg.multimodal and pixel_values is not None:
            logger.warning("pixel_values provided with past_key_values: image ignored for this step")
        x = self.embed_dropout(x)
        mask = None if ssm else self._prepare_decoder_attention_mask(attention_mask, (B, x.shape[1]), x, past_len)

        all_hs, all_att, all_cache = [], [], []
        lbs, rzs = [], []
        pre_all = None
        if isinstance(past_key_values, _StackedPast) and x.shape[1] == 1 and use_c and not out_att and not torch.is_grad_enabled():
            pre_all = self._decode_prepass(past_key_values)
        try:
            for i, layer in enumerate(self.layers):
                if out_hs:
                    all_hs.append(x)
                past = past_key_values[i] if past_key_values and i < len(past_key_values) else None
                if pre_all is not None:
                    layer.attention.attention_mechanism_impl._decode_pre = pre_all[i]
                if self.gradient_checkpointing and self.training and not use_c:             # core.py:1258
                    x, att_w, cache, lb, rz = torch.utils.checkpoint.checkpoint(layer, x, mask, pos_layers, past, out_att,
                                                                                use_c, use_reentrant=False)
                else:
                    # hidden states are only materialised at layer boundaries when somebody asked for them
                    x, att_w, cache, lb, rz = layer(x, mask, pos_layers, past, out_att, use_c, defer=not out_hs)
                if out_att:
                    all_att.append(att_w)
                if use_c:
                    all_cache.append(cache)
                if cfg.use_expert_system:
                    lbs.append(lb)
                    rzs.append(rz)
        finally:
            #  the pre-pass has advanced every layer's SSM state in place; its per-layer results travel through the
            # module attribute _decode_pre.  A layer that raised or left its single-token path must not leave a stale tensor
            # behind for the next, unrelated forward on that module: whatever happened above, nothing stays pending.
            if pre_all is not None:
                for layer in self.layers:
                    layer.attention.attention_mechanism_impl._decode_pre = None
        x, _ = _enter_block(self.final_post_norm, x)
        if out_hs:
            all_hs.append(x)
        if cfg.use_expert_system:
            # one reduction over the layers instead of two scalar adds per layer (core.py:1283-1287 sums as it goes)
            lb_tot = torch.stack(lbs).sum() if lbs else _zero_scalar(x.device, x.dtype)
            rz_tot = torch.stack(rzs).sum() if rzs else _zero_scalar(x.device, x.dtype)
        return (x, tuple(all_hs) if out_hs and all_hs else None, tuple(all_att) if out_att and all_att else None,
                tuple(all_cache) if use_c and all_cache else None,
                lb_tot if cfg.use_expert_system else None, rz_tot if cfg.use_expert_system else None)

    def _decode_prepass(self, st):
        """The cache-only half of a single-token step for ALL layers in three launches (csrc/decode_step.hip): the reference keeps
        the FIRST conv output of [cached window | new xp] (core.py:369-373), which sees window[0] only, so conv output,
        x_param_proj, dt projection and state update of a token step do not depend on the token.  Returns pre [NL,B,Dn] fp32 =
        C s + D xc per layer (the states in `st.state_all` are updated in place), or None when the stack is not uniform."""
        impls = [l.attention.attention_mechanism_impl for l in self.layers]
        m0 = impls[0]
        NL = len(impls)
        if (NL == 0 or any(not isinstance(m, SelectiveLinearAttention) for m in impls) or st.conv_all.shape[0] != NL
                or not st.conv_all.is_cuda or m0.conv_kernel_size < 2 or m0.conv_kernel_size > 16
                or st.conv_all.dtype not in (torch.float32, torch.bfloat16)):
            return None
        Dn, R, h, N = m0.d_inner, m0.dt_rank, m0.num_heads, m0.d_state
        if any((m.d_inner, m.dt_rank, m.num_heads, m.d_state, m.conv_kernel_size) != (Dn, R, h, N, m0.conv_kernel_size) for m in impls):
            return None
        # (the layers' decode branch must take what is computed here - the states are updated in place, once: the caches have to be
        #  exactly what that branch accepts as a window)
        if (tuple(st.conv_all.shape[2:]) != (Dn, m0.conv_kernel_size - 1) or tuple(st.state_all.shape[1:]) != (st.conv_all.shape[1], Dn)
                or st.state_all.dtype != torch.float32 or not (1 <= R <= 64 and 1 <= h <= 16)):    # (ops.tiny_linear_supported: dt inside the state kernel)
            return None
        B = st.conv_all.shape[1]
        if any((m.dt_proj_head.bias is None) != (m0.dt_proj_head.bias is None) or m.conv1d.bias is None for m in impls):
            return None
        srcs = tuple(t for m in impls for t in (m.conv1d.weight, m.conv1d.bias, m.x_param_proj.weight, m.dt_proj_head.weight,
                                               m.dt_proj_head.bias, m.A_log, m.D) if t is not None)

        def make():
            pads = [m._padded_param_weight() for m in impls]
            return (torch.stack([m.conv1d.weight.detach().float().reshape(Dn, -1) for m in impls]).contiguous(),
                    torch.stack([m.conv1d.bias.detach().float() for m in impls]).contiguous(),
                    torch.stack([p[0].detach().float() for p in pads]).contiguous(),
                    torch.stack([m.dt_proj_head.weight.detach().float() for m in impls]).contiguous(),
                    (None if impls[0].dt_proj_head.bias is None else
                     torch.stack([m.dt_proj_head.bias.detach().float() for m in impls]).contiguous()),
                    torch.stack([m.A_log.detach().float() for m in impls]).contiguous(),
                    torch.stack([m.D.detach().float() for m in impls]).contiguous(), pads[0][1], pads[0][2])

        conv_w, conv_b, wp, w_dt, b_dt, a_log, d_all, Wb, Wr = ops.cached_prep(("decode_stack", id(self)), srcs, make)
        xc_all = ops.decode_pre_conv(st.conv_all, conv_w, conv_b)
        offs = st.offsets(NL, B)
        p_all = ops.grouped_linear(xc_all.reshape(NL * B, Dn), wp, None, offs, NL * B, compute_dtype=xc_all.dtype)
        return ops.decode_pre_state(p_all, 0, Wb, 2 * Wb, w_dt, b_dt, a_log, d_all, xc_all, st.state_all)

    def vision_projection_forward(self, feats):
        """Linear(vision_embed_dim -> hidden) on the MFMA GEMM tile (core.py:1035,1209)."""
        if isinstance(self.vision_projection, nn.Identity):
            return feats
        return ops.linear_mfma(feats, self.vision_projection.weight, self.vision_projection.bias,
                               compute_dtype=_compute_dtype(feats))


class ApertisForCausalLM(nn.Module):
    def __init__(self, config: ApertisConfig):
        super().__init__()
        self.config = config
        self.model = ApertisModel(config)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        if config.tie_word_embeddings:
            self.lm_head.weight = self.model.token_embeddings.weight
        else:
            self.lm_head.weight.data.normal_(0.0, config.initializer_range)
        # training-loop opt-in: loss straight from the hidden states, logits never materialised (forward returns
        # None in their slot).  Off by default: the reference's callers may read outputs[1].
        self.fused_lm_head_loss = False

    def load_state_dict(self, state_dict, strict=True):
        return super().load_state_dict(state_dict, strict=strict)

    def save_pretrained(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        torch.save(self.state_dict(), os.path.join(save_directory, "pytorch_model.bin"))
        self.config.save_pretrained(save_directory)

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_lm_head):
        self.lm_head = new_lm_head

    def get_input_embeddings(self):
        return self.model.token_embeddings

    def gradient_checkpointing_enable(self):
        self.model.gradient_checkpointing_enable()

    def resize_token_embeddings(self, new_num_tokens: int) -> nn.Linear:
        self.model.resize_token_embeddings(new_num_tokens)
        if self.config.tie_word_embeddings and self.lm_head.weight.shape[0] == new_num_tokens:
            pass
        if self.config.tie_word_embeddings:
            self.lm_head.weight = self.model.token_embeddings.weight
            self.lm_head.out_features = new_num_tokens
        else:
            old = self.lm_head
            self.lm_head = nn.Linear(self.config.hidden_size, new_num_tokens, bias=False, device=old.weight.device,
                                     dtype=old.weight.dtype)
            self.lm_head.weight.data.normal_(0.0, self.config.initializer_range)
            n = min(old.out_features, new_num_tokens)
            self.lm_head.weight.data[:n] = old.weight.data[:n]
        self.config.vocab_size = self.lm_head.out_features
        return self.lm_head

    @_on_input_device
    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, pixel_values=None, labels=None, use_cache=None, output_attentions=None,
                output_hidden_states=None):
        """Returns the reference's 7-tuple (loss, logits, hidden_states, attentions,
        past_key_values, lb_loss, rz_loss)  (core.py:1361-1472)."""
        outs = self.model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                          past_key_values=past_key_values, inputs_embeds=inputs_embeds, pixel_values=pixel_values,
                          use_cache=use_cache, output_attentions=output_attentions,
                          output_hidden_states=output_hidden_states)
        hs = outs[0]
        if self.config.multimodal and pixel_values is not None and past_key_values is None and input_ids is not None:
            start = hs.shape[1] - input_ids.shape[1]                     # text positions are last, core.py:1399-1406
            if start >= 0:
                hs = hs[:, start:]
        if (labels is not None and self.fused_lm_head_loss and self.training and
                ops.linear_cross_entropy_supported(hs, self.lm_head.weight, labels)):
            # LM head + shifted cross entropy one sequence at a time (core.py:1412-1450): no [B, L, V] logits tensor, the
            # tuple's logits slot is None.  Opt-in (TrainStep / ApertisTrainer set it: they only read the loss).
            loss = ops.linear_cross_entropy(hs, self.lm_head.weight, labels, ignore_index=-100, compute_dtype=_compute_dtype(hs))
            if outs[4] is not None:
                loss = loss + outs[4]
            if outs[5] is not None:
                loss = loss + outs[5]
            return (loss, None) + outs[1:]
        logits = self.lm_head(hs)
        loss = None
        if labels is not None:
            sl, tl = logits[..., :-1, :], labels[..., 1:]
            n = min(sl.shape[1], tl.shape[1])
            if n == 0:
                loss = torch.zeros((), device=logits.device, requires_grad=self.training)
            elif ops.shifted_cross_entropy_supported(logits, labels):
                loss = ops.shifted_cross_entropy(logits, labels, ignore_index=-100)     # core.py:1407-1416, one pass
            else:
                loss = F.cross_entropy(sl[:, :n].reshape(-1, sl.shape[-1]).float(), tl[:, :n].reshape(-1),
                                       ignore_index=-100)
            if outs[4] is not None:
                loss = loss + outs[4]
            if outs[5] is not None:
                loss = loss + outs[5]
        return (loss, logits) + outs[1:]

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None,
                                      position_ids_override=None, **kwargs):
        B, L = input_ids.shape
        if past_key_values is not None:
            ids = input_ids[:, -1:]
            pos = torch.full((B, 1), attention_mask.shape[1] - 1, dtype=torch.long, device=input_ids.device)
        else:
            ids = input_ids
            pos = torch.arange(L, device=input_ids.device).unsqueeze(0).expand(B, -1)
        if position_ids_override is not None:
            pos = position_ids_override
        inputs = {"input_ids": ids, "past_key_values": past_key_values, "attention_mask": attention_mask,
                  "position_ids": pos, "use_cache": kwargs.get("use_cache", True)}
        if past_key_values is None and "pixel_values" in kwargs:
            inputs["pixel_values"] = kwargs["pixel_values"]
        return inputs

    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, position_ids=None, pixel_values=None,
                 max_new_tokens: Optional[int] = 20, min_new_tokens: Optional[int] = 0, do_sample: Optional[bool] = False,
                 temperature: Optional[float] = 1.0, top_k: Optional[int] = 50, top_p: Optional[float] = 1.0,
                 repetition_penalty: Optional[float] = 1.0, eos_token_id=None, pad_token_id=None,
                 use_cache: bool = True, **kwargs):
        """Greedy / top-k / top-p / repetition-penalty decoding loop (reference core.py:1520-1644).  The prepared copies of
        the weights (stacked / padded / cast) are reused from token to token inside the call (ops.prep_cache_scope)."""
        if input_ids is None:
            raise ValueError("input_ids must be provided.")
        with ops.prep_cache_scope():
            return self._generate(input_ids, attention_mask, position_ids, pixel_values, max_new_tokens, min_new_tokens,
                                  do_sample, temperature, top_k, top_p, repetition_penalty, eos_token_id, pad_token_id,
                                  use_cache)

    def _generate(self, input_ids, attention_mask, position_ids, pixel_values, max_new_tokens, min_new_tokens, do_sample,
                  temperature, top_k, top_p, repetition_penalty, eos_token_id, pad_token_id, use_cache):
        B, prompt_len = input_ids.shape
        temp = max(temperature, 1e-6) if do_sample else 1.0
        eos = self.config.eos_token_id if eos_token_id is None else eos_token_id
        eos = [] if eos is None else (eos if isinstance(eos, list) else [eos])
        pad = pad_token_id if pad_token_id is not None else self.config.pad_token_id
        pad = 0 if pad is None else pad
        mask = torch.ones_like(input_ids) if attention_mask is None else attention_mask
        pos = position_ids
        if pos is None:
            pos = torch.arange(prompt_len, device=input_ids.device).unsqueeze(0).expand(B, -1)
        px = pixel_values
        if self.config.multimodal and px is not None:
            n_img = (self.config.image_size // self.config.vision_patch_size) ** 2 + 1
            mask = torch.cat([mask.new_ones(B, n_img), mask], dim=1)
            pos = torch.cat([torch.arange(n_img, device=input_ids.device).unsqueeze(0).expand(B, -1), pos + n_img], dim=1)
        tokens, past = input_ids, None
        alive = torch.ones(B, dtype=torch.long, device=input_ids.device)
        for _ in range(max_new_tokens):
            inp = self.prepare_inputs_for_generation(tokens if past is None else tokens[:, -1:], past_key_values=past,
                                                     attention_mask=mask,
                                                     position_ids_override=pos if past is None else None,
                                                     pixel_values=px, use_cache=use_cache)
            px = None
            out = self(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"],
                       position_ids=inp["position_ids"], past_key_values=inp["past_key_values"],
                       pixel_values=inp.get("pixel_values"), use_cache=inp["use_cache"])
            nxt_logits = out[1][:, -1, :].float()
            past = out[4] if use_cache else None
            if repetition_penalty != 1.0:
                for b in range(B):
                    if alive[b]:
                        seen = tokens[b][tokens[b] < nxt_logits.shape[-1]]
                        for t_ in seen.tolist():                    # divides once per occurrence, like the reference
                            nxt_logits[b, t_] /= repetition_penalty
            if do_sample:
                if temp != 1.0:
                    nxt_logits = nxt_logits / temp
                if top_k > 0:
                    kth = torch.topk(nxt_logits, top_k).values[:, -1:]
                    nxt_logits = nxt_logits.masked_fill(nxt_logits < kth, float("-inf"))
                if top_p < 1.0:
                    srt, order = torch.sort(nxt_logits, descending=True)
                    drop = torch.cumsum(F.softmax(srt, dim=-1), dim=-1) > top_p
                    drop[..., 1:] = drop[..., :-1].clone()
                    drop[..., 0] = False
                    nxt_logits = nxt_logits.masked_fill(torch.zeros_like(drop).scatter_(-1, order, drop), float("-inf"))
                nxt = torch.multinomial(F.softmax(nxt_logits, dim=-1), 1).squeeze(1)
            else:
                nxt = torch.argmax(nxt_logits, dim=-1)
            nxt = nxt * alive + pad * (1 - alive)
            tokens = torch.cat([tokens, nxt.unsqueeze(-1)], dim=-1)
            mask = torch.cat([mask, alive.unsqueeze(-1).to(mask.dtype)], dim=1)
            for e_ in eos:
                if e_ is not None:
                    alive = alive.masked_fill((nxt == e_) & (alive == 1), 0)
            if alive.max() == 0 and tokens.shape[1] - prompt_len >= min_new_tokens:
                break
            left = max_new_tokens - (tokens.shape[1] - prompt_len)
            if past is not None and left >= DECODE_GRAPH_MIN_STEPS and self._decode_graph_ok(tokens, do_sample, repetition_penalty):
                # the remaining single-token steps as ONE captured HIP graph replayed `left` times (same kernels, same tokens)
                return self._generate_graph_tail(tokens, past, alive, left, prompt_len, min_new_tokens, eos, pad)
        return tokens

    def _decode_graph_ok(self, tokens, do_sample, repetition_penalty):
        cfg = self.config
        return (DECODE_GRAPH and tokens.is_cuda and not do_sample and repetition_penalty == 1.0 and not torch.is_grad_enabled()
                and cfg.attention_type != "standard_mha" and cfg.position_embedding_type != "absolute" and not self.training)

    def _generate_graph_tail(self, tokens, past, alive, left, prompt_len, min_new_tokens, eos, pad):
        """Greedy decoding of `left` more tokens through a captured HIP graph of the single-token step (reference
        core.py:1578-1644: the same forward through the cache, argmax, eos / pad bookkeeping - attention mask and position ids
        do not enter an SSM model's step).  An eager token step is ~2 600 small launches, 18 ms of mostly host time at 44
        layers; the replay is 10 ms (tools/decode_graph_try.py).  Token, cache, alive flags, the step counter and the outputs
        live in static buffers that the graph updates in place; the host looks at the alive flags every 16 steps only."""
        B, dev = tokens.shape[0], tokens.device
        s_tok = tokens[:, -1:].clone()
        # (contiguous copies: the prefill hands the conv window over as a transposed view, and a cache that is not contiguous
        #  is copied in and out of every token step instead of being updated in place - two launches per layer)
        s_past = [(c.clone(memory_format=torch.contiguous_format), st.clone(memory_format=torch.contiguous_format)) for (c, st) in past]
        if (DECODE_PREPASS and len(s_past) > 0 and all(c.dim() == 3 and st.dim() == 3 and st.dtype == torch.float32 and
                                                       c.shape == s_past[0][0].shape and st.shape == s_past[0][1].shape
                                                       and c.dtype == s_past[0][0].dtype for c, st in s_past)):
            # one tensor per kind, the per-layer caches as views: the cache-only half of every layer's step runs at once
            conv_all = torch.stack([c for c, _ in s_past]).contiguous()
            state_all = torch.stack([st.reshape(st.shape[0], -1) for _, st in s_past]).contiguous()
            s_past = _StackedPast(conv_all, state_all, s_past[0][1].shape[1], s_past[0][1].shape[2])
        s_alive = alive.clone()
        s_idx = torch.zeros(1, dtype=torch.long, device=dev)
        s_out = torch.full((B, left), pad, dtype=tokens.dtype, device=dev)
        s_any = torch.ones(left, dtype=alive.dtype, device=dev)          # max over the batch of `alive` after each step

        def body():
            out = self(input_ids=s_tok, past_key_values=s_past, use_cache=True)
            nxt = torch.argmax(out[1][:, -1, :].float(), dim=-1)
            nxt = nxt * s_alive + pad * (1 - s_alive)
            s_out.scatter_(1, s_idx.expand(B, 1), nxt.unsqueeze(1).to(s_out.dtype))
            al = s_alive
            for e_ in eos:
                if e_ is not None:
                    al = al.masked_fill((nxt == e_) & (al == 1), 0)
            s_alive.copy_(al)
            s_any.scatter_(0, s_idx, al.max().reshape(1))
            s_idx.add_(1)
            s_tok.copy_(nxt.unsqueeze(1))
            for (sc, ss), (nc, ns) in zip(s_past, out[4]):
                if nc.data_ptr() != sc.data_ptr():     # (both updated in place by the step: _inplace_cache)
                    sc.copy_(nc)
                if ns.data_ptr() != ss.data_ptr():
                    ss.copy_(ns)

        ssm_blocks = [m for m in self.modules() if isinstance(m, SelectiveLinearAttention)]
        for m in ssm_blocks:
            m._inplace_cache = True
        try:
            return self._generate_graph_run(body, tokens, s_tok, s_past, s_alive, s_idx, s_out, s_any, left, prompt_len, min_new_tokens, pad)
        finally:
            for m in ssm_blocks:
                m._inplace_cache = False

    def _generate_graph_run(self, body, tokens, s_tok, s_past, s_alive, s_idx, s_out, s_any, left, prompt_len, min_new_tokens, pad):
        dev = tokens.device
        # warm-up on a side stream (lazy bindings, prepared-weight cache, allocator), then restore the state it advanced
        keep = (s_tok.clone(), [(c.clone(), st.clone()) for (c, st) in s_past], s_alive.clone())
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(2):
                body()
        torch.cuda.current_stream(dev).wait_stream(side)

        def restore():
            s_tok.copy_(keep[0])
            for (sc, ss), (c, st) in zip(s_past, keep[1]):
                sc.copy_(c)
                ss.copy_(st)
            s_alive.copy_(keep[2])
            s_idx.zero_()
            s_out.fill_(pad)
            s_any.fill_(1)

        restore()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):    # (another thread's GPU work - a trainer thread beside chat - must not fail on this capture)
            body()
        restore()
        done = left
        for i in range(left):
            graph.replay()
            if (i + 1) % 16 == 0 or i + 1 == left:
                flags = s_any[:i + 1].tolist()                         # the only host sync: every 16 steps
                stop = next((j for j, a in enumerate(flags)
                             if a == 0 and tokens.shape[1] - prompt_len + j + 1 >= min_new_tokens), None)
                if stop is not None:
                    done = stop + 1
                    break
        return torch.cat([tokens, s_out[:, :done]], dim=-1)


# ----------------------------------------------------------------------------------------------
# Model sizing  (reference core.py:1709-2105)
# ----------------------------------------------------------------------------------------------
def parse_param_count(param_str: Union[str, int]) -> int:
    """'10M' / '1.5B' / '350k' / int -> integer parameter count (core.py:1709-1739)."""
    if isinstance(param_str, int):
        return param_str
    s = str(param_str).strip().upper()
    if not s:
        raise ValueError("Parameter string cannot be empty.")
    mult = {"K": 1_000, "M": 1_000_000, "B": 1_000_000_000}.get(s[-1], 1)
    if mult != 1:
        s = s[:-1]
    try:
        return int(float(s) * mult)
    except ValueError:
        raise ValueError(f"Invalid numeric value in parameter string: '{param_str}'")


def _calculate_params_for_dims(vocab_size, hidden_size, num_layers, intermediate_size, tie_word_embeddings=True,
                               use_expert_system=False, num_experts=0) -> int:
    """The reference's MHA-shaped estimator (4h^2 attention), used for sizing even for SSM models
    (core.py:1741-1769)."""
    h, I = hidden_size, intermediate_size
    p = vocab_size * h * (1 if tie_word_embeddings else 2)
    p += num_layers * 4 * h * h
    if use_expert_system and num_experts > 0:
        p += num_layers * (num_experts * 2 * h * I + h * num_experts)
    else:
        p += num_layers * 2 * h * I
    return p + (2 * num_layers + 1) * 2 * h


def calculate_model_dimensions(target_params_str, vocab_size, use_expert_system=False, num_experts_target=8,
                               min_hidden_size=256, max_hidden_size=8192, min_layers=2, max_layers=128,
                               head_dim_preference=64, intermediate_multiple_of=256, intermediate_ratio=4.0,
                               tie_word_embeddings=True) -> Dict[str, Any]:
    """Grid search (layers in steps of 2, hidden a multiple of the head dim with a growing step,
    I = 4h rounded up to 256) for the closest estimated parameter count; reproduces the
    reference's choices exactly (core.py:1771-1893; golden table in tests/golden)."""
    target = parse_param_count(target_params_str)
    hd = head_dim_preference
    best, best_diff = None, float("inf")

    def dims_for(h):
        heads = max(1, h // hd)
        I = max(intermediate_multiple_of, -(-int(h * intermediate_ratio) // intermediate_multiple_of) * intermediate_multiple_of)
        return heads, I

    for layers in range(min_layers, max_layers + 1, 2):
        cur = min_hidden_size
        while cur <= max_hidden_size:
            h = cur if cur % hd == 0 else (cur // hd + 1) * hd
            h = h or hd
            if h > max_hidden_size:
                break
            heads, I = dims_for(h)
            params = _calculate_params_for_dims(vocab_size, h, layers, I, tie_word_embeddings, use_expert_system,
                                                num_experts_target if use_expert_system else 0)
            diff = abs(params - target)
            if diff < best_diff:
                best_diff = diff
                best = {"hidden_size": h, "num_hidden_layers": layers, "num_attention_heads": heads,
                        "intermediate_size": I, "calculated_params": params, "target_params": target,
                        "param_diff": diff}
            if params > target and diff > best_diff:
                break
            cur += max(hd, h // 16)
            if cur > max_hidden_size and best is None:
                cur = max_hidden_size
    if best is None:
        h = min_hidden_size
        heads, I = dims_for(h)
        params = _calculate_params_for_dims(vocab_size, h, min_layers, I, tie_word_embeddings, use_expert_system,
                                            num_experts_target if use_expert_system else 0)
        return {"hidden_size": h, "num_hidden_layers": min_layers, "num_attention_heads": heads,
                "intermediate_size": I, "calculated_params": params, "target_params": target,
                "param_diff": abs(params - target), "備考": "Fallback configuration"}
    return best


def estimate_model_parameters(config: ApertisConfig) -> int:
    """Parameter estimate from a config (core.py:1895-1965): embeddings (+ untied head), 4h^2
    attention, dense or E-expert FFN (+ router), 2L+1 norms, absolute positions, vision projection."""
    h, I, L = config.hidden_size, config.intermediate_size, config.num_hidden_layers
    total = config.vocab_size * h * (1 if config.tie_word_embeddings else 2)
    ffn = 2 * h * I
    if config.use_expert_system and config.num_experts > 0:
        ffn = config.num_experts * 2 * h * I + h * config.num_experts
    total += L * (4 * h * h + ffn) + (2 * L + 1) * 2 * h
    if config.position_embedding_type == "absolute":
        total += config.max_position_embeddings * h
    if config.multimodal and config.vision_embed_dim != h:
        total += config.vision_embed_dim * h
    return total


def create_apertis_model(target_param_count: Union[str, int] = "125M", vocab_size_override: Optional[int] = None,
                         multimodal: bool = False, use_flash_attention: bool = False, use_expert_system: bool = False,
                         num_experts_target_override: Optional[int] = None,
                         experts_per_token_target_override: Optional[int] = None,
                         attention_type_override: Optional[str] = None, ssm_d_inner: Optional[int] = None,
                         ssm_d_state: int = 16, ssm_dt_rank: Union[int, str] = "auto", ssm_conv_kernel: int = 4,
                         config_overrides: Optional[Dict[str, Any]] = None) -> ApertisForCausalLM:
    """Size a model for a parameter budget and build it (reference core.py:1969-2105)."""
    vocab = vocab_size_override if vocab_size_override is not None else ApertisConfig(**(config_overrides or {})).vocab_size
    dims = calculate_model_dimensions(target_param_count, vocab, use_expert_system=use_expert_system,
                                      num_experts_target=8 if num_experts_target_override is None else num_experts_target_override)
    cfg = {k: dims[k] for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size")}
    cfg["vocab_size"] = vocab
    cfg["attention_type"] = attention_type_override or "standard_mha"
    cfg.update(multimodal=multimodal, use_flash_attention=use_flash_attention, use_expert_system=use_expert_system,
               ssm_d_inner=ssm_d_inner, ssm_d_state=ssm_d_state, ssm_dt_rank=ssm_dt_rank, ssm_conv_kernel=ssm_conv_kernel)
    if use_expert_system:
        cfg["num_experts"] = 8 if num_experts_target_override is None else num_experts_target_override
        cfg["experts_per_token"] = 2 if experts_per_token_target_override is None else experts_per_token_target_override
    if config_overrides:
        cfg.update(config_overrides)
    h, heads = cfg["hidden_size"], cfg["num_attention_heads"]
    if h % heads != 0:
        head_dim = h // heads if heads > 0 else 64
        if head_dim > 0 and h % head_dim == 0:
            cfg["num_attention_heads"] = h // head_dim
        else:
            cfg["num_attention_heads"] = next((i for i in range(min(heads, h), 0, -1) if h % i == 0), 1)
    config = ApertisConfig(**cfg)
    logger.info("Model configured with H=%d L=%d A=%d I=%d V=%d (~%.2fM params by the reference estimator)",
                config.hidden_size, config.num_hidden_layers, config.num_attention_heads, config.intermediate_size,
                config.vocab_size, estimate_model_parameters(config) / 1e6)
    return ApertisForCausalLM(config)
