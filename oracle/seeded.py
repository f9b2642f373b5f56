"""Seeded weights for size-scaled fixtures — ORACLE / test infrastructure only.

A 125M or 1.5B-shaped state dict cannot be committed (400 MB+), so the big-shape fixtures store only the inputs and
the reference's scalar / sampled outputs, and both sides (tools/gen_golden.py running the reference, the tests running
the HIP path or the CPU oracle) rebuild the SAME weights from this rule: every tensor is drawn from its own generator,
seeded by the CRC of its state-dict key, so the values do not depend on module construction order or on how much RNG a
constructor consumed.
"""
import math
import zlib

import torch


def seeded_tensor(key, shape, dtype=torch.float32, gain=1.0):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    r = lambda: torch.randn(shape, generator=g, dtype=torch.float32)
    u = lambda lo, hi: torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo
    if key.endswith("A_log"):                                   # core.py:317
        t = u(math.log(0.5), math.log(0.99))
    elif key.endswith("dt_proj_head.bias"):                     # core.py:315
        t = u(math.log(1e-3), math.log(1e-2))
    elif key.endswith(".D") or key == "D":
        t = 1.0 + 0.1 * r()
    elif key.endswith("conv1d.weight"):
        t = 0.3 * r()
    elif key.endswith("cls_token") or key.endswith("vision_pos_embed"):
        t = 0.02 * r()
    elif len(shape) == 1 and key.endswith("weight"):            # LayerNorm / RMSNorm scale
        t = 1.0 + 0.1 * r()
    elif len(shape) <= 1:                                       # biases, w_noise
        t = 0.02 * r()
    else:
        t = (0.02 * gain) * r()
    return t.to(dtype)


def fill_state_dict(sd, gain=1.0):
    """New state dict with the keys / shapes / dtypes of `sd` and seeded values.  Tied tensors (lm_head.weight ==
    token_embeddings.weight) stay tied because the caller loads both keys from the same entry."""
    out = {}
    for k in sorted(sd):
        v = sd[k]
        if not torch.is_floating_point(v):
            out[k] = v.clone()
            continue
        src = "model.token_embeddings.weight" if k == "lm_head.weight" and "model.token_embeddings.weight" in sd and \
            tuple(sd["model.token_embeddings.weight"].shape) == tuple(v.shape) else k
        out[k] = seeded_tensor(src, v.shape, v.dtype, gain)
    return out
