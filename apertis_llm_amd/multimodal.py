"""UnifiedMultimodalEncoder mirror (reference src/multimodal/module.py:10-161).

Patch embedding runs as a GEMM on the MFMA tile: stride == kernel, so Conv2d(3, Dv, p, p) is
`patches[B*gh*gw, 3*p*p] @ weight.view(Dv, -1).T + bias` with K index c*p*p + ky*p + kx
(module.py:35-40,102-103).  The 12-layer ViT body stays stock nn.TransformerEncoderLayer on
ROCm (not a kernel target, ~98.7 % of the encoder FLOPs).
"""
import threading
from typing import List

import torch
import torch.nn as nn

from . import ops


_fp_lock = threading.Lock()
_fp_depth = 0
_fp_saved = True


def _fastpath_enter():
    global _fp_depth, _fp_saved
    with _fp_lock:
        if _fp_depth == 0:
            _fp_saved = torch.backends.mha.get_fastpath_enabled()
            torch.backends.mha.set_fastpath_enabled(False)
        _fp_depth += 1


def _fastpath_exit():
    global _fp_depth
    with _fp_lock:
        _fp_depth -= 1
        if _fp_depth == 0:
            torch.backends.mha.set_fastpath_enabled(_fp_saved)


class UnifiedMultimodalEncoder(nn.Module):
    """Vision tower: patch embed -> [cls] + learned positions -> pre-norm transformer -> LayerNorm."""

    _MEAN = (0.485, 0.456, 0.406)
    _STD = (0.229, 0.224, 0.225)

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.image_size = config.image_size
        self.vision_embed_dim = config.vision_embed_dim
        self.vision_patch_size = config.vision_patch_size
        # nn.Conv2d only holds the parameters (checkpoint names/shapes); forward() is a GEMM
        self.patch_embed = nn.Conv2d(3, self.vision_embed_dim, kernel_size=self.vision_patch_size,
                                     stride=self.vision_patch_size)
        self.num_patches = (self.image_size // self.vision_patch_size) ** 2
        self.vision_pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, self.vision_embed_dim))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.vision_embed_dim))
        heads = getattr(config, "vision_heads", 12)
        layers = getattr(config, "vision_layers", 12)
        self.vision_layers = nn.ModuleList([
            nn.TransformerEncoderLayer(d_model=self.vision_embed_dim, nhead=heads,
                                       dim_feedforward=self.vision_embed_dim * 4, dropout=0.1, activation="gelu",
                                       batch_first=True, norm_first=True)
            for _ in range(layers)])
        self.vision_ln = nn.LayerNorm(self.vision_embed_dim)
        nn.init.normal_(self.vision_pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=0.02)
        nn.init.normal_(self.patch_embed.weight, std=0.02)
        nn.init.zeros_(self.patch_embed.bias)

    def patchify(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """[B,3,Hi,Wi] -> [B, gh*gw, 3*p*p]; row = py*gw + px, column = c*p*p + ky*p + kx."""
        B, C, Hi, Wi = pixel_values.shape
        p = self.vision_patch_size
        gh, gw = Hi // p, Wi // p
        x = pixel_values[:, :, :gh * p, :gw * p].reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(B, gh * gw, C * p * p)

    def embed_patches(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """Patch-embed stage of forward() (module.py:102-103) -> [B, num_patches, Dv]."""
        with ops.device_guard(pixel_values):
            return self._embed_patches(pixel_values)

    def _embed_patches(self, pixel_values: torch.Tensor) -> torch.Tensor:
        cols = self.patchify(pixel_values)
        cd = cols.dtype
        if torch.is_autocast_enabled():
            cd = torch.bfloat16
        w = self.patch_embed.weight.reshape(self.vision_embed_dim, -1)
        return ops.linear_mfma(cols, w, self.patch_embed.bias, compute_dtype=cd)

    def forward(self, pixel_values: torch.Tensor) -> torch.Tensor:
        B = pixel_values.shape[0]
        patches = self.embed_patches(pixel_values).to(self.vision_pos_embed.dtype)
        x = torch.cat([self.cls_token.expand(B, -1, -1), patches], dim=1) + self.vision_pos_embed   # module.py:106-110
        # torch's fused inference "fast path" for TransformerEncoderLayer loses ~1e-4 on ROCm (measured: 1.4e-4 vs 3e-7
        # abs error against fp64 for one layer); parity with the reference's fp32 CPU path needs the op-by-op path.
        # The switch is process-global and the reference runs training in daemon threads (interface.py:1188): it is
        # flipped under a lock that counts the forwards inside, restored when the last one leaves.
        _fastpath_enter()
        try:
            for layer in self.vision_layers:                                                          # module.py:113-114
                x = layer(x)
        finally:
            _fastpath_exit()
        return self.vision_ln(x)                                                                      # module.py:117

    # -- PIL helpers (module.py:121-161) without the torchvision dependency ----------------------
    def process_image(self, image_path: str) -> torch.Tensor:
        try:
            import numpy as np
            from PIL import Image
            img = Image.open(image_path).convert("RGB").resize((self.image_size, self.image_size), Image.BILINEAR)
            t = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
            mean = torch.tensor(self._MEAN).view(3, 1, 1)
            std = torch.tensor(self._STD).view(3, 1, 1)
            return ((t - mean) / std).unsqueeze(0)
        except Exception as e:  # same contract as the reference: blank image on failure
            print(f"Error processing image {image_path}: {e}")
            return torch.zeros(1, 3, self.image_size, self.image_size)

    def process_batch(self, image_paths: List[str]) -> torch.Tensor:
        return torch.stack([self.process_image(p).squeeze(0) for p in image_paths])
