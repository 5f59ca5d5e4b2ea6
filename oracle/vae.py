"""fp32 CPU restatement of diffusers' ``AutoencoderKL.decode`` for SD-1.5.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- parity unpinned.

Reference call site: ``image = self.vae.decode(latents / self.vae.config.scaling_factor)[0]``
(``src/models.py:287-302``) followed by ``postprocess("pt")`` = ``(x / 2 + 0.5).clamp(0, 1)``
(``:312``).  Architecture (diffusers 0.32.1): post_quant_conv 1x1 -> decoder.conv_in -> mid block
(ResNet, single-head attention with head dim 512 over all tokens, ResNet) -> 4 up blocks of 3 ResNets
(512, 512, 256, 128 channels; nearest-2x upsample + 3x3 conv after the first three) ->
GroupNorm(32, eps 1e-6) -> SiLU -> conv_out.  ResNets carry no time embedding.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VaeConfig:
    sample_size: int = 64                      # latent H = W
    in_channels: int = 4                       # latent channels
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


def _resnet(w, p, x, groups):
    h = F.silu(F.group_norm(x, groups, w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-6))
    h = F.conv2d(h, w[p + "conv1.weight"], w[p + "conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-6))
    h = F.conv2d(h, w[p + "conv2.weight"], w[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in w:
        x = F.conv2d(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"])
    return x + h


def _attention(w, p, x, groups):
    b, c, hh, ww = x.shape
    h = F.group_norm(x, groups, w[p + "group_norm.weight"], w[p + "group_norm.bias"], 1e-6)
    h = h.view(b, c, hh * ww).transpose(1, 2)
    q = F.linear(h, w[p + "to_q.weight"], w[p + "to_q.bias"])
    k = F.linear(h, w[p + "to_k.weight"], w[p + "to_k.bias"])
    v = F.linear(h, w[p + "to_v.weight"], w[p + "to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]      # one head of dim c
    o = F.linear(o, w[p + "to_out.0.weight"], w[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(b, c, hh, ww) + x


@torch.no_grad()
def vae_decode(w: Dict[str, torch.Tensor], cfg: VaeConfig, latents: torch.Tensor, taps=None) -> torch.Tensor:
    """latents [B,4,h,w] (already divided by scaling_factor) -> images [B,3,8h,8w] in ~[-1,1]."""
    g = cfg.norm_num_groups
    z = F.conv2d(latents, w["post_quant_conv.weight"], w["post_quant_conv.bias"])
    h = F.conv2d(z, w["decoder.conv_in.weight"], w["decoder.conv_in.bias"], padding=1)
    if taps is not None: taps["conv_in"] = h
    h = _resnet(w, "decoder.mid_block.resnets.0.", h, g)
    h = _attention(w, "decoder.mid_block.attentions.0.", h, g)
    h = _resnet(w, "decoder.mid_block.resnets.1.", h, g)
    if taps is not None: taps["mid"] = h
    nl = len(cfg.block_out_channels)
    for i in range(nl):
        for j in range(cfg.layers_per_block + 1):
            h = _resnet(w, f"decoder.up_blocks.{i}.resnets.{j}.", h, g)
        if i < nl - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, w[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         w[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
        if taps is not None: taps[f"up{i}"] = h
    h = F.silu(F.group_norm(h, g, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], 1e-6))
    return F.conv2d(h, w["decoder.conv_out.weight"], w["decoder.conv_out.bias"], padding=1)
