"""``AutoencoderKL`` decoder on libsdhip -- SURVEY.md §8f "next" row 1.

Replaces ``self.vae.decode(latents / self.vae.config.scaling_factor)`` of the reference pipeline
(``src/models.py:287-302``): 2.5 TFLOP per 512x512 image, outside the reference's timed loop but what
turns latents into the ``[B,3,512,512]`` tensor the harness consumes (``output_type="pt"``).  Same
kernels as the UNet (implicit-GEMM conv with fused upsample, GroupNorm+SiLU, GEMM) plus a row softmax
for the single 512-wide attention head of the mid block.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib


@dataclass
class VaeConfig:
    sample_size: int = 64                      # latent H = W
    in_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


def vae_param_shapes(cfg: VaeConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    out: List[Tuple[str, Tuple[int, ...]]] = []
    add = lambda n, s: out.append((n, tuple(s)))
    nl = len(cfg.block_out_channels)
    top = cfg.block_out_channels[-1]

    def resnet(p, cin, cout):
        add(p + "norm1.weight", (cin,)); add(p + "norm1.bias", (cin,))
        add(p + "conv1.weight", (cout, cin, 3, 3)); add(p + "conv1.bias", (cout,))
        add(p + "norm2.weight", (cout,)); add(p + "norm2.bias", (cout,))
        add(p + "conv2.weight", (cout, cout, 3, 3)); add(p + "conv2.bias", (cout,))
        if cin != cout:
            add(p + "conv_shortcut.weight", (cout, cin, 1, 1)); add(p + "conv_shortcut.bias", (cout,))

    add("post_quant_conv.weight", (cfg.in_channels, cfg.in_channels, 1, 1)); add("post_quant_conv.bias", (cfg.in_channels,))
    add("decoder.conv_in.weight", (top, cfg.in_channels, 3, 3)); add("decoder.conv_in.bias", (top,))
    resnet("decoder.mid_block.resnets.0.", top, top)
    a = "decoder.mid_block.attentions.0."
    add(a + "group_norm.weight", (top,)); add(a + "group_norm.bias", (top,))
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        add(a + n + ".weight", (top, top)); add(a + n + ".bias", (top,))
    resnet("decoder.mid_block.resnets.1.", top, top)
    ch = top
    for i in range(nl):
        co = cfg.block_out_channels[nl - 1 - i]
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}.", ch, co)
            ch = co
        if i < nl - 1:
            add(f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (co, co, 3, 3))
            add(f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (co,))
    add("decoder.conv_norm_out.weight", (ch,)); add("decoder.conv_norm_out.bias", (ch,))
    add("decoder.conv_out.weight", (cfg.out_channels, ch, 3, 3)); add("decoder.conv_out.bias", (cfg.out_channels,))
    return out


def make_synthetic_vae_state_dict(cfg: VaeConfig, seed: int = 4321) -> Dict[str, torch.Tensor]:
    """Seeded SD-1.5-shaped decoder weights on the bf16 grid (no VAE weights exist offline)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in vae_param_shapes(cfg):
        leaf = name.rsplit(".", 2)[-2]
        is_norm = "norm" in leaf
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * (0.1 if is_norm else 0.05)
        elif is_norm:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(math.prod(shape[1:]))
        sd[name] = t.to(torch.bfloat16).float()
    return sd


def load_vae_state_dict(model_dir: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    for cand in (os.path.join(model_dir, "vae", "diffusion_pytorch_model.safetensors"),
                 os.path.join(model_dir, "diffusion_pytorch_model.safetensors")):
        if os.path.isfile(cand):
            return {k: v.float() for k, v in load_file(cand).items()}
    raise FileNotFoundError(f"no local VAE weights under {model_dir!r}")


def _c_config(cfg: VaeConfig) -> _lib.SdUnetConfig:
    c = _lib.SdUnetConfig()
    c.sample_size, c.in_channels, c.out_channels = cfg.sample_size, cfg.in_channels, cfg.out_channels
    c.num_levels = len(cfg.block_out_channels)
    for i, v in enumerate(cfg.block_out_channels):
        c.block_out_channels[i] = v
    c.layers_per_block = cfg.layers_per_block
    c.norm_num_groups, c.norm_eps = cfg.norm_num_groups, 1e-6
    c.num_heads, c.cross_attention_dim, c.context_len = 1, 64, 1
    return c


class HipVaeDecoder:
    """``vae.decode`` replacement: ``decoder(latents) -> [B,3,8h,8w]`` fp32 on the GPU."""

    def __init__(self, config: VaeConfig, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.SdHipError("HipVaeDecoder needs an MI355X (no CPU fallback exists)")
        self.config = config
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        torch.cuda.set_device(self.device)
        _lib.check(self._lib.sd_vae_create(C.byref(_c_config(config)), C.byref(self._handle)), "sd_vae_create")
        for name, shape in vae_param_shapes(config):
            if name not in state_dict:
                raise KeyError(f"state_dict lacks VAE parameter {name!r}")
            t = state_dict[name].detach().to("cpu", torch.float32).contiguous()
            if t.dim() == 4 and len(shape) == 2:       # legacy attention weights stored as 1x1 convs
                t = t.reshape(shape)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
            _lib.check(self._lib.sd_unet_load_param(self._handle, name.encode(), t.data_ptr(), t.numel()),
                       f"load_param({name})")
        _lib.check(self._lib.sd_unet_finalize(self._handle), "finalize")
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = None

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                self._lib.sd_unet_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or self._ws_batch != batch:
            n = self._lib.sd_unet_workspace_bytes(self._handle, batch, -1)
            if n < 0:
                _lib.check(-1, "sd_unet_workspace_bytes")
            self._ws = None
            self._ws = torch.empty(n + 256, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def decode(self, latents: torch.Tensor, latent_scale: float = 1.0, chunk: int = 8) -> torch.Tensor:
        """``latents * latent_scale`` -> decoded images (pass ``1 / scaling_factor`` to fuse the division
        of src/models.py:288).  Large batches are decoded ``chunk`` images at a time."""
        lat = latents.to(self.device, torch.float32).contiguous()
        b, c, h, w = lat.shape
        if c != self.config.in_channels or h != self.config.sample_size or w != self.config.sample_size:
            raise ValueError(f"latents must be [B,{self.config.in_channels},{self.config.sample_size},"
                             f"{self.config.sample_size}], got {tuple(lat.shape)}")
        out = torch.empty((b, self.config.out_channels, 8 * h, 8 * w), dtype=torch.float32, device=self.device)
        for s in range(0, b, chunk):
            n = min(chunk, b - s)
            ws = self._workspace(n)
            wsp = (ws.data_ptr() + 255) // 256 * 256
            _lib.check(self._lib.sd_vae_decode(self._handle, _lib.current_stream(), lat[s:s + n].data_ptr(), n,
                                               float(latent_scale), out[s:s + n].data_ptr(), wsp, ws.numel() - 256),
                       "sd_vae_decode")
        return out

    __call__ = decode
