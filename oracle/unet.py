"""fp32 CPU restatement of diffusers' SD-1.5 ``UNet2DConditionModel.forward``.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- parity unpinned.

Follows the call site ``src/models.py:227-235`` of the reference
(``self.unet(latent_model_input, t, encoder_hidden_states=prompt_embeds, ...)[0]``) and the
architecture of diffusers==0.32.1 (poetry.lock:454-455) as written down in SURVEY.md
App. A.1-A.4.  Every operator is a stock ``torch.nn.functional`` CPU op -- the same ATen
kernels the reference's CPU path would dispatch to.

Weights are a plain ``dict[str, Tensor]`` keyed by the diffusers state_dict names
(``down_blocks.0.resnets.0.conv1.weight`` ...), in diffusers layouts (conv OIHW, linear
[out, in]).

DeepCache 0.1.1 semantics (reference call site
``src/experiments/deep_cache.py:24-29,58``) are restated through ``DeepCacheState``: every
wrapped sub-module goes through ``_cached`` which returns the stored output on skip steps.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # which down levels carry Transformer2DModel blocks (SD-1.5: all but the last)
    attn_levels: Tuple[bool, ...] = (True, True, True, False)
    cross_attention_dim: int = 768
    num_heads: int = 8          # diffusers `attention_head_dim=8` is used as the head COUNT
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    context_len: int = 77


# --------------------------------------------------------------------------------------
# DeepCache state (A.5)
# --------------------------------------------------------------------------------------
@dataclass
class DeepCacheState:
    cache_interval: int = 1
    cache_branch_id: int = 0
    enabled: bool = False
    cur_timestep: int = 0                # index of t in scheduler.timesteps
    start_timestep: Optional[int] = None
    cached: Dict[tuple, object] = field(default_factory=dict)

    @property
    def cache_layer_id(self) -> int:
        return self.cache_branch_id % 3

    @property
    def cache_block_id(self) -> int:
        return self.cache_branch_id // 3

    def is_skip(self, block_i: int, layer_i: int, blocktype: str) -> bool:
        if not self.enabled:
            return False
        if self.start_timestep is None:
            self.start_timestep = self.cur_timestep
        if (self.cur_timestep - self.start_timestep) % self.cache_interval == 0:
            return False
        if block_i > self.cache_block_id or blocktype == "mid":
            return True
        if block_i < self.cache_block_id:
            return False
        if blocktype == "down":
            return layer_i >= self.cache_layer_id
        return layer_i > self.cache_layer_id


def _cached(dc: Optional[DeepCacheState], key: tuple, block_i: int, layer_i: int, blocktype: str, fn):
    """Run ``fn`` or return the cached output of the wrapped module (A.5)."""
    if dc is None or not dc.enabled:
        return fn()
    if dc.is_skip(block_i, layer_i, blocktype):
        return dc.cached[key]
    out = fn()
    dc.cached[key] = out
    return out


# --------------------------------------------------------------------------------------
# building blocks (A.2-A.4)
# --------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int = 320) -> torch.Tensor:
    """Sinusoidal embedding, flip_sin_to_cos=True, freq_shift=0 (A.2 step 1)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


# Optional wall-clock bound for callers that TIME the oracle (bench.py's cpu_baseline leg): checked between blocks, so an
# unexpectedly slow dtype path (fp16 on a CPU without vectorised half kernels) stops after at most one block too many.
DEADLINE = None
BLOCKS_DONE = 0


def _tick():
    global BLOCKS_DONE
    BLOCKS_DONE += 1
    if DEADLINE is not None and time.time() > DEADLINE:
        raise TimeoutError(f"oracle deadline reached after {BLOCKS_DONE} blocks")


def resnet_block(w, p: str, x: torch.Tensor, temb: torch.Tensor, cfg: UNetConfig, fq=None) -> torch.Tensor:
    """ResnetBlock2D (A.3).  ``fq`` (oracle/fp8.py::Fp8Emulation) applies the build's fp8 rounding points."""
    _tick()
    g = cfg.norm_num_groups
    W = (lambda n: fq.w(w, n)) if fq is not None else (lambda n: w[n])
    h = F.group_norm(x, g, w[p + "norm1.weight"], w[p + "norm1.bias"], cfg.norm_eps)
    h = F.silu(h)
    if fq is not None:
        h = fq.act_norm(h, p + "norm1")
    h = F.conv2d(h, W(p + "conv1.weight"), w[p + "conv1.bias"], padding=1)
    tp = F.linear(F.silu(temb), w[p + "time_emb_proj.weight"], w[p + "time_emb_proj.bias"])
    h = h + tp[:, :, None, None]
    h = F.group_norm(h, g, w[p + "norm2.weight"], w[p + "norm2.bias"], cfg.norm_eps)
    h = F.silu(h)
    if fq is not None:
        h = fq.act_norm(h, p + "norm2")
    h = F.conv2d(h, W(p + "conv2.weight"), w[p + "conv2.bias"], padding=1)
    if (p + "conv_shortcut.weight") in w:
        x = F.conv2d(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"])
    return x + h


def _attention(w, p: str, x: torch.Tensor, ctx: torch.Tensor, heads: int, fq=None) -> torch.Tensor:
    """diffusers Attention + AttnProcessor2_0: q/k/v no bias, to_out.0 with bias, SDPA."""
    b, n, c = x.shape
    W = (lambda nm: fq.w(w, nm)) if fq is not None else (lambda nm: w[nm])
    q = F.linear(x, W(p + "to_q.weight"))
    k = F.linear(ctx, W(p + "to_k.weight"))
    v = F.linear(ctx, W(p + "to_v.weight"))
    d = c // heads
    q = q.view(b, n, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(b, n, c)
    return F.linear(o, w[p + "to_out.0.weight"], w[p + "to_out.0.bias"])


def transformer_block(w, p: str, x: torch.Tensor, ctx: torch.Tensor, cfg: UNetConfig, fq=None) -> torch.Tensor:
    """Transformer2DModel with one BasicTransformerBlock (A.4), use_linear_projection=False."""
    _tick()
    b, c, hh, ww = x.shape
    res = x
    W = (lambda n: fq.w(w, n)) if fq is not None else (lambda n: w[n])
    A = fq.act_norm if fq is not None else (lambda v, name=None: v)
    h = F.group_norm(x, cfg.norm_num_groups, w[p + "norm.weight"], w[p + "norm.bias"], 1e-6)
    h = F.conv2d(A(h, p + "norm"), W(p + "proj_in.weight"), w[p + "proj_in.bias"])
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    t = p + "transformer_blocks.0."
    n1 = A(F.layer_norm(h, (c,), w[t + "norm1.weight"], w[t + "norm1.bias"], 1e-5), t + "norm1")
    h = h + _attention(w, t + "attn1.", n1, n1, cfg.num_heads, fq)
    n2 = F.layer_norm(h, (c,), w[t + "norm2.weight"], w[t + "norm2.bias"], 1e-5)
    h = h + _attention(w, t + "attn2.", n2, ctx, cfg.num_heads)          # the prompt cross-attention stays bf16
    n3 = A(F.layer_norm(h, (c,), w[t + "norm3.weight"], w[t + "norm3.bias"], 1e-5), t + "norm3")
    proj = F.linear(n3, W(t + "ff.net.0.proj.weight"), w[t + "ff.net.0.proj.bias"])
    a, gate = proj.chunk(2, dim=-1)
    hid = a * F.gelu(gate)
    if fq is not None:
        hid = fq.act_ff(hid, t + "ff.net.0")
    ff = F.linear(hid, W(t + "ff.net.2.weight"), w[t + "ff.net.2.bias"])
    h = h + ff
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    h = F.conv2d(h, w[p + "proj_out.weight"], w[p + "proj_out.bias"])
    return h + res


# --------------------------------------------------------------------------------------
# full forward (A.2)
# --------------------------------------------------------------------------------------
def unet_forward(w: Dict[str, torch.Tensor], cfg: UNetConfig, sample: torch.Tensor, t,
                 ctx: torch.Tensor, dc: Optional[DeepCacheState] = None,
                 taps: Optional[dict] = None, fq=None) -> torch.Tensor:
    """eps = UNet(sample [N,4,H,W], t scalar, ctx [N,L,768]) -> [N,4,H,W].

    ``taps`` (optional dict) receives named intermediate activations for layer-by-layer
    parity debugging.  ``fq`` (oracle/fp8.py::Fp8Emulation) applies the rounding points of the build's fp8 plan.
    """
    n = sample.shape[0]
    nlev = len(cfg.block_out_channels)
    tt = torch.as_tensor(t, dtype=torch.float32).reshape(-1)
    if tt.numel() == 1:
        tt = tt.expand(n)
    temb = timestep_embedding(tt, cfg.block_out_channels[0]).to(sample.dtype)      # "cast to model dtype" (A.2 step 1)
    temb = F.linear(temb, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"])
    temb = F.silu(temb)
    temb = F.linear(temb, w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])

    h = F.conv2d(sample, w["conv_in.weight"], w["conv_in.bias"], padding=1)
    if taps is not None:
        taps["conv_in"] = h
    skips = [h]

    # ---- down ----
    for i in range(nlev):
        def run_down(i=i, h_in=h):
            hcur = h_in
            outs = []
            for j in range(cfg.layers_per_block):
                p = f"down_blocks.{i}.resnets.{j}."
                hcur = _cached(dc, ("down", "resnet", i, j), i, j, "down",
                               lambda hcur=hcur, p=p: resnet_block(w, p, hcur, temb, cfg, fq))
                if cfg.attn_levels[i]:
                    p = f"down_blocks.{i}.attentions.{j}."
                    hcur = _cached(dc, ("down", "attentions", i, j), i, j, "down",
                                   lambda hcur=hcur, p=p: transformer_block(w, p, hcur, ctx, cfg, fq))
                outs.append(hcur)
            if i < nlev - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv."
                hcur = _cached(dc, ("down", "downsampler", i, cfg.layers_per_block), i,
                               cfg.layers_per_block, "down",
                               lambda hcur=hcur, p=p: F.conv2d(hcur, w[p + "weight"], w[p + "bias"],
                                                              stride=2, padding=1))
                outs.append(hcur)
            return hcur, outs
        h, outs = _cached(dc, ("down", "block", i, 0), i, 0, "down", run_down)
        skips.extend(outs)
        if taps is not None:
            taps[f"down{i}"] = h

    # ---- mid ----
    def run_mid(h_in=h):
        hcur = resnet_block(w, "mid_block.resnets.0.", h_in, temb, cfg, fq)
        hcur = transformer_block(w, "mid_block.attentions.0.", hcur, ctx, cfg, fq)
        hcur = resnet_block(w, "mid_block.resnets.1.", hcur, temb, cfg, fq)
        return hcur
    h = _cached(dc, ("mid", "mid_block", 0, 0), 0, 0, "mid", run_mid)
    if taps is not None:
        taps["mid"] = h

    # ---- up ----
    nres = cfg.layers_per_block + 1
    for i in range(nlev):
        lev = nlev - 1 - i                       # resolution level this up block works at
        has_attn = cfg.attn_levels[lev]
        res_samples = skips[-nres:]
        skips = skips[:-nres]
        rb = nlev - 1 - i                        # DeepCache reversed block index

        def run_up(i=i, h_in=h, res_samples=res_samples, has_attn=has_attn, rb=rb):
            hcur = h_in
            rs = list(res_samples)
            for j in range(nres):
                skip = rs.pop()
                rl = nres - 1 - j                # DeepCache reversed layer index
                # the block concatenates before calling the (possibly skipped) resnet
                p = f"up_blocks.{i}.resnets.{j}."
                hcur = _cached(dc, ("up", "resnet", rb, rl), rb, rl, "up",
                               lambda hcur=hcur, skip=skip, p=p: resnet_block(
                                   w, p, torch.cat([hcur, skip], dim=1), temb, cfg, fq))
                if has_attn:
                    p = f"up_blocks.{i}.attentions.{j}."
                    hcur = _cached(dc, ("up", "attentions", rb, rl), rb, rl, "up",
                                   lambda hcur=hcur, p=p: transformer_block(w, p, hcur, ctx, cfg, fq))
            if i < nlev - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv."
                hcur = _cached(dc, ("up", "upsampler", rb, 0), rb, 0, "up",
                               lambda hcur=hcur, p=p: F.conv2d(
                                   F.interpolate(hcur, scale_factor=2.0, mode="nearest"),
                                   w[p + "weight"], w[p + "bias"], padding=1))
            return hcur
        h = _cached(dc, ("up", "block", rb, 0), rb, 0, "up", run_up)
        if taps is not None:
            taps[f"up{i}"] = h

    h = F.group_norm(h, cfg.norm_num_groups, w["conv_norm_out.weight"], w["conv_norm_out.bias"], cfg.norm_eps)
    h = F.silu(h)
    return F.conv2d(h, w["conv_out.weight"], w["conv_out.bias"], padding=1)
