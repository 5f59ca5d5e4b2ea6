"""UNet configuration, parameter enumeration and the synthetic-weight generator.

No SD-1.5 weights exist offline, so benchmarks and parity tests run on
SD-1.5-SHAPED weights drawn from a fixed seed; a local diffusers ``unet`` directory can be
loaded instead when a box has one (``load_unet_state_dict``).  Names and layouts are those of
diffusers' ``UNet2DConditionModel.state_dict()`` -- the same names libsdhip enumerates through
``sd_unet_param_info`` (checked by tests/test_host_cpu.py).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch


@dataclass
class UNetConfig:
    """SD-1.5 UNet2DConditionModel config subset."""
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    attn_levels: Tuple[bool, ...] = (True, True, True, False)
    cross_attention_dim: int = 768
    num_heads: int = 8
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    context_len: int = 77
    time_cond_proj_dim = None  # read by the reference loop at src/models.py:196


def param_shapes(cfg: UNetConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """(name, shape) of every UNet parameter in forward order."""
    out: List[Tuple[str, Tuple[int, ...]]] = []
    add = lambda n, s: out.append((n, tuple(s)))
    c0 = cfg.block_out_channels[0]
    temb = 4 * c0
    nl = len(cfg.block_out_channels)

    def resnet(p, cin, cout):
        add(p + "norm1.weight", (cin,)); add(p + "norm1.bias", (cin,))
        add(p + "conv1.weight", (cout, cin, 3, 3)); add(p + "conv1.bias", (cout,))
        add(p + "time_emb_proj.weight", (cout, temb)); add(p + "time_emb_proj.bias", (cout,))
        add(p + "norm2.weight", (cout,)); add(p + "norm2.bias", (cout,))
        add(p + "conv2.weight", (cout, cout, 3, 3)); add(p + "conv2.bias", (cout,))
        if cin != cout:
            add(p + "conv_shortcut.weight", (cout, cin, 1, 1)); add(p + "conv_shortcut.bias", (cout,))

    def transformer(p, c):
        ctx = cfg.cross_attention_dim
        add(p + "norm.weight", (c,)); add(p + "norm.bias", (c,))
        add(p + "proj_in.weight", (c, c, 1, 1)); add(p + "proj_in.bias", (c,))
        t = p + "transformer_blocks.0."
        for i in (1, 2, 3):
            add(t + f"norm{i}.weight", (c,)); add(t + f"norm{i}.bias", (c,))
        add(t + "attn1.to_q.weight", (c, c)); add(t + "attn1.to_k.weight", (c, c)); add(t + "attn1.to_v.weight", (c, c))
        add(t + "attn1.to_out.0.weight", (c, c)); add(t + "attn1.to_out.0.bias", (c,))
        add(t + "attn2.to_q.weight", (c, c)); add(t + "attn2.to_k.weight", (c, ctx)); add(t + "attn2.to_v.weight", (c, ctx))
        add(t + "attn2.to_out.0.weight", (c, c)); add(t + "attn2.to_out.0.bias", (c,))
        add(t + "ff.net.0.proj.weight", (8 * c, c)); add(t + "ff.net.0.proj.bias", (8 * c,))
        add(t + "ff.net.2.weight", (c, 4 * c)); add(t + "ff.net.2.bias", (c,))
        add(p + "proj_out.weight", (c, c, 1, 1)); add(p + "proj_out.bias", (c,))

    add("time_embedding.linear_1.weight", (temb, c0)); add("time_embedding.linear_1.bias", (temb,))
    add("time_embedding.linear_2.weight", (temb, temb)); add("time_embedding.linear_2.bias", (temb,))
    add("conv_in.weight", (c0, cfg.in_channels, 3, 3)); add("conv_in.bias", (c0,))
    ch = c0
    skip_ch = [c0]
    for i in range(nl):
        co = cfg.block_out_channels[i]
        for j in range(cfg.layers_per_block):
            resnet(f"down_blocks.{i}.resnets.{j}.", ch, co)
            ch = co
            if cfg.attn_levels[i]:
                transformer(f"down_blocks.{i}.attentions.{j}.", co)
            skip_ch.append(co)
        if i < nl - 1:
            add(f"down_blocks.{i}.downsamplers.0.conv.weight", (co, co, 3, 3))
            add(f"down_blocks.{i}.downsamplers.0.conv.bias", (co,))
            skip_ch.append(co)
    resnet("mid_block.resnets.0.", ch, ch)
    transformer("mid_block.attentions.0.", ch)
    resnet("mid_block.resnets.1.", ch, ch)
    for i in range(nl):
        lev = nl - 1 - i
        co = cfg.block_out_channels[lev]
        for j in range(cfg.layers_per_block + 1):
            resnet(f"up_blocks.{i}.resnets.{j}.", ch + skip_ch.pop(), co)
            ch = co
            if cfg.attn_levels[lev]:
                transformer(f"up_blocks.{i}.attentions.{j}.", co)
        if i < nl - 1:
            add(f"up_blocks.{i}.upsamplers.0.conv.weight", (co, co, 3, 3))
            add(f"up_blocks.{i}.upsamplers.0.conv.bias", (co,))
    add("conv_norm_out.weight", (c0,)); add("conv_norm_out.bias", (c0,))
    add("conv_out.weight", (cfg.out_channels, c0, 3, 3)); add("conv_out.bias", (cfg.out_channels,))
    return out


_SYNTHETIC_CACHE: Dict[tuple, Dict[str, torch.Tensor]] = {}


def make_synthetic_state_dict(cfg: UNetConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Seeded SD-1.5-shaped weights, fp32 values already on the bf16 grid.

    Conv/linear weights ~ N(0, 1/fan_in) (unit gain, so activations stay O(1) through the
    normalised blocks), biases ~ N(0, 0.05^2), norm gains 1 + N(0, 0.1^2), norm shifts
    N(0, 0.1^2) (non-trivial affine so parity tests exercise it).  Rounding to bf16 here means
    the CPU oracle and the HIP path consume bit-identical parameters.

    The values do not depend on ``sample_size`` (no parameter shape does); generating 0.86 G Gaussians takes ~10 s, so the
    result is cached per (architecture, seed) for the life of the process and every call returns a NEW dict over the same
    read-only tensors (callers replace entries -- LoRA fusion, tests -- and never write into a tensor).
    """
    key = (cfg.in_channels, cfg.out_channels, tuple(cfg.block_out_channels), cfg.layers_per_block, tuple(cfg.attn_levels),
           cfg.cross_attention_dim, int(seed))
    if key in _SYNTHETIC_CACHE:
        return dict(_SYNTHETIC_CACHE[key])
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg):
        leaf = name.rsplit(".", 2)[-2]
        is_norm = leaf.startswith("norm") or leaf == "conv_norm_out"
        if name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * (0.1 if is_norm else 0.05)
        elif is_norm:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = math.prod(shape[1:])
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        sd[name] = t.to(torch.bfloat16).float()
    if len(_SYNTHETIC_CACHE) >= 2:            # (3.4 GB per SD-1.5-sized entry)
        _SYNTHETIC_CACHE.pop(next(iter(_SYNTHETIC_CACHE)))
    _SYNTHETIC_CACHE[key] = sd
    return dict(sd)


def load_unet_config(model_dir: str) -> "UNetConfig | None":
    """``<dir>/unet/config.json`` of a LOCAL diffusers checkpoint -> ``UNetConfig`` (None if the file is absent: the
    SD-1.5 defaults then apply).  Only the keys this build implements are read; a checkpoint that needs anything else
    (another block type, ``use_linear_projection``, per-level head counts that are not ``attention_head_dim`` heads
    at every level, an activation other than silu) is refused here rather than mis-run."""
    import json
    for cand in (os.path.join(model_dir, "unet", "config.json"), os.path.join(model_dir, "config.json")):
        if os.path.isfile(cand):
            with open(cand) as f:
                c = json.load(f)
            break
    else:
        return None
    down = list(c.get("down_block_types", ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]))
    up = list(c.get("up_block_types", ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3))
    known = {"CrossAttnDownBlock2D": True, "DownBlock2D": False}
    if any(d not in known for d in down):
        raise NotImplementedError(f"down_block_types {down}: CrossAttnDownBlock2D / DownBlock2D are built")
    attn = tuple(known[d] for d in down)
    if [u == "CrossAttnUpBlock2D" for u in up] != list(reversed(attn)):
        raise NotImplementedError(f"up_block_types {up} do not mirror down_block_types {down}")
    heads = c.get("attention_head_dim", 8)          # SD-1.5 quirk: this key holds the NUMBER of heads
    if isinstance(heads, (list, tuple)):
        if len(set(heads)) != 1:
            raise NotImplementedError("per-level head counts are not built")
        heads = heads[0]
    for key, want in (("use_linear_projection", False), ("act_fn", "silu"), ("flip_sin_to_cos", True), ("freq_shift", 0),
                      ("time_cond_proj_dim", None), ("class_embed_type", None), ("addition_embed_type", None),
                      ("dual_cross_attention", False), ("only_cross_attention", False), ("upcast_attention", False)):
        if c.get(key, want) != want:
            raise NotImplementedError(f"unet config {key}={c[key]!r}: this build implements {want!r} (SD-1.5)")
    return UNetConfig(sample_size=int(c.get("sample_size", 64)), in_channels=int(c.get("in_channels", 4)),
                      out_channels=int(c.get("out_channels", 4)),
                      block_out_channels=tuple(int(v) for v in c.get("block_out_channels", (320, 640, 1280, 1280))),
                      layers_per_block=int(c.get("layers_per_block", 2)), attn_levels=attn,
                      cross_attention_dim=int(c.get("cross_attention_dim", 768)), num_heads=int(heads),
                      norm_num_groups=int(c.get("norm_num_groups", 32)), norm_eps=float(c.get("norm_eps", 1e-5)))


def load_unet_state_dict(model_dir: str) -> Dict[str, torch.Tensor]:
    """Load a LOCAL diffusers UNet (``<dir>/unet/diffusion_pytorch_model.safetensors``).
    Never fetches: names that are not local directories raise."""
    from safetensors.torch import load_file
    for cand in (os.path.join(model_dir, "unet", "diffusion_pytorch_model.safetensors"),
                 os.path.join(model_dir, "diffusion_pytorch_model.safetensors")):
        if os.path.isfile(cand):
            return {k: v.float() for k, v in load_file(cand).items()}
    raise FileNotFoundError(
        f"no local UNet weights under {model_dir!r}; model names are network fetches and are "
        "unavailable offline -- use synthetic weights or point at a local diffusers directory")


def fuse_lora_state_dict(sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor], scale: float = 1.0) -> int:
    """``pipe.load_lora_weights(...); pipe.fuse_lora()`` on the host copy of the UNet weights
    (``src/experiments/consistency_model.py:20-21``): ``W += scale * (alpha / r) * up @ down`` for every adapted
    module, linear or conv (3x3 ``down`` [r,I,kh,kw] with 1x1 ``up`` [O,r,1,1]).  Accepts the two layouts LoRA
    files for SD-1.5 come in [upstream-recall]: kohya (``lora_unet_<module with _>.lora_down/.lora_up.weight`` +
    ``.alpha``, what ``latent-consistency/lcm-lora-sdv1-5`` ships) and peft/diffusers
    (``unet.<module>.lora_A/.lora_B.weight``, alpha = r).  Text-encoder entries are ignored.  Returns the number of
    fused modules; raises if an adapted UNet module does not exist."""
    flat = {k[: -len(".weight")].replace(".", "_"): k for k in sd if k.endswith(".weight")}
    pairs = {}
    for k, v in lora_sd.items():
        if k.startswith("lora_te") or k.startswith("text_encoder."):
            continue
        if k.startswith("lora_unet_"):
            mod, _, leaf = k[len("lora_unet_"):].partition(".")
            target = flat.get(mod)
            if target is None:
                raise KeyError(f"LoRA module {mod!r} has no counterpart in the UNet")
            slot = {"lora_down.weight": "down", "lora_up.weight": "up", "alpha": "alpha"}.get(leaf)
        elif k.startswith("unet."):
            body = k[len("unet."):]
            for tag, slot_ in ((".lora_A.weight", "down"), (".lora_B.weight", "up"), (".lora.down.weight", "down"),
                               (".lora.up.weight", "up")):
                if body.endswith(tag):
                    target, slot = body[: -len(tag)] + ".weight", slot_
                    break
            else:
                continue
            if target not in sd:
                raise KeyError(f"LoRA module {target!r} has no counterpart in the UNet")
        else:
            continue
        if slot is not None:
            pairs.setdefault(target, {})[slot] = v
    n = 0
    for target, p in pairs.items():
        if "down" not in p or "up" not in p:
            raise KeyError(f"incomplete LoRA pair for {target}")
        down, up = p["down"].float(), p["up"].float()
        r = down.shape[0]
        alpha = float(p["alpha"]) if "alpha" in p else float(r)
        w = sd[target]
        delta = (up.reshape(up.shape[0], r) @ down.reshape(r, -1)).reshape(w.shape)
        sd[target] = (w.float() + scale * (alpha / r) * delta).to(torch.bfloat16).float()
        n += 1
    return n
