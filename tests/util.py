"""Shared helpers for the parity tests."""
import dataclasses

import torch


def oracle_cfg(cfg):
    from oracle.unet import UNetConfig as OC
    d = dataclasses.asdict(cfg)
    return OC(**d)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-12)).item()


def synth_inputs(cfg, latent_batch, seed=29):
    """Seeded latents + prompt / negative embeddings on CPU fp32."""
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn((latent_batch, cfg.in_channels, cfg.sample_size, cfg.sample_size), generator=g)
    pe = torch.randn((latent_batch, cfg.context_len, cfg.cross_attention_dim), generator=g)
    ne = torch.randn((1, cfg.context_len, cfg.cross_attention_dim), generator=g).repeat(latent_batch, 1, 1)
    return lat, pe, ne
