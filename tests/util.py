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


def synthetic_clip_vocab():
    """A small CLIP-style byte-level BPE vocabulary (every byte symbol, its ``</w>`` form, a few merges and the
    two special tokens) for the tokenizer tests and ``tests/golden/make_clip_golden.py``."""
    from sonicdiffusionbayeslab_amd.clip import bytes_to_unicode
    b2u = bytes_to_unicode()
    chars = [b2u[b] for b in range(256)]
    vocab = {}
    for c in chars:
        vocab[c] = len(vocab)
    for c in chars:
        vocab[c + "</w>"] = len(vocab)
    merges = [("t", "h"), ("th", "e</w>"), ("a", "n"), ("an", "d</w>"), ("c", "a"), ("ca", "t</w>"), ("i", "n"),
              ("in", "g</w>"), ("p", "h"), ("ph", "o"), ("pho", "t"), ("phot", "o</w>"), ("o", "f</w>")]
    for a, b in merges:
        vocab[a + b] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    return vocab, merges


CLIP_TINY = dict(vocab_size=527, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=16)
CLIP_TEXTS = ["A photo of the cat", "the  CAT and\tthe hat's thing!!", "naïve café 123 photo", "", "x" * 200,
              "<|endoftext|> of <|startoftext|>"]
