"""fp32 CPU restatement of the CLIP text encoder behind ``encode_prompt`` (``src/models.py:139-155``).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The arithmetic lives in transformers==4.48.0
(``poetry.lock:2741-2742``; ``CLIPTextModel`` -> ``CLIPTextTransformer``, modeling_clip.py), absent from
/root/reference; restated here from its published algorithm: token + learned position embeddings, pre-LN
encoder layers (q scaled by d^-1/2, causal mask, softmax, out_proj; quick_gelu MLP), final LayerNorm;
``last_hidden_state`` is what Stable Diffusion consumes.  Unlike the UNet oracle this one IS pinned: the
transformers build in this image (5.15.0, third-party, importable offline) produces the same numbers for a
seeded random-init ``CLIPTextModel`` -- ``tests/golden/make_clip_golden.py`` wrote the committed fixture and
``tests/test_oracle_cpu.py`` checks against it (and against the live library when importable)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class ClipTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


@torch.no_grad()
def clip_text_forward(w, cfg: ClipTextConfig, input_ids: torch.Tensor) -> torch.Tensor:
    """``CLIPTextModel(input_ids).last_hidden_state``: ``w`` uses the ``text_model.*`` names."""
    B, L = input_ids.shape
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    d = H // nh
    P = lambda n: w["text_model." + n].float()
    h = P("embeddings.token_embedding.weight")[input_ids.long()] + P("embeddings.position_embedding.weight")[:L]
    mask = torch.full((L, L), float("-inf")).triu(1)                      # causal: key j <= query i
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}."
        r = h
        x = F.layer_norm(h, (H,), P(p + "layer_norm1.weight"), P(p + "layer_norm1.bias"), cfg.layer_norm_eps)
        q = F.linear(x, P(p + "self_attn.q_proj.weight"), P(p + "self_attn.q_proj.bias")) * d ** -0.5
        k = F.linear(x, P(p + "self_attn.k_proj.weight"), P(p + "self_attn.k_proj.bias"))
        v = F.linear(x, P(p + "self_attn.v_proj.weight"), P(p + "self_attn.v_proj.bias"))
        sp = lambda t: t.view(B, L, nh, d).transpose(1, 2)
        a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) + mask, dim=-1) @ sp(v)
        a = a.transpose(1, 2).reshape(B, L, H)
        h = r + F.linear(a, P(p + "self_attn.out_proj.weight"), P(p + "self_attn.out_proj.bias"))
        r = h
        x = F.layer_norm(h, (H,), P(p + "layer_norm2.weight"), P(p + "layer_norm2.bias"), cfg.layer_norm_eps)
        x = quick_gelu(F.linear(x, P(p + "mlp.fc1.weight"), P(p + "mlp.fc1.bias")))
        h = r + F.linear(x, P(p + "mlp.fc2.weight"), P(p + "mlp.fc2.bias"))
    return F.layer_norm(h, (H,), P("final_layer_norm.weight"), P("final_layer_norm.bias"), cfg.layer_norm_eps)
