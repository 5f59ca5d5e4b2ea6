"""GPU parity of the CLIP text encoder on libsdhip against the CPU oracle -- which is itself
pinned by the transformers golden vectors (tests/test_clip_cpu.py).  bf16 kernels vs fp32: rel-L2 <= 1.5e-2."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import CLIP_TEXTS, CLIP_TINY, cosine, rel_l2, synthetic_clip_vocab

TOL = 1.5e-2


def test_tiny_model_matches_golden_and_oracle():
    from oracle.clip import ClipTextConfig as OC, clip_text_forward
    from sonicdiffusionbayeslab_amd.clip import ClipTextConfig, HipClipTextModel, make_synthetic_clip_state_dict
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "clip_golden.json")))
    cfg = ClipTextConfig(**CLIP_TINY)
    sd = make_synthetic_clip_state_dict(cfg, seed=777)
    m = HipClipTextModel(cfg, sd)
    ids = torch.tensor(gold["input_ids"])
    out = m.encode(ids)
    want = torch.tensor(gold["last_hidden_state"])
    ref = clip_text_forward(sd, OC(**CLIP_TINY), ids)
    print(f"tiny CLIP: vs transformers golden rel-L2 {rel_l2(out, want):.3e}, vs oracle {rel_l2(out, ref):.3e}")
    assert out.dtype == torch.float32 and tuple(out.shape) == (len(CLIP_TEXTS), 16, 64)
    assert rel_l2(out, want) < TOL and rel_l2(out, ref) < TOL and cosine(out, ref) > 0.9995
    with pytest.raises(ValueError):
        m.encode(torch.full((1, 16), 9999))


def test_full_size_text_tower_and_prompt_encoder():
    """ViT-L/14 text tower shape (12 x 768, 77 tokens), 8 prompts, through tokenizer + ClipPromptEncoder into the
    pipeline's `text_encoder` slot."""
    from oracle.clip import ClipTextConfig as OC, clip_text_forward
    from sonicdiffusionbayeslab_amd.clip import (ClipBpeTokenizer, ClipPromptEncoder, ClipTextConfig, HipClipTextModel,
                                                 make_synthetic_clip_state_dict)
    vocab, merges = synthetic_clip_vocab()
    cfg = ClipTextConfig(vocab_size=len(vocab))
    sd = make_synthetic_clip_state_dict(cfg, seed=11)
    tk = ClipBpeTokenizer(vocab, merges)
    enc = ClipPromptEncoder(tk, HipClipTextModel(cfg, sd))
    prompts = CLIP_TEXTS + ["a cat and the thing", "photo of of the"]
    out = enc(prompts)
    ids = tk(prompts)
    kw = {k: getattr(cfg, k) for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                        "intermediate_size", "max_position_embeddings", "layer_norm_eps")}
    ref = clip_text_forward(sd, OC(**kw), ids.long())
    err, cs = rel_l2(out, ref), cosine(out, ref)
    print(f"CLIP ViT-L/14 text tower, 8 prompts: rel-L2 {err:.3e} cos {cs:.5f}")
    assert tuple(out.shape) == (8, 77, 768) and err < TOL and cs > 0.9995
    # causal: the embedding of token i must not depend on later tokens
    ids2 = ids.clone(); ids2[:, 40:] = vocab["x"]
    out2 = enc.text_model.encode(ids2)
    assert torch.equal(out[:, :40], out2[:, :40]) and not torch.equal(out[:, 40:], out2[:, 40:])
    # plugs into the pipeline
    from sonicdiffusionbayeslab_amd.models import StableDiffusionModel
    from sonicdiffusionbayeslab_amd.registry import schedulers_registry
    from sonicdiffusionbayeslab_amd.schedulers import PNDMConfigStub
    from sonicdiffusionbayeslab_amd.weights import UNetConfig, make_synthetic_state_dict
    ucfg = UNetConfig(sample_size=16)
    model = StableDiffusionModel(unet_config=ucfg, state_dict=make_synthetic_state_dict(ucfg, seed=1234), text_encoder=enc).to("cuda:0")
    model.scheduler = schedulers_registry["ddim_scheduler"].from_config(PNDMConfigStub().config)
    res, secs, _ = model(["A photo of the cat", "the thing"], num_inference_steps=2, guidance_scale=7.5,
                         generator=torch.Generator().manual_seed(29), output_type="latent")
    assert tuple(res.images.shape) == (2, 4, 16, 16) and torch.isfinite(res.images).all()
