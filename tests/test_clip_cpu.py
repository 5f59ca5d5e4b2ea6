"""CPU tests of the CLIP text path: the oracle and the tokenizer against the golden vectors the
transformers build of this image produced (tests/golden/make_clip_golden.py), and the library's parameter table."""
import ctypes as C
import json
import os

import pytest
import torch

from sonicdiffusionbayeslab_amd import _lib
from sonicdiffusionbayeslab_amd.clip import (ClipBpeTokenizer, ClipTextConfig, clip_param_shapes,
                                             make_synthetic_clip_state_dict, normalise_clip_state_dict)
from tests.util import CLIP_TEXTS, CLIP_TINY, synthetic_clip_vocab

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "clip_golden.json")))


def test_tokenizer_matches_transformers_golden():
    vocab, merges = synthetic_clip_vocab()
    tk = ClipBpeTokenizer(vocab, merges, model_max_length=CLIP_TINY["max_position_embeddings"])
    assert GOLD["texts"] == CLIP_TEXTS
    for text, want in zip(GOLD["texts"], GOLD["input_ids"]):
        assert tk.encode(text) == want, text
    ids = tk(CLIP_TEXTS)
    assert ids.dtype == torch.int32 and tuple(ids.shape) == (len(CLIP_TEXTS), 16)
    # bos ... eos then <|endoftext|> padding; truncation keeps bos + 14 + eos
    assert ids[0, 0] == vocab["<|startoftext|>"] and ids[3].tolist() == [vocab["<|startoftext|>"]] + [vocab["<|endoftext|>"]] * 15
    assert ids[4, -1] == vocab["<|endoftext|>"] and (ids[4, 1:-1] == vocab["x"]).all()


def test_tokenizer_from_files_and_live_transformers(tmp_path):
    vocab, merges = synthetic_clip_vocab()
    json.dump(vocab, open(tmp_path / "vocab.json", "w"))
    open(tmp_path / "merges.txt", "w").write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    mine = ClipBpeTokenizer.from_pretrained(str(tmp_path), model_max_length=16)
    tr = pytest.importorskip("transformers")
    ref = tr.CLIPTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))
    for text in CLIP_TEXTS + ["Hello, World's 42nd   photo-shoot...", "ÀÉÎ õü ß", "of of of the the"]:
        assert mine.encode(text) == ref(text, padding="max_length", max_length=16, truncation=True).input_ids, text


def test_oracle_matches_transformers_golden():
    from oracle.clip import ClipTextConfig as OC, clip_text_forward
    cfg = ClipTextConfig(**CLIP_TINY)
    sd = make_synthetic_clip_state_dict(cfg, seed=777)
    out = clip_text_forward(sd, OC(**CLIP_TINY), torch.tensor(GOLD["input_ids"]))
    want = torch.tensor(GOLD["last_hidden_state"])
    assert out.shape == want.shape
    assert (out - want).abs().max().item() < 2e-5          # fixture rounded to 1e-6, fp32 re-association


def test_oracle_matches_live_transformers_model():
    tr = pytest.importorskip("transformers")
    from oracle.clip import ClipTextConfig as OC, clip_text_forward
    cfg = ClipTextConfig(**CLIP_TINY)
    sd = make_synthetic_clip_state_dict(cfg, seed=5)
    m = tr.CLIPTextModel(tr.CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=1, eos_token_id=2, pad_token_id=2, **CLIP_TINY)).eval()
    prefixed = any(k.startswith("text_model.") for k in m.state_dict())
    m.load_state_dict({(k if prefixed else k[len("text_model."):]): v for k, v in sd.items()}, strict=False)
    ids = torch.randint(0, CLIP_TINY["vocab_size"], (3, 16), generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = m(ids).last_hidden_state
    assert (clip_text_forward(sd, OC(**CLIP_TINY), ids) - want).abs().max().item() < 2e-5
    # both key layouts are accepted by the loader
    bare = {k[len("text_model."):]: v for k, v in sd.items()}
    bare["embeddings.position_ids"] = torch.arange(16)[None]
    assert set(normalise_clip_state_dict(bare)) == set(sd)


def test_library_enumerates_clip_parameters():
    lib = _lib.load()
    cfg = ClipTextConfig()
    h = C.c_void_p()
    c = _lib.SdClipConfig(cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                          cfg.intermediate_size, cfg.max_position_embeddings, cfg.layer_norm_eps)
    _lib.check(lib.sd_clip_create(C.byref(c), C.byref(h)))
    n = lib.sd_unet_num_params(h)
    shapes = clip_param_shapes(cfg)
    assert n == len(shapes) == 196
    total = 0
    for i, (name, shape) in enumerate(shapes):
        buf, shp, nd = C.create_string_buffer(256), (C.c_longlong * 4)(), C.c_int()
        _lib.check(lib.sd_unet_param_info(h, i, buf, 256, shp, C.byref(nd)))
        assert buf.value.decode() == name and tuple(shp[: nd.value]) == shape
        total += int(torch.tensor(shape).prod())
    assert total == 123_060_480                      # CLIP ViT-L/14 text tower
    lib.sd_unet_destroy(h)
    bad = _lib.SdClipConfig(100, 96, 1, 4, 128, 16, 1e-5)      # hidden not a multiple of 64
    assert lib.sd_clip_create(C.byref(bad), C.byref(h)) != 0 and b"multiples of 64" in lib.sd_last_error()


def test_byte_level_tokenizer():
    tk = ClipBpeTokenizer.byte_level()
    assert len(tk.encoder) == 514 and tk.bos_token_id == 512 and tk.eos_token_id == 513
    ids = tk.encode("A cat")
    assert len(ids) == 77 and ids[0] == 512 and ids[5] == 513 and ids[-1] == 513
    sym = {v: k for k, v in tk.encoder.items()}
    assert [sym[i] for i in ids[1:5]] == ["a</w>", "c", "a", "t</w>"]
