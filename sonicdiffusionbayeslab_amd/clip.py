"""CLIP text encoder + tokenizer -- SURVEY.md §8f "next" row 2.

Replaces ``encode_prompt`` of the reference pipeline (``src/models.py:139-155``): tokenise the prompts
(CLIP byte-level BPE, 77 tokens, padded with ``<|endoftext|>``), run ``CLIPTextModel`` (ViT-L/14 text tower:
12 pre-LN layers, 12 heads of 64, quick_gelu MLP, causal mask, final LayerNorm) and hand its
``last_hidden_state`` ``[B,77,768]`` to the UNet as ``prompt_embeds``.

* ``HipClipTextModel``   -- the transformer on libsdhip (``sd_clip_create`` / ``sd_clip_encode``); no CPU fallback.
* ``ClipBpeTokenizer``   -- host-side text processing (the reference does it in Python too): the algorithm of
  transformers' ``CLIPTokenizer`` without ``ftfy`` [upstream-recall], reading the checkpoint's ``vocab.json`` and
  ``merges.txt``.
* ``ClipPromptEncoder``  -- ``prompts -> [B,77,768]``: the callable a pipeline takes as ``text_encoder``.

The CLIP checkpoint is a network fetch in the reference; offline this module is exercised on
seeded synthetic weights / a synthetic vocabulary, and picks up a local ``tokenizer/`` + ``text_encoder/``
directory when one is given (``SD_AMD_MODEL_DIR``).
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
from dataclasses import dataclass
from functools import lru_cache
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib


@dataclass
class ClipTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5
    bos_token_id: int = 49406
    eos_token_id: int = 49407
    pad_token_id: int = 49407          # SD-1.5's tokenizer pads with <|endoftext|>


def clip_param_shapes(cfg: ClipTextConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """transformers ``CLIPTextModel`` state_dict names (4.48.0: ``text_model.`` prefix) in module order."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    out: List[Tuple[str, Tuple[int, ...]]] = []
    add = lambda n, s: out.append((n, tuple(s)))
    add("text_model.embeddings.token_embedding.weight", (cfg.vocab_size, H))
    add("text_model.embeddings.position_embedding.weight", (cfg.max_position_embeddings, H))
    for i in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            add(p + f"self_attn.{n}.weight", (H, H)); add(p + f"self_attn.{n}.bias", (H,))
        add(p + "layer_norm1.weight", (H,)); add(p + "layer_norm1.bias", (H,))
        add(p + "mlp.fc1.weight", (I, H)); add(p + "mlp.fc1.bias", (I,))
        add(p + "mlp.fc2.weight", (H, I)); add(p + "mlp.fc2.bias", (H,))
        add(p + "layer_norm2.weight", (H,)); add(p + "layer_norm2.bias", (H,))
    add("text_model.final_layer_norm.weight", (H,)); add("text_model.final_layer_norm.bias", (H,))
    return out


def make_synthetic_clip_state_dict(cfg: ClipTextConfig, seed: int = 777) -> Dict[str, torch.Tensor]:
    """Seeded CLIP-text-shaped weights on the bf16 grid (no CLIP weights exist offline)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in clip_param_shapes(cfg):
        if "embedding" in name:
            t = torch.randn(shape, generator=g) * 0.02 * (1.0 if "token" in name else 0.5)
        elif name.endswith(".bias"):
            t = torch.randn(shape, generator=g) * 0.02
        elif "layer_norm" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) / math.sqrt(shape[1])
        sd[name] = t.to(torch.bfloat16).float()
    return sd


def normalise_clip_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept both key layouts: ``text_model.*`` (transformers 4.x checkpoints) and bare ``embeddings.*`` /
    ``encoder.*`` (what a 5.x ``CLIPTextModel.state_dict()`` yields); drop the ``position_ids`` buffer."""
    out = {}
    for k, v in sd.items():
        if k.endswith("position_ids"):
            continue
        out[k if k.startswith("text_model.") else "text_model." + k] = v
    return out


def load_clip_state_dict(model_dir: str) -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    for cand in (os.path.join(model_dir, "text_encoder", "model.safetensors"), os.path.join(model_dir, "model.safetensors")):
        if os.path.isfile(cand):
            return normalise_clip_state_dict({k: v.float() for k, v in load_file(cand).items()})
    raise FileNotFoundError(f"no local CLIP text-encoder weights under {model_dir!r}")


class HipClipTextModel:
    """``text_encoder(input_ids)[0]`` replacement: int token ids ``[B,77]`` -> fp32 ``[B,77,768]`` on the GPU."""

    def __init__(self, config: ClipTextConfig, state_dict: Dict[str, torch.Tensor], device: str = "cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.SdHipError("HipClipTextModel needs an MI355X (no CPU fallback exists)")
        self.config = config
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        torch.cuda.set_device(self.device)
        c = _lib.SdClipConfig(config.vocab_size, config.hidden_size, config.num_hidden_layers, config.num_attention_heads,
                              config.intermediate_size, config.max_position_embeddings, config.layer_norm_eps)
        _lib.check(self._lib.sd_clip_create(C.byref(c), C.byref(self._handle)), "sd_clip_create")
        sd = normalise_clip_state_dict(state_dict)
        for name, shape in clip_param_shapes(config):
            if name not in sd:
                raise KeyError(f"state_dict lacks CLIP parameter {name!r}")
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
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

    def encode(self, input_ids: torch.Tensor) -> torch.Tensor:
        L, H = self.config.max_position_embeddings, self.config.hidden_size
        ids = input_ids.to(self.device, torch.int32).contiguous()
        if ids.dim() != 2 or ids.shape[1] != L:
            raise ValueError(f"input_ids must be [B,{L}], got {tuple(ids.shape)}")
        if int(ids.min()) < 0 or int(ids.max()) >= self.config.vocab_size:
            raise ValueError("token id outside the vocabulary")
        b = ids.shape[0]
        out = torch.empty((b, L, H), dtype=torch.float32, device=self.device)
        ws = self._workspace(b)
        wsp = (ws.data_ptr() + 255) // 256 * 256
        _lib.check(self._lib.sd_clip_encode(self._handle, _lib.current_stream(), ids.data_ptr(), b, out.data_ptr(), wsp,
                                            ws.numel() - 256), "sd_clip_encode")
        return out

    __call__ = encode


# --------------------------------------------------------------------------------------------------
# CLIP byte-level BPE (transformers CLIPTokenizer without ftfy) -- host-side text processing
# --------------------------------------------------------------------------------------------------
@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    """The reversible byte -> printable-unicode table of GPT-2/CLIP BPE."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def _basic_clean(text: str) -> str:
    """What transformers' BasicTokenizer(strip_accents=False, do_split_on_punc=False) + whitespace_clean leave:
    control characters dropped, every whitespace run collapsed to one space, stripped."""
    import unicodedata
    out = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD:
            continue
        cat = unicodedata.category(ch)
        if ch in ("\t", "\n", "\r") or cat == "Zs":
            out.append(" ")
        elif cat.startswith("C"):
            continue
        else:
            out.append(ch)
    return " ".join("".join(out).split())


class ClipBpeTokenizer:
    PAT = r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+"

    def __init__(self, vocab: Dict[str, int], merges: List[Tuple[str, str]], model_max_length: int = 77,
                 bos_token: str = "<|startoftext|>", eos_token: str = "<|endoftext|>", pad_token: str = "<|endoftext|>"):
        import regex
        self.encoder = dict(vocab)
        self.bpe_ranks = {tuple(m): i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.pat = regex.compile(self.PAT, regex.IGNORECASE)
        self.model_max_length = model_max_length
        self.bos_token_id, self.eos_token_id = self.encoder[bos_token], self.encoder[eos_token]
        self.pad_token_id = self.encoder[pad_token]
        self.unk_token_id = self.encoder.get(eos_token)      # CLIP's unk token is <|endoftext|>
        self.cache = {bos_token: bos_token, eos_token: eos_token}

    @classmethod
    def from_pretrained(cls, tokenizer_dir: str, **kw) -> "ClipBpeTokenizer":
        with open(os.path.join(tokenizer_dir, "vocab.json"), encoding="utf-8") as f:
            vocab = json.load(f)
        with open(os.path.join(tokenizer_dir, "merges.txt"), encoding="utf-8") as f:
            lines = f.read().strip().split("\n")
        if lines and lines[0].startswith("#version"):
            lines = lines[1:]
        merges = [tuple(l.split()) for l in lines if l and len(l.split()) == 2]
        return cls(vocab, merges, **kw)

    @classmethod
    def byte_level(cls, **kw) -> "ClipBpeTokenizer":
        """A merge-free byte-level vocabulary (every byte symbol, its ``</w>`` form, the two special tokens:
        514 ids) -- what the benchmarks tokenize with when no checkpoint vocabulary exists offline."""
        syms = list(bytes_to_unicode().values())
        vocab = {s: i for i, s in enumerate(syms)}
        vocab.update({s + "</w>": len(syms) + i for i, s in enumerate(syms)})
        vocab["<|startoftext|>"] = len(vocab)
        vocab["<|endoftext|>"] = len(vocab)
        return cls(vocab, [], **kw)

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = set(zip(word[:-1], word[1:]))
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new.extend(word[i:])
                    break
                new.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
            if len(word) == 1:
                break
            pairs = set(zip(word[:-1], word[1:]))
        out = " ".join(word)
        self.cache[token] = out
        return out

    def tokenize(self, text: str) -> List[str]:
        import regex
        text = _basic_clean(text).lower()
        toks: List[str] = []
        for tok in regex.findall(self.pat, text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            toks.extend(self.bpe(tok).split(" "))
        return toks

    def encode(self, text: str) -> List[int]:
        """``tokenizer(text, padding="max_length", max_length=77, truncation=True).input_ids``."""
        ids = [self.encoder.get(t, self.unk_token_id) for t in self.tokenize(text)]
        ids = [self.bos_token_id] + ids[: self.model_max_length - 2] + [self.eos_token_id]
        return ids + [self.pad_token_id] * (self.model_max_length - len(ids))

    def __call__(self, prompts: List[str]) -> torch.Tensor:
        return torch.tensor([self.encode(p) for p in prompts], dtype=torch.int32)


class ClipPromptEncoder:
    """``prompts -> prompt_embeds`` (tokenise + ``text_encoder(ids)[0]``, src/models.py:139-150): what
    ``StableDiffusionModel(text_encoder=...)`` expects."""

    def __init__(self, tokenizer: ClipBpeTokenizer, text_model: HipClipTextModel):
        self.tokenizer, self.text_model = tokenizer, text_model

    @classmethod
    def from_pretrained(cls, model_dir: str, device: str = "cuda:0") -> "ClipPromptEncoder":
        """A local diffusers checkpoint directory with ``tokenizer/`` and ``text_encoder/``."""
        tok = ClipBpeTokenizer.from_pretrained(os.path.join(model_dir, "tokenizer"))
        cfg_path = os.path.join(model_dir, "text_encoder", "config.json")
        cfg = ClipTextConfig()
        if os.path.isfile(cfg_path):
            with open(cfg_path) as f:
                j = json.load(f)
            cfg = ClipTextConfig(**{k: j[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers",
                                                       "num_attention_heads", "intermediate_size",
                                                       "max_position_embeddings", "layer_norm_eps") if k in j})
        return cls(tok, HipClipTextModel(cfg, load_clip_state_dict(model_dir), device=device))

    def __call__(self, prompts: List[str]) -> torch.Tensor:
        return self.text_model.encode(self.tokenizer(list(prompts)))
