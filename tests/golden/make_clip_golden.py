"""Writes tests/golden/clip_golden.json from the transformers build in this image (third-party library, NOT the
reference): (1) token ids of ``transformers.CLIPTokenizer`` on the synthetic vocabulary of tests/util.py,
(2) ``CLIPTextModel(...).last_hidden_state`` of a seeded tiny model (weights =
``make_synthetic_clip_state_dict(CLIP_TINY, seed=777)``) on those ids.  These pin ``ClipBpeTokenizer`` and
``oracle/clip.py``.  Run from the repo root: ``python tests/golden/make_clip_golden.py``."""
import json
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from sonicdiffusionbayeslab_amd.clip import ClipTextConfig, make_synthetic_clip_state_dict  # noqa: E402
from tests.util import CLIP_TEXTS, CLIP_TINY, synthetic_clip_vocab  # noqa: E402


def main():
    import transformers
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    vocab, merges = synthetic_clip_vocab()
    L = CLIP_TINY["max_position_embeddings"]
    with tempfile.TemporaryDirectory() as d:
        json.dump(vocab, open(os.path.join(d, "vocab.json"), "w"))
        open(os.path.join(d, "merges.txt"), "w").write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
        tk = CLIPTokenizer(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))
        ids = [tk(t, padding="max_length", max_length=L, truncation=True).input_ids for t in CLIP_TEXTS]
    cfg = ClipTextConfig(**CLIP_TINY)
    sd = make_synthetic_clip_state_dict(cfg, seed=777)
    tcfg = CLIPTextConfig(hidden_act="quick_gelu", bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"],
                          pad_token_id=vocab["<|endoftext|>"], **CLIP_TINY)
    m = CLIPTextModel(tcfg).eval()
    own = m.state_dict()
    prefixed = any(k.startswith("text_model.") for k in own)
    missing = m.load_state_dict({(k if prefixed else k[len("text_model."):]): v for k, v in sd.items()}, strict=False)
    assert not [k for k in missing.missing_keys if "position_ids" not in k], missing
    with torch.no_grad():
        out = m(torch.tensor(ids)).last_hidden_state
    res = {"transformers_version": transformers.__version__, "texts": CLIP_TEXTS, "input_ids": ids,
           "weights": "make_synthetic_clip_state_dict(ClipTextConfig(**CLIP_TINY), seed=777)",
           "last_hidden_state": [[[round(float(v), 6) for v in row] for row in b] for b in out]}
    path = os.path.join(ROOT, "tests", "golden", "clip_golden.json")
    json.dump(res, open(path, "w"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
