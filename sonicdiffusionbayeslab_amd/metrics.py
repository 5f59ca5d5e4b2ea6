"""Metric plugins.  ``time_metric`` keeps the reference's definition of speed
(``src/metrics/metrics.py:115-131``): sum(loop seconds) / sum(batch_size) -> seconds per image.
``clip_score`` is the reference's parity metric (``:25-41``, ``calc_clip_score.py:13-37``); it
needs CLIP ViT-B/16 weights that only exist as a network fetch, so it is registered but raises
unless a LOCAL checkpoint directory is given.  ``BaseMethod.setup_metrics`` builds it when
``quality_metrics.clip_score.model_name_or_path`` is such a directory and ``validate`` then reports it
(``src/experiments/base_experiment.py:96-98,198-201``)."""
from __future__ import annotations

import os

import torch

from .registry import metrics_registry


@metrics_registry.add_to_registry("time_metric")
class TimeMetric:
    def __init__(self):
        self.reset()

    def update(self, time: float, batch_size: int) -> None:
        self.time += float(time)
        self.total += int(batch_size)

    def compute(self):
        return torch.tensor(self.time / self.total if self.total else float("nan"))

    def reset(self) -> None:
        self.time = 0.0
        self.total = 0


@metrics_registry.add_to_registry("clip_score")
class ClipScoreMetric:
    """100 * cos(E_img, E_txt), clamped at 0, averaged (torchmetrics 1.6.1 CLIPScore, A.8)."""

    def __init__(self, model_name_or_path: str = "openai/clip-vit-base-patch16"):
        if not os.path.isdir(str(model_name_or_path)):
            raise FileNotFoundError(
                f"CLIP checkpoint {model_name_or_path!r} is a network fetch and unavailable offline; "
                "pass a local directory to compute CLIP score")
        from transformers import CLIPModel, CLIPProcessor
        self.model = CLIPModel.from_pretrained(model_name_or_path).eval()
        self.processor = CLIPProcessor.from_pretrained(model_name_or_path)
        self.reset()

    @torch.no_grad()
    def update(self, images, text):
        inp = self.processor(text=list(text), images=[i for i in images], return_tensors="pt", padding=True, truncation=True)
        # (transformers 4.48 returns the projected embedding itself, 5.x an output object whose pooler_output is it)
        emb = lambda o: o if torch.is_tensor(o) else o.pooler_output
        img = emb(self.model.get_image_features(pixel_values=inp["pixel_values"]))
        txt = emb(self.model.get_text_features(input_ids=inp["input_ids"], attention_mask=inp["attention_mask"]))
        img = img / img.norm(p=2, dim=-1, keepdim=True)
        txt = txt / txt.norm(p=2, dim=-1, keepdim=True)
        self.score += (100 * (img * txt).sum(-1)).clamp(min=0).sum().item()
        self.n += len(text)

    def compute(self):
        return torch.tensor(self.score / max(self.n, 1))

    def reset(self):
        self.score, self.n = 0.0, 0
