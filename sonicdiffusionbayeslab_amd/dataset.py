"""Prompt source of the harness (``src/dataset/dataset.py:8-41``).  The reference lists an image
directory and looks each file's prompt up in a JSON dict; the image directory is not part of the
repository, so when it is absent the prompts are taken in file order --
the hot path only consumes the prompt strings."""
from __future__ import annotations

import json
import os


class PromptDataset:
    def __init__(self, image_dir: str, prompts_file: str):
        with open(prompts_file, "r") as f:
            self.prompts_json = json.load(f)
        if image_dir and os.path.isdir(image_dir):
            self.image_files = [f for f in os.listdir(image_dir) if os.path.isfile(os.path.join(image_dir, f))
                                and f in self.prompts_json]
        else:
            self.image_files = list(self.prompts_json.keys())

    def __len__(self):
        return len(self.image_files)

    def __getitem__(self, idx):
        f = self.image_files[idx]
        return {"image_file": f, "prompt": self.prompts_json[f]}

    def batches(self, batch_size: int):
        """DataLoader(batch_size, shuffle=False) equivalent."""
        for s in range(0, len(self), batch_size):
            items = [self[i] for i in range(s, min(len(self), s + batch_size))]
            yield {"image_file": [it["image_file"] for it in items], "prompt": [it["prompt"] for it in items]}
