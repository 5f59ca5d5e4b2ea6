"""YAML experiment configs with the attribute / ``.get`` access the reference uses through
OmegaConf (``main.py:11``; ``src/experiments/base_experiment.py:56-62,137``).  omegaconf is not
installed offline, so this is a small PyYAML-backed stand-in with the same read semantics:
attribute access raises on a missing key, ``.get(key, default)`` does not."""
from __future__ import annotations

import os

import yaml


class ConfigNode(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(f"Missing key {key}") from None

    def __setattr__(self, key, value):
        self[key] = _wrap(value)

    def to_container(self):
        return {k: (v.to_container() if isinstance(v, ConfigNode) else v) for k, v in self.items()}


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, ConfigNode):
        return ConfigNode({k: _wrap(x) for k, x in v.items()})
    return v


def load_config(path: str) -> ConfigNode:
    with open(path, "r") as f:
        return _wrap(yaml.safe_load(f))


def load_named_config(config_file: str, config_dir: str = "./configs") -> ConfigNode:
    """``OmegaConf.load(os.path.join("./configs", config_file))`` of ``main.py:11``."""
    return load_config(os.path.join(config_dir, config_file))
