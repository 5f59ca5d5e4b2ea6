"""Importing this package registers every method plugin (``src/experiments/__init__.py``)."""
import importlib

for _name in ("base_experiment", "ddim", "dpm_solver", "consistency_model", "deep_cache", "default_sd",
              "two_schedulers", "interliving_exp", "skip_steps_exp"):
    importlib.import_module(f"{__name__}.{_name}")

from .base_experiment import BaseMethod  # noqa: E402,F401
