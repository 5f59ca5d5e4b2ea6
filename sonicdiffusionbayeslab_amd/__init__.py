"""sonicdiffusionbayeslab_amd -- MI355X-native Stable Diffusion sampling hot path behind the
plugin surface of Kotstantinovskiy/SonicDiffusionBayesLab (registries, scheduler / pipeline
protocol, configs/*.yaml entry points).  Importing the package registers every plugin, like the
reference's ``src/__init__.py:1-5``; it does NOT load libsdhip (that happens at first use and
fails loudly if the library is not built)."""
from . import experiments, metrics, models, schedulers  # noqa: F401  (registration side effects)
from .registry import methods_registry, metrics_registry, models_registry, schedulers_registry  # noqa: F401

__all__ = ["methods_registry", "metrics_registry", "models_registry", "schedulers_registry"]
