"""The four plugin registries the reference exposes from ``src/registry.py:3-6``; plugins decorate themselves
into these (``@models_registry.add_to_registry("stable_diffusion_model")`` ...) and the harness looks them up by
the names the YAML files carry."""
from .utils.class_registry import ClassRegistry

models_registry, methods_registry, metrics_registry, schedulers_registry = (
    ClassRegistry(kind) for kind in ("model", "method", "metric", "scheduler"))

__all__ = ["models_registry", "methods_registry", "metrics_registry", "schedulers_registry"]
