"""The four plugin registries of the reference (``src/registry.py:3-6``)."""
from .utils.class_registry import ClassRegistry

models_registry = ClassRegistry()
methods_registry = ClassRegistry()
metrics_registry = ClassRegistry()
schedulers_registry = ClassRegistry()
