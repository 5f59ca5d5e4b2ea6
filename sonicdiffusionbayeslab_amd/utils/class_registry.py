"""Name -> class plugin registry with the reference's public surface.

What callers of ``src/utils/class_registry.py:8-68`` rely on and therefore what is kept: ``ClassRegistry()``,
``registry[name]`` (``:14-15``), the decorator ``@registry.add_to_registry(name, arg_keys=None,
stop_args=("self", "args", "kwargs"))`` (``:58-68``), and the attributes ``classes`` / ``args`` /
``arg_keys`` with the three ``make_dataclass_from_*`` helpers (``:17-56``) that turn ``__init__`` signatures into
config dataclasses (never consumed by the reference itself).  omegaconf is not available offline, so its ``MISSING``
sentinel is the literal ``"???"`` it stands for.
"""
from __future__ import annotations

import dataclasses as dc
import inspect
from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple

MISSING = "???"
_Field = Tuple[str, Any, Any]


def _signature_fields(func: Callable, skip: Iterable[str]) -> List[_Field]:
    """(name, type, default) triples for the keyword parameters of ``func``: required -> ``MISSING``,
    ``None`` defaults -> ``Optional[Any]``, anything else typed after its default."""
    out: List[_Field] = []
    for name, par in inspect.signature(func).parameters.items():
        if name in skip or par.kind in (par.VAR_POSITIONAL, par.VAR_KEYWORD):
            continue
        default = par.default
        if default is inspect.Parameter.empty:
            out.append((name, Any, MISSING))
        elif default is None:
            out.append((name, Optional[Any], None))
        elif isinstance(default, (list, dict, set)):                 # mutable: dataclasses need a factory
            out.append((name, type(default), dc.field(default_factory=lambda d=default: type(d)(d))))
        else:
            out.append((name, type(default), dc.field(default=default)))
    return out


def _bundle(name: str, members: Dict[str, type]):
    """A dataclass whose fields are default-constructed instances of ``members``."""
    return dc.make_dataclass(name, [(k, cls, dc.field(default_factory=cls)) for k, cls in members.items()])


class ClassRegistry:
    def __init__(self, kind: str = "plugin"):
        self.kind = kind
        self.classes: Dict[str, type] = {}
        self.args: Dict[str, type] = {}
        self.arg_keys = None

    # lookup ---------------------------------------------------------------------------------------
    def __getitem__(self, item):
        try:
            return self.classes[item]
        except KeyError:
            raise KeyError(f"no {self.kind} registered as {item!r}; known: {sorted(self.classes)}") from None

    def __contains__(self, item):
        return item in self.classes

    def keys(self):
        return self.classes.keys()

    # dataclass helpers ----------------------------------------------------------------------------
    def make_dataclass_from_init(self, func, name, arg_keys, stop_args):
        fields = _signature_fields(func, stop_args)
        if not arg_keys:
            return dc.make_dataclass(name, fields)
        self.arg_keys = arg_keys
        return _bundle(name, {key: dc.make_dataclass(key, fields) for key in arg_keys})

    def make_dataclass_from_classes(self, name):
        return _bundle(name, self.classes)

    def make_dataclass_from_args(self, name):
        return _bundle(name, self.args)

    # registration ---------------------------------------------------------------------------------
    def add_to_registry(self, name, arg_keys=None, stop_args=("self", "args", "kwargs")):
        def register(cls):
            self.classes[name] = cls
            self.args[name] = self.make_dataclass_from_init(cls.__init__, name.replace(" ", "_"), arg_keys, stop_args)
            return cls

        return register
