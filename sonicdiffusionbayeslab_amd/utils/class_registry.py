"""Name -> class plugin registry with the reference's surface.

Mirrors ``src/utils/class_registry.py:8-68`` of the reference: ``ClassRegistry()``,
``registry[name]`` (``:14-15``), ``@registry.add_to_registry(name, arg_keys=None,
stop_args=("self", "args", "kwargs"))`` (``:58-68``) which also records a dataclass built from
the class' ``__init__`` signature in ``registry.args[name]`` (``:17-44``; never consumed by the
reference, kept for drop-in compatibility).  omegaconf is not available offline, so its
``MISSING`` sentinel is the literal ``"???"`` it stands for.
"""
import dataclasses
import inspect
import typing

MISSING = "???"


class ClassRegistry:
    def __init__(self):
        self.classes = dict()
        self.args = dict()
        self.arg_keys = None

    def __getitem__(self, item):
        return self.classes[item]

    def __contains__(self, item):
        return item in self.classes

    def keys(self):
        return self.classes.keys()

    def make_dataclass_from_init(self, func, name, arg_keys, stop_args):
        fields = []
        for k, v in inspect.signature(func).parameters.items():
            if k in stop_args or v.kind in (v.VAR_POSITIONAL, v.VAR_KEYWORD):
                continue
            if v.default is inspect.Parameter.empty:
                fields.append((k, typing.Any, MISSING))
            elif v.default is None:
                fields.append((k, typing.Optional[typing.Any], None))
            elif isinstance(v.default, (list, dict, set)):
                fields.append((k, type(v.default), dataclasses.field(default_factory=lambda d=v.default: type(d)(d))))
            else:
                fields.append((k, type(v.default), dataclasses.field(default=v.default)))
        if arg_keys:
            self.arg_keys = arg_keys
            arg_classes = {key: dataclasses.make_dataclass(key, fields) for key in arg_keys}
            return dataclasses.make_dataclass(
                name, [(k, v, dataclasses.field(default_factory=v)) for k, v in arg_classes.items()])
        return dataclasses.make_dataclass(name, fields)

    def make_dataclass_from_classes(self, name):
        return dataclasses.make_dataclass(
            name, [(k, v, dataclasses.field(default_factory=v)) for k, v in self.classes.items()])

    def make_dataclass_from_args(self, name):
        return dataclasses.make_dataclass(
            name, [(k, v, dataclasses.field(default_factory=v)) for k, v in self.args.items()])

    def add_to_registry(self, name, arg_keys=None, stop_args=("self", "args", "kwargs")):
        def add_class_by_name(cls):
            self.classes[name] = cls
            self.args[name] = self.make_dataclass_from_init(cls.__init__, name.replace(" ", "_"), arg_keys, stop_args)
            return cls

        return add_class_by_name
