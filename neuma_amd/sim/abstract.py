"""Base classes of the simulator front-end: State, Model, ModelBuilder, StateInitializer, StaticsInitializer.

Same public surface as /root/reference/modules/nclaw/sim/abstract.py:7-116 (class names, class-level `*Type` hooks,
`model.state(shape)`, `model.statics(shape)`, `builder.reserve / ready / finalize`, the two RuntimeErrors), on torch
devices instead of Warp devices.  The concrete MPM classes live in mpm.py.
"""
from collections import OrderedDict
from typing import Any, Optional

import torch


def resolve_device(device=None) -> torch.device:
    """None -> the current GPU (CPU if there is none); strings and torch.device pass through."""
    if isinstance(device, torch.device):
        return device
    if device is not None:
        return torch.device(str(device))
    if torch.cuda.is_available():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _abstract(name: str):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__}.{name}")
    method.__name__ = name
    return method


class _OnDevice(object):
    """Anything that lives on one device and knows whether it tracks gradients."""

    def __init__(self, device=None, requires_grad: bool = False) -> None:
        self.device = resolve_device(device)
        self.requires_grad = requires_grad


class State(_OnDevice):
    """Per-particle simulation state; subclasses exchange it with torch through the four converters below."""

    def __init__(self, shape: Any, device=None, requires_grad: bool = False) -> None:
        _OnDevice.__init__(self, device, requires_grad)
        self.shape = shape

    to_torch = _abstract("to_torch")
    to_torch_grad = _abstract("to_torch_grad")
    from_torch = _abstract("from_torch")
    from_torch_grad = _abstract("from_torch_grad")


class Model(_OnDevice):
    """Owns the constants of a simulation and creates matching states / statics on its device."""

    ConstantType = Any
    StaticsType = Any
    StateType = State

    def __init__(self, constant, device=None, requires_grad: int = False) -> None:
        _OnDevice.__init__(self, device, requires_grad)
        self.constant = constant

    def state(self, shape: Any, requires_grad: Optional[bool] = None):
        track = self.requires_grad if requires_grad is None else requires_grad
        return self.StateType(shape=shape, device=self.device, requires_grad=track)

    def statics(self, shape: Any):
        out = self.StaticsType()
        out.init(shape=shape, device=self.device)
        return out


class ModelBuilder(object):
    """Collects one value per annotated field of `ConstantType`, then builds the model."""

    ConstantType = Any
    StateType = State
    ModelType = Model

    def __init__(self) -> None:
        self.config = OrderedDict()
        for field in getattr(self.ConstantType, "__annotations__", {}):
            self.reserve(field)

    def reserve(self, name: str, init: Optional[Any] = None) -> None:
        if name in self.config:
            raise RuntimeError(f'duplicated key ({name}) reserved in ModelBuilder')
        self.config[name] = init

    @property
    def ready(self) -> bool:
        return not any(value is None for value in self.config.values())

    def build_constant(self):
        return self.ConstantType()

    def finalize(self, device=None, requires_grad: bool = False):
        if not self.ready:
            raise RuntimeError(f'config uninitialized: {self.config}')
        return self.ModelType(self.build_constant(), device, requires_grad)


class _ModelBound(object):
    ModelType = Model

    def __init__(self, model) -> None:
        self.model = model


class StateInitializer(_ModelBound):
    StateType = State

    def finalize(self, shape: Any, requires_grad: bool = False):
        return self.model.state(shape=shape, requires_grad=requires_grad)


class StaticsInitializer(_ModelBound):
    StaticsType = Any

    update = _abstract("update")

    def finalize(self, shape: Any):
        return self.model.statics(shape)
