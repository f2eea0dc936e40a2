"""Generic State / Model / ModelBuilder / Initializer bases.
Mirrors /root/reference/modules/nclaw/sim/abstract.py:7-116 (torch devices instead of Warp devices)."""
from collections import OrderedDict
from typing import Any, Optional

import torch


def _device(device) -> torch.device:
    if device is None:
        return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if isinstance(device, torch.device):
        return device
    s = str(device)
    return torch.device(s)


class State(object):
    def __init__(self, shape: Any, device=None, requires_grad: bool = False) -> None:
        self.shape = shape
        self.device = _device(device)
        self.requires_grad = requires_grad

    def to_torch(self):
        raise NotImplementedError

    def to_torch_grad(self):
        raise NotImplementedError

    def from_torch(self):
        raise NotImplementedError

    def from_torch_grad(self):
        raise NotImplementedError


class Model(object):
    ConstantType = Any
    StaticsType = Any
    StateType = State

    def __init__(self, constant, device=None, requires_grad: int = False) -> None:
        self.constant = constant
        self.device = _device(device)
        self.requires_grad = requires_grad

    def state(self, shape: Any, requires_grad: Optional[bool] = None):
        if requires_grad is None:
            requires_grad = self.requires_grad
        return self.StateType(shape=shape, device=self.device, requires_grad=requires_grad)

    def statics(self, shape: Any):
        statics = self.StaticsType()
        statics.init(shape=shape, device=self.device)
        return statics


class ModelBuilder(object):
    ConstantType = Any
    StateType = State
    ModelType = Model

    def __init__(self) -> None:
        self.config = OrderedDict()
        for name in self.ConstantType.__annotations__.keys():
            self.reserve(name)

    def reserve(self, name: str, init: Optional[Any] = None) -> None:
        if name in self.config:
            raise RuntimeError(f'duplicated key ({name}) reserved in ModelBuilder')
        self.config[name] = init

    @property
    def ready(self) -> bool:
        return all(v is not None for v in self.config.values())

    def build_constant(self):
        return self.ConstantType()

    def finalize(self, device=None, requires_grad: bool = False):
        if not self.ready:
            raise RuntimeError(f'config uninitialized: {self.config}')
        constant = self.build_constant()
        return self.ModelType(constant, device, requires_grad)


class StateInitializer(object):
    StateType = State
    ModelType = Model

    def __init__(self, model) -> None:
        self.model = model

    def finalize(self, shape: Any, requires_grad: bool = False):
        return self.model.state(shape=shape, requires_grad=requires_grad)


class StaticsInitializer(object):
    StaticsType = Any
    ModelType = Model

    def __init__(self, model) -> None:
        self.model = model

    def update(self, statics, step: int = 0) -> None:
        raise NotImplementedError

    def finalize(self, shape: Any):
        return self.model.statics(shape)
