"""torch <-> engine boundary of the simulator.

One substep is an autograd.Function (MPMSimFunction); the nn.Module front-ends differ only in where the two state
buffers of a substep come from:

    MPMDiffSim(model)(statics, x, v, C, F, stress)              fresh buffers every call
    MPMCacheDiffSim(model, num_steps)(statics, step, x, ...)    one pair of buffers per time step, reused across epochs
    MPMForwardSim(model)(statics, state)                        no autograd, in place on `state`
    MPMExtraSim(model)(statics, state, statics_extra, state_extra)   g2p onto a passive particle set

Names, argument order, return arity and the nan_to_num_ of every returned gradient follow
/root/reference/modules/nclaw/sim/interface.py:12-147.  Two things are added underneath without changing that surface:
shuffled particle sets are stepped on an internally Hilbert-sorted copy (order.py), and the `tape` slot of
MPMModel.forward/backward carries a grid cache record (mpm.GridTape) instead of a Warp tape.
"""
from typing import List, Optional

import torch
import torch.autograd as autograd
import torch.nn as nn
from torch import Tensor

from .mpm import MPMModel, MPMState, MPMStatics
from .order import ParticleOrder

_FIELDS = ("x", "v", "C", "F", "stress")


class MPMSimFunction(autograd.Function):
    """(x, v, C, F, stress) at t  ->  (x, v, C, F) at t + dt, differentiable w.r.t. all five inputs."""

    @staticmethod
    def forward(ctx, model: MPMModel, statics: MPMStatics, state_curr: MPMState, state_next: MPMState, *particle: Tensor):
        # a grid cache record rides in the reference's `tape` position when a backward pass can follow
        ctx.tape = model.new_tape() if any(ctx.needs_input_grad) else None
        ctx.engine = (model, statics, state_curr, state_next)
        state_curr.from_torch(**dict(zip(_FIELDS, particle)))
        model.forward(statics, state_curr, state_next, ctx.tape)
        model._size_cache()
        return tuple(state_next.to_torch()[:4])

    @staticmethod
    def backward(ctx, *grad_next: Tensor):
        model, statics, state_curr, state_next = ctx.engine
        state_next.from_torch_grad(**{"grad_" + n: g for n, g in zip(_FIELDS[:4], grad_next)})
        model.backward(statics, state_curr, state_next, ctx.tape)
        grads = state_curr.to_torch_grad()
        for g in grads:
            if g is not None:
                torch.nan_to_num_(g, 0.0, 0.0, 0.0)          # non-finite gradients become zeros, as in the reference
        return (None, None, None, None) + tuple(grads)


class MPMSim(nn.Module):
    """Common part of the front-ends: the model, the particle-order helper, and `state(...)` (interface.py:79-93)."""

    def __init__(self, model: MPMModel, reorder="auto") -> None:
        super().__init__()
        self.model = model
        # not in the reference: shuffled particle sets (what prepare_simulation_data produces) are run on an internally
        # Hilbert-sorted copy, see order.py; reorder=False switches it off
        self.order = ParticleOrder(int(model.constant.num_grids), reorder)

    def state(self, x: Tensor, v: Tensor, C: Tensor, F: Tensor, stress: Tensor, state: Optional[MPMState] = None) -> MPMState:
        target = state if state is not None else self.model.state(x.size(0))
        target.from_torch(x=x, v=v, C=C, F=F, stress=stress)
        return target

    def _step(self, statics: MPMStatics, buffers, particle):
        """Differentiable substep on `buffers` = (state_curr, state_next); re-ordered if the particle order is poor."""
        if not self.order.active(particle[0]):
            return MPMSimFunction.apply(self.model, statics, *buffers, *particle)
        fwd, back = self.order.perm, self.order.inv
        out = MPMSimFunction.apply(self.model, self.order.statics(statics), *buffers, *(t[fwd] for t in particle))
        return tuple(t[back] for t in out)


class MPMDiffSim(MPMSim):
    """interface.py:96-105"""

    def forward(self, statics: MPMStatics, x: Tensor, v: Tensor, C: Tensor, F: Tensor, stress: Tensor):
        n = x.size(0)
        return self._step(statics, (self.model.state(n), self.model.state(n)), (x, v, C, F, stress))


class MPMCacheDiffSim(MPMSim):
    """interface.py:108-123: the state buffers of time step `step` are allocated once and reused."""

    def __init__(self, model: MPMModel, num_steps: int, reorder="auto") -> None:
        super().__init__(model, reorder)
        self.curr_states: List[Optional[MPMState]] = [None] * num_steps
        self.next_states: List[Optional[MPMState]] = [None] * num_steps

    def forward(self, statics: MPMStatics, step: int, x: Tensor, v: Tensor, C: Tensor, F: Tensor, stress: Tensor):
        for slots in (self.curr_states, self.next_states):
            if slots[step] is None:
                slots[step] = self.model.state(x.size(0))
        return self._step(statics, (self.curr_states[step], self.next_states[step]), (x, v, C, F, stress))


class MPMForwardSim(MPMSim):
    """interface.py:126-135 (in place: state is both current and next)"""

    def forward(self, statics: MPMStatics, state: MPMState):
        caller = state.to_torch()
        if self.order.active(caller[0]):
            # shuffled particles: step an internally sorted copy and write the result back into the caller's buffers
            scratch = getattr(self, "_sorted_state", None)
            if scratch is None or scratch.particle.x.shape[0] != caller[0].shape[0]:
                scratch = self._sorted_state = self.model.state(caller[0].shape[0])
            for dst, src in zip(scratch.to_torch(), caller):
                torch.index_select(src.detach(), 0, self.order.perm, out=dst)
            self.model.forward(self.order.statics(statics), scratch, scratch, None)
            for dst, src in zip(caller[:4], scratch.to_torch()[:4]):
                torch.index_select(src, 0, self.order.inv, out=dst.detach())
        else:
            self.model.forward(statics, state, state, None)
        return tuple(state.to_torch()[:4])


class MPMExtraSim(MPMSim):
    """interface.py:138-147"""

    def forward(self, statics: MPMStatics, state: MPMState, statics_extra: MPMStatics, state_extra: MPMState) -> Tensor:
        self.model.forward_extra(statics, state, statics_extra, state_extra)
        return state_extra.to_torch()[0]
