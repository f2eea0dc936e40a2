"""torch <-> engine boundary of the simulator: one substep as an autograd.Function and the nn.Module front-ends.
Mirrors /root/reference/modules/nclaw/sim/interface.py:12-147 (same names, argument order, return arity,
nan_to_num_ on every returned gradient)."""
from typing import Optional

import torch
import torch.autograd as autograd
import torch.nn as nn
from torch import Tensor

from .mpm import MPMModel, MPMState, MPMStatics
from .order import ParticleOrder


class MPMSimFunction(autograd.Function):
    """interface.py:12-76"""

    @staticmethod
    def forward(ctx, model: MPMModel, statics: MPMStatics, state_curr: MPMState, state_next: MPMState,
                x: Tensor, v: Tensor, C: Tensor, F: Tensor, stress: Tensor):
        # the reference passes a Warp tape here; ours carries the substep's grid cache record when a backward pass can follow
        tape = model.new_tape() if any(ctx.needs_input_grad) else None
        state_curr.from_torch(x=x, v=v, C=C, F=F, stress=stress)
        model.forward(statics, state_curr, state_next, tape)
        model._size_cache()
        x_next, v_next, C_next, F_next, _ = state_next.to_torch()
        ctx.model = model
        ctx.tape = tape
        ctx.statics = statics
        ctx.state_curr = state_curr
        ctx.state_next = state_next
        return x_next, v_next, C_next, F_next

    @staticmethod
    def backward(ctx, grad_x_next: Tensor, grad_v_next: Tensor, grad_C_next: Tensor, grad_F_next: Tensor):
        model, tape, statics = ctx.model, ctx.tape, ctx.statics
        state_curr, state_next = ctx.state_curr, ctx.state_next
        state_next.from_torch_grad(grad_x=grad_x_next, grad_v=grad_v_next, grad_C=grad_C_next, grad_F=grad_F_next)
        model.backward(statics, state_curr, state_next, tape)
        grad_x, grad_v, grad_C, grad_F, grad_stress = state_curr.to_torch_grad()
        for g in (grad_x, grad_v, grad_C, grad_F, grad_stress):
            if g is not None:
                torch.nan_to_num_(g, 0.0, 0.0, 0.0)          # interface.py:65-74
        return None, None, None, None, grad_x, grad_v, grad_C, grad_F, grad_stress


class MPMSim(nn.Module):
    """interface.py:79-93"""

    def __init__(self, model: MPMModel, reorder="auto") -> None:
        super().__init__()
        self.model = model
        # not in the reference: shuffled particle sets (what prepare_simulation_data produces) are run on an internally
        # Hilbert-sorted copy, see order.py; reorder=False switches it off
        self.order = ParticleOrder(int(model.constant.num_grids), reorder)

    def _apply(self, statics: MPMStatics, state_curr: MPMState, state_next: MPMState, x, v, C, F, stress):
        if self.order.active(x):
            pm, inv = self.order.perm, self.order.inv
            out = MPMSimFunction.apply(self.model, self.order.statics(statics), state_curr, state_next, x[pm], v[pm], C[pm], F[pm],
                                       stress[pm])
            return tuple(o[inv] for o in out)
        return MPMSimFunction.apply(self.model, statics, state_curr, state_next, x, v, C, F, stress)

    def state(self, x: Tensor, v: Tensor, C: Tensor, F: Tensor, stress: Tensor, state: Optional[MPMState] = None) -> MPMState:
        model = self.model
        shape = x.size(0)
        if state is None:
            state = model.state(shape)
        state.from_torch(x=x, v=v, C=C, F=F, stress=stress)
        return state


class MPMDiffSim(MPMSim):
    """interface.py:96-105"""

    def forward(self, statics: MPMStatics, x: Tensor, v: Tensor, C: Tensor, F: Tensor, stress: Tensor):
        shape = x.size(0)
        state_curr = self.model.state(shape)
        state_next = self.model.state(shape)
        return self._apply(statics, state_curr, state_next, x, v, C, F, stress)


class MPMCacheDiffSim(MPMSim):
    """interface.py:108-123"""

    def __init__(self, model: MPMModel, num_steps: int, reorder="auto") -> None:
        super().__init__(model, reorder)
        self.curr_states = [None for _ in range(num_steps)]
        self.next_states = [None for _ in range(num_steps)]

    def forward(self, statics: MPMStatics, step: int, x: Tensor, v: Tensor, C: Tensor, F: Tensor, stress: Tensor):
        shape = x.size(0)
        if self.curr_states[step] is None:
            self.curr_states[step] = self.model.state(shape)
        if self.next_states[step] is None:
            self.next_states[step] = self.model.state(shape)
        return self._apply(statics, self.curr_states[step], self.next_states[step], x, v, C, F, stress)


class MPMForwardSim(MPMSim):
    """interface.py:126-135 (in place: state is both current and next)"""

    def forward(self, statics: MPMStatics, state: MPMState):
        x, v, C, F, stress = state.to_torch()
        if self.order.active(x):
            # shuffled particles: step an internally sorted copy and write the result back into the caller's buffers
            pm, inv = self.order.perm, self.order.inv
            ps = getattr(self, "_sorted_state", None)
            if ps is None or ps.particle.x.shape[0] != x.shape[0]:
                ps = self._sorted_state = self.model.state(x.shape[0])
            for dst, src in zip(ps.to_torch(), (x, v, C, F, stress)):
                torch.index_select(src.detach(), 0, pm, out=dst)
            self.model.forward(self.order.statics(statics), ps, ps, None)
            for dst, src in zip((x, v, C, F), ps.to_torch()[:4]):
                torch.index_select(src, 0, inv, out=dst.detach())
        else:
            self.model.forward(statics, state, state, None)
        x_next, v_next, C_next, F_next, _ = state.to_torch()
        return x_next, v_next, C_next, F_next


class MPMExtraSim(MPMSim):
    """interface.py:138-147"""

    def forward(self, statics: MPMStatics, state: MPMState, statics_extra: MPMStatics, state_extra: MPMState) -> Tensor:
        self.model.forward_extra(statics, state, statics_extra, state_extra)
        x_extra, _, _, _, _ = state_extra.to_torch()
        return x_extra
