"""Fused roll-out fast path: S substeps of  stress=E(F); x,v,C,F=sim(...); F=P(F)  (finetune.py:360-364) as ONE
autograd node backed by nm_rollout_forward / nm_rollout_backward.  Keeps 96 B/particle/substep of checkpoints
and recomputes everything else in the reverse sweep (the substep's stress is checkpointed too: 132 B/particle).  Results equal the per-operator path
(MPMCacheDiffSim + material modules) up to fp32 summation order; tests/test_gpu_rollout.py checks that."""
import ctypes as C

import torch
import torch.autograd as autograd
import torch.nn as nn
from torch import Tensor

from . import _lib as L
from .sim.mpm import MPMModel, MPMStatics

_WSZ = (64 * 13, 64 * 64, 9 * 64)


class _Rollout(autograd.Function):

    @staticmethod
    def forward(ctx, model: MPMModel, statics: MPMStatics, substeps: int, alpha: float, x, v, C_, F, e0, e1, e2, p0, p1, p2):
        lib = L.lib()
        dev = x.device
        n = x.size(0)
        S = int(substeps)
        states = torch.empty(S + 1, 33 * n, dtype=torch.float32, device=dev)   # x|v|C|F|stress per record
        rec0 = states[0]
        rec0[:3 * n].copy_(x.detach().float().reshape(-1))
        rec0[3 * n:6 * n].copy_(v.detach().float().reshape(-1))
        rec0[6 * n:15 * n].copy_(C_.detach().float().reshape(-1))
        rec0[15 * n:24 * n].copy_(F.detach().float().reshape(-1))
        we = [t.detach().float().contiguous() for t in (e0, e1, e2)]
        wp = [t.detach().float().contiguous() for t in (p0, p1, p2)]
        ws_bytes = int(lib.nm_rollout_workspace(n, S))
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        cfg = L.nm_rollout_cfg(S, float(alpha))
        st = statics.c_struct()
        mle = L.nm_mlp(*[L.ptr(t) for t in we])
        mlp = L.nm_mlp(*[L.ptr(t) for t in wp])
        L.check(lib.nm_rollout_forward(model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp), L.ptr(states),
                                       L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "nm_rollout_forward")
        ctx.model, ctx.statics, ctx.S, ctx.alpha, ctx.n = model, statics, S, float(alpha), n
        ctx.save_for_backward(states, *we, *wp)
        last = states[S]
        return (last[:3 * n].view(n, 3), last[3 * n:6 * n].view(n, 3), last[6 * n:15 * n].view(n, 3, 3),
                last[15 * n:24 * n].view(n, 3, 3))

    @staticmethod
    def backward(ctx, gx, gv, gC, gF):
        lib = L.lib()
        states, e0, e1, e2, p0, p1, p2 = ctx.saved_tensors
        dev = states.device
        n, S = ctx.n, ctx.S
        glast = torch.empty(24 * n, dtype=torch.float32, device=dev)
        for sl, g, cnt in ((slice(0, 3 * n), gx, 3 * n), (slice(3 * n, 6 * n), gv, 3 * n), (slice(6 * n, 15 * n), gC, 9 * n),
                           (slice(15 * n, 24 * n), gF, 9 * n)):
            if g is None:
                glast[sl].zero_()
            else:
                glast[sl].copy_(g.float().reshape(-1))
        gfirst = torch.empty(24 * n, dtype=torch.float32, device=dev)
        gwe = torch.empty(sum(_WSZ), dtype=torch.float32, device=dev)
        gwp = torch.empty(sum(_WSZ), dtype=torch.float32, device=dev)
        ws_bytes = int(lib.nm_rollout_workspace(n, S))
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        cfg = L.nm_rollout_cfg(S, ctx.alpha)
        st = ctx.statics.c_struct()
        mle = L.nm_mlp(L.ptr(e0), L.ptr(e1), L.ptr(e2))
        mlp = L.nm_mlp(L.ptr(p0), L.ptr(p1), L.ptr(p2))
        L.check(lib.nm_rollout_backward(ctx.model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp),
                                        L.ptr(states), L.ptr(glast), L.ptr(gfirst), L.ptr(gwe), L.ptr(gwp), L.ptr(ws), ws_bytes,
                                        L.stream_ptr(dev)), "nm_rollout_backward")
        torch.nan_to_num_(gfirst, 0.0, 0.0, 0.0)   # interface.py:65-74 at the boundary of the fused node
        a, b = _WSZ[0], _WSZ[0] + _WSZ[1]
        return (None, None, None, None,
                gfirst[:3 * n].view(n, 3), gfirst[3 * n:6 * n].view(n, 3), gfirst[6 * n:15 * n].view(n, 3, 3),
                gfirst[15 * n:].view(n, 3, 3),
                gwe[:a].view(64, 13), gwe[a:b].view(64, 64), gwe[b:].view(9, 64),
                gwp[:a].view(64, 13), gwp[a:b].view(64, 64), gwp[b:].view(9, 64))


class MPMFusedDiffSim(nn.Module):
    """sim(statics, x, v, C, F) -> (x, v, C, F) after `substeps` substeps, constitutive nets included."""

    def __init__(self, model: MPMModel, elasticity: nn.Module, plasticity: nn.Module, substeps: int) -> None:
        super().__init__()
        self.model, self.elasticity, self.plasticity, self.substeps = model, elasticity, plasticity, int(substeps)

    def forward(self, statics: MPMStatics, x: Tensor, v: Tensor, C_: Tensor, F: Tensor):
        e = self.elasticity.effective_weights()
        p = self.plasticity.effective_weights()
        return _Rollout.apply(self.model, statics, self.substeps, self.plasticity.alpha, x, v, C_, F, *e, *p)
