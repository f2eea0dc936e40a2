"""LoRA for the constitutive nets' dense layers.

Same parameter names / state-dict keys / merge semantics as the Microsoft loralib copy the reference ships
(/root/reference/modules/nclaw/material/loralib.py: LinearLoRA 162-224, replace_with_linear_lora 52-59,
init_linear_lora 62-81, mark_only_lora_as_trainable 13-30, lora_state_dict 33-49), written against the
fused HIP MLP: a layer never runs by itself, it only hands its *effective* weight
    W_eff = W + (lora_B @ lora_A) * (lora_alpha / r)          (loralib.py:209-213 == 216-224 algebraically)
to the kernel; torch autograd carries dL/dW_eff back to lora_A / lora_B.
"""
import math
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F


class _LoraMerge(torch.autograd.Function):
    """W + scaling * B @ A and its adjoint through nm_lora_merge / nm_lora_merge_bwd (GPU tensors only)."""

    @staticmethod
    def forward(ctx, W, B, A, scaling):
        from .. import _lib as L
        Wc, Bc, Ac = W.detach().contiguous().float(), B.detach().contiguous().float(), A.detach().contiguous().float()
        out = torch.empty_like(Wc)
        L.check(L.lib().nm_lora_merge(Wc.shape[0], Wc.shape[1], Bc.shape[1], float(scaling), L.ptr(Wc), L.ptr(Bc), L.ptr(Ac),
                                      L.ptr(out), L.stream_ptr(Wc.device)), "nm_lora_merge")
        ctx.save_for_backward(Bc, Ac)
        ctx.scaling = float(scaling)
        return out

    @staticmethod
    def backward(ctx, g):
        from .. import _lib as L
        Bc, Ac = ctx.saved_tensors
        g = g.contiguous().float()
        gB, gA = torch.empty_like(Bc), torch.empty_like(Ac)
        L.check(L.lib().nm_lora_merge_bwd(g.shape[0], g.shape[1], Bc.shape[1], ctx.scaling, L.ptr(g), L.ptr(Bc), L.ptr(Ac),
                                          L.ptr(gB), L.ptr(gA), L.stream_ptr(g.device)), "nm_lora_merge_bwd")
        return (g if ctx.needs_input_grad[0] else None), gB, gA, None


class _LoraMergeLayers(torch.autograd.Function):
    """The effective weights of several LinearLoRA layers (one constitutive net) with ONE launch in each direction
    (nm_lora_merge_layers / nm_lora_merge_layers_bwd).  Inputs: scalings, then W, B, A per layer."""

    @staticmethod
    def forward(ctx, scalings, *wba):
        from .. import _lib as L
        n = len(scalings)
        prep = [t.detach().contiguous().float() for t in wba]
        outs = [torch.empty_like(prep[3 * i]) for i in range(n)]
        jobs = (L.nm_lora_layer * n)()
        for i in range(n):
            W, B, A = prep[3 * i:3 * i + 3]
            jobs[i] = L.nm_lora_layer(W.shape[0], W.shape[1], B.shape[1], float(scalings[i]), L.ptr(W), L.ptr(B), L.ptr(A),
                                      L.ptr(outs[i]), None)
        L.check(L.lib().nm_lora_merge_layers(n, jobs, L.stream_ptr(prep[0].device)), "nm_lora_merge_layers")
        ctx.save_for_backward(*[prep[3 * i + k] for i in range(n) for k in (1, 2)])
        ctx.scalings = [float(s) for s in scalings]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        from .. import _lib as L
        n = len(ctx.scalings)
        BA = ctx.saved_tensors
        gs = [(g if g is not None else None) for g in gs]
        jobs = (L.nm_lora_layer * n)()
        gB, gA, gW = [], [], []
        for i in range(n):
            B, A = BA[2 * i], BA[2 * i + 1]
            g = gs[i].contiguous().float() if gs[i] is not None else torch.zeros(B.shape[0], A.shape[1], device=B.device)
            gW.append(g)
            gB.append(torch.empty_like(B)); gA.append(torch.empty_like(A))
            jobs[i] = L.nm_lora_layer(g.shape[0], g.shape[1], B.shape[1], ctx.scalings[i], L.ptr(g), L.ptr(B), L.ptr(A),
                                      L.ptr(gB[i]), L.ptr(gA[i]))
        L.check(L.lib().nm_lora_merge_layers_bwd(n, jobs, L.stream_ptr(BA[0].device)), "nm_lora_merge_layers_bwd")
        out = [None]
        for i in range(n):
            out += [gW[i] if ctx.needs_input_grad[1 + 3 * i] else None, gB[i], gA[i]]
        return tuple(out)


def merged_weights(layers, owner) -> tuple:
    """Effective weights of `layers` (LinearLoRA modules, unmerged, on the GPU) through _LoraMergeLayers, cached on
    `owner` under the same rules as LinearLoRA.effective_weight: reused until a parameter changes (version counters),
    the grad mode changes, or a backward pass has consumed the node."""
    key = tuple(v for l in layers for v in (l.weight._version, l.lora_A._version, l.lora_B._version, l.lora_A.requires_grad,
                                            l.lora_B.requires_grad, l.weight.requires_grad)) + (torch.is_grad_enabled(),)
    cached = getattr(owner, "_eff_cache", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    flat = [t for l in layers for t in (l.weight, l.lora_B, l.lora_A)]
    ws = _LoraMergeLayers.apply([l.scaling for l in layers], *flat)
    owner._eff_cache = (key, ws)
    if ws[0].grad_fn is not None:
        ws[0].grad_fn.register_hook(lambda *_: setattr(owner, "_eff_cache", None))
    return ws


class LinearLoRA(nn.Linear):
    def __init__(self, in_features: int, out_features: int, r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.,
                 merge_weights: bool = True, **kwargs):
        nn.Linear.__init__(self, in_features, out_features, **kwargs)
        if lora_dropout and lora_dropout > 0.:
            raise NotImplementedError("lora_dropout > 0 is never used by NeuMA and not supported by the fused MLP")
        self.r = r
        self.lora_alpha = lora_alpha
        self.merged = False
        self.merge_weights = merge_weights
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r, in_features)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_features, r)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        self.reset_parameters()

    def reset_parameters(self):
        nn.Linear.reset_parameters(self)
        if hasattr(self, 'lora_A'):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))   # loralib.py:190-193
            nn.init.zeros_(self.lora_B)

    def train(self, mode: bool = True):
        nn.Linear.train(self, mode)
        if mode:
            if self.merge_weights and self.merged:
                if self.r > 0:
                    self.weight.data -= (self.lora_B @ self.lora_A) * self.scaling
                self.merged = False
        else:
            if self.merge_weights and not self.merged:
                if self.r > 0:
                    self.weight.data += (self.lora_B @ self.lora_A) * self.scaling
                self.merged = True
        return self

    def effective_weight(self) -> torch.Tensor:
        if self.r > 0 and not self.merged:
            if self.weight.is_cuda:
                # A roll-out asks for the merged weight once per substep and net while the parameters only change at the
                # optimiser step: the merged tensor (and its autograd node) is reused until a parameter is modified
                # (version counters), the grad mode changes, or a backward pass has consumed the node.
                key = (self.weight._version, self.lora_A._version, self.lora_B._version, torch.is_grad_enabled(),
                       self.lora_A.requires_grad, self.lora_B.requires_grad, self.weight.requires_grad)
                cached = getattr(self, "_eff_cache", None)
                if cached is not None and cached[0] == key:
                    return cached[1]
                w = _LoraMerge.apply(self.weight, self.lora_B, self.lora_A, self.scaling)
                self._eff_cache = (key, w)
                if w.grad_fn is not None:
                    w.grad_fn.register_hook(lambda *_: setattr(self, "_eff_cache", None))
                return w
            return self.weight + (self.lora_B @ self.lora_A) * self.scaling      # host tensors (CPU-side unit tests)
        return self.weight

    def forward(self, x: torch.Tensor):
        # stand-alone use (not on the fused path): loralib.py:216-224
        if self.r > 0 and not self.merged:
            result = F.linear(x, self.weight, bias=self.bias)
            result = result + (x @ self.lora_A.transpose(0, 1) @ self.lora_B.transpose(0, 1)) * self.scaling
            return result
        return F.linear(x, self.weight, bias=self.bias)


def init_linear_lora(linear: nn.Linear, r: int, lora_alpha: int) -> LinearLoRA:
    """loralib.py:62-81: new LinearLoRA carrying the old layer's weight."""
    new = LinearLoRA(linear.in_features, linear.out_features, r=r, lora_alpha=lora_alpha, bias=linear.bias is not None)
    new = new.to(device=linear.weight.device, dtype=linear.weight.dtype)
    new.weight.data = linear.weight.data.clone()
    if linear.bias is not None:
        new.bias.data = linear.bias.data.clone()
    return new


def replace_with_linear_lora(model, old, r, lora_alpha):
    """loralib.py:52-59"""
    for n, module in model.named_children():
        if len(list(module.children())) > 0:
            replace_with_linear_lora(module, old, r=r, lora_alpha=lora_alpha)
        if isinstance(module, old) and not isinstance(module, LinearLoRA):
            setattr(model, n, init_linear_lora(module, r=r, lora_alpha=lora_alpha))


def mark_only_lora_as_trainable(model: nn.Module, bias: str = 'none') -> None:
    """loralib.py:13-30"""
    for n, p in model.named_parameters():
        if 'lora_' not in n:
            p.requires_grad = False
    if bias == 'none':
        return
    if bias == 'all':
        for n, p in model.named_parameters():
            if 'bias' in n:
                p.requires_grad = True
        return
    raise NotImplementedError


def lora_state_dict(model: nn.Module, bias: str = 'none') -> Dict[str, torch.Tensor]:
    """loralib.py:33-49"""
    sd = model.state_dict()
    if bias == 'none':
        return {k: sd[k] for k in sd if 'lora_' in k}
    if bias == 'all':
        return {k: sd[k] for k in sd if 'lora_' in k or 'bias' in k}
    raise NotImplementedError
