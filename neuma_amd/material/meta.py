"""Neural constitutive models used by every NeuMA driver.

Mirrors /root/reference/modules/nclaw/material/meta.py: MLPBlock 20-42, InvariantFullMetaElasticity 170-221,
InvariantFullMetaPlasticity 442-489 — same module tree, so the shipped checkpoints
(`layers.{0,1}.fc.weight`, `final_layer.fc.weight`) and `*_lora.pt` files load unchanged.  forward() hands the
LoRA-merged weights to one fused HIP kernel (SVD -> 13 invariants -> 64 -> 64 -> 9 MLP on MFMA -> R X F^T or
F + alpha R X); there is no per-layer torch path.
"""
import ctypes as C
from typing import Optional

import torch
import torch.autograd as autograd
import torch.nn as nn
from torch import Tensor

from .. import _lib as L
from .loralib import LinearLoRA, mark_only_lora_as_trainable, lora_state_dict, merged_weights, replace_with_linear_lora

NM_ELASTICITY, NM_PLASTICITY = 0, 1


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class MaterialFunction(autograd.Function):
    """(F, W0, W1, W2) -> out, through nm_material_fwd / nm_material_bwd."""

    @staticmethod
    def forward(ctx, F: Tensor, w0: Tensor, w1: Tensor, w2: Tensor, kind: int, alpha: float, svd_adjoint: int = 0):
        Fc = F.detach().float().contiguous()
        w = [t.detach().float().contiguous() for t in (w0, w1, w2)]
        n = Fc.size(0)
        out = torch.empty_like(Fc)
        mlp = L.nm_mlp(L.ptr(w[0]), L.ptr(w[1]), L.ptr(w[2]))
        L.check(L.lib().nm_material_fwd(n, kind, float(alpha), L.ptr(Fc), C.byref(mlp), L.ptr(out), L.stream_ptr(Fc.device)),
                "nm_material_fwd")
        ctx.save_for_backward(Fc, *w)
        ctx.kind, ctx.alpha = kind, float(alpha)
        ctx.flags = 2 if svd_adjoint else 0          # NM_BWD_POLAR_ADJOINT
        ctx.need_w = any(ctx.needs_input_grad[1:4])
        return out

    @staticmethod
    def backward(ctx, gout: Tensor):
        Fc, w0, w1, w2 = ctx.saved_tensors
        n = Fc.size(0)
        g = gout.float().contiguous()
        gF = torch.empty_like(Fc)
        lib = L.lib()
        mlp = L.nm_mlp(L.ptr(w0), L.ptr(w1), L.ptr(w2))
        if ctx.need_w:
            gw0, gw1, gw2 = torch.empty_like(w0), torch.empty_like(w1), torch.empty_like(w2)
            nbytes = int(lib.nm_material_bwd_workspace(n))
            ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=Fc.device)
            L.check(lib.nm_material_bwd_ex(n, ctx.kind, ctx.alpha, L.ptr(Fc), C.byref(mlp), L.ptr(g), L.ptr(gF), L.ptr(gw0),
                                           L.ptr(gw1), L.ptr(gw2), ctx.flags, L.ptr(ws), nbytes, L.stream_ptr(Fc.device)), "nm_material_bwd_ex")
        else:
            gw0 = gw1 = gw2 = None
            L.check(lib.nm_material_bwd_ex(n, ctx.kind, ctx.alpha, L.ptr(Fc), C.byref(mlp), L.ptr(g), L.ptr(gF), None, None, None,
                                           ctx.flags, None, 0, L.stream_ptr(Fc.device)), "nm_material_bwd_ex")
        return gF, gw0, gw1, gw2, None, None, None


class MLPBlock(nn.Module):
    """meta.py:20-42 restricted to what the drivers instantiate: Linear without bias (+ GELU, applied in-kernel)."""

    def __init__(self, in_planes: int, out_planes: int, no_bias: bool, norm: Optional[str], nonlinearity: Optional[str]):
        super().__init__()
        if norm is not None or not no_bias:
            raise NotImplementedError("the fused MLP implements norm=None, no_bias=True (all shipped NeuMA configs)")
        if nonlinearity not in (None, 'gelu', 'GELU'):
            raise NotImplementedError("the fused MLP implements nonlinearity='gelu' (all shipped NeuMA configs)")
        self.fc = nn.Linear(in_planes, out_planes, bias=False)
        self.norm = nn.Identity()
        self.nonlinearity = nn.GELU() if nonlinearity else nn.Identity()

    def effective_weight(self) -> Tensor:
        fc = self.fc
        return fc.effective_weight() if isinstance(fc, LinearLoRA) else fc.weight


class _InvariantFullMeta(nn.Module):
    KIND = None

    def __init__(self, cfg) -> None:
        super().__init__()
        self.dim = 3
        widths = list(_get(cfg, "layer_widths"))
        if widths != [64, 64]:
            raise NotImplementedError("fused constitutive MLP is specialised to layer_widths [64, 64]")
        if not _get(cfg, "normalize_input", True):
            raise NotImplementedError("normalize_input=False is never used by NeuMA and not implemented")
        self.normalize_input = True
        self.layers = nn.ModuleList()
        width = self.dim + self.dim * self.dim + 1
        for next_width in widths:
            self.layers.append(MLPBlock(width, next_width, _get(cfg, "no_bias"), _get(cfg, "norm"), _get(cfg, "nonlinearity")))
            width = next_width
        self.final_layer = MLPBlock(width, self.dim * self.dim, _get(cfg, "no_bias"), None, None)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)      # material/utils.py:47-54

    # LoRA API, meta.py:186-194 / 458-466
    def init_lora_layers(self, r: int, lora_alpha: int = 1):
        replace_with_linear_lora(self, nn.Linear, r, lora_alpha)
        print(f"Initialized LoRA layers in {type(self).__name__} with r={r} and alpha={lora_alpha}.")

    def freeze_all_except_lora(self):
        mark_only_lora_as_trainable(self)

    def lora_state_dict(self, bias: str = 'none'):
        return lora_state_dict(self, bias)

    def effective_weights(self):
        fcs = (self.layers[0].fc, self.layers[1].fc, self.final_layer.fc)
        if all(isinstance(fc, LinearLoRA) and fc.r > 0 and not fc.merged and fc.weight.is_cuda for fc in fcs):
            return merged_weights(fcs, self)       # the three layers' merges (and their adjoints) in one launch each
        return (self.layers[0].effective_weight(), self.layers[1].effective_weight(), self.final_layer.effective_weight())

    def _alpha(self) -> float:
        return 0.0

    def forward(self, F: Tensor) -> Tensor:
        w0, w1, w2 = self.effective_weights()
        # svd_adjoint (attribute, "reference" by default): see rollout.MPMFusedDiffSim
        return MaterialFunction.apply(F, w0, w1, w2, self.KIND, self._alpha(), L.SVD_ADJOINT[getattr(self, "svd_adjoint", "reference")])


class InvariantFullMetaElasticity(_InvariantFullMeta):
    """meta.py:170-221: F -> Kirchhoff-type stress R sym(X) F^T as consumed by p2g."""
    KIND = NM_ELASTICITY


class InvariantFullMetaPlasticity(_InvariantFullMeta):
    """meta.py:442-489: F -> F + alpha R sym(X)."""
    KIND = NM_PLASTICITY

    def __init__(self, cfg) -> None:
        super().__init__(cfg)
        self.alpha = float(_get(cfg, "alpha"))

    def _alpha(self) -> float:
        return self.alpha
