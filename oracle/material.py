"""Oracle (TEST INFRASTRUCTURE ONLY) — batched 3x3 SVD convention and the neural constitutive nets.

Follows
  /root/reference/modules/nclaw/warp/svd.py:61-96      batch_svd det/sign rule
  /root/reference/modules/nclaw/material/meta.py:20-42   MLPBlock (Linear no-bias -> GELU exact)
  /root/reference/modules/nclaw/material/meta.py:196-221 InvariantFullMetaElasticity.forward
  /root/reference/modules/nclaw/material/meta.py:468-489 InvariantFullMetaPlasticity.forward
  /root/reference/modules/nclaw/material/loralib.py:216-224 LinearLoRA.forward (un-merged)
  /root/reference/modules/nclaw/material/loralib.py:199-214 merge on eval
  /root/reference/modules/nclaw/material/preset.py:21-27  ComposeMaterial.forward

Pinned: tests/golden/material_*.npz were produced by the reference's own classes (imported with a
stub `warp`) on the three shipped checkpoints; tests/test_oracle_material.py checks this file
against them.  wp.svd3 itself lives in warp-lang (absent): its internal sigma ordering is
"parity unpinned"; the convention adopted is sigma0 >= sigma1 >= |sigma2|, U,V in SO(3), sign on
sigma2, which is what torch.linalg.svd + the svd.py:76-92 rule yields.
"""
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as Fnn
from torch import Tensor


def svd3(F: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """svd.py:61-96: U, sigma, Vh with det(U)=det(V)=+1, sign carried by sigma[2]."""
    U, s, Vh = torch.linalg.svd(F)
    detU = torch.linalg.det(U)
    detV = torch.linalg.det(Vh)
    fu = torch.where(detU < 0, -torch.ones_like(detU), torch.ones_like(detU))
    fv = torch.where(detV < 0, -torch.ones_like(detV), torch.ones_like(detV))
    one = torch.ones_like(fu)
    U = U * torch.stack([one, one, fu], dim=-1)[:, None, :]        # flip column 2 of U   :76-82
    Vh = Vh * torch.stack([one, one, fv], dim=-1)[:, :, None]      # flip row 2 of Vh (= column 2 of V) :83-89
    s = s * torch.stack([one, one, fu * fv], dim=-1)
    return U, s, Vh


def svd3_adjoint(U: Tensor, s: Tensor, Vh: Tensor, gU: Tensor, gs: Tensor, gVh: Tensor,
                 clamp: float = 1e-6) -> Tensor:
    """Closed-form adjoint of svd3 with the denominator clamp of warp's adj_svd3.
    E_ij = 1/min(s_j^2 - s_i^2, -clamp) for i<j, antisymmetric."""
    V = Vh.transpose(-1, -2)
    gV = gVh.transpose(-1, -2)
    s2 = s * s
    diff = s2[:, None, :] - s2[:, :, None]                    # [i,j] = s_j^2 - s_i^2
    E = torch.zeros_like(diff)
    for i in range(3):
        for j in range(i + 1, 3):
            e = 1.0 / torch.minimum(diff[:, i, j], torch.full_like(diff[:, i, j], -clamp))
            E[:, i, j] = e
            E[:, j, i] = -e
    UtgU = U.transpose(-1, -2) @ gU
    VtgV = V.transpose(-1, -2) @ gV
    S = torch.diag_embed(s)
    inner = (E * (UtgU - UtgU.transpose(-1, -2))) @ S + S @ (E * (VtgV - VtgV.transpose(-1, -2))) \
        + torch.diag_embed(gs)
    return U @ inner @ Vh


def gelu(x: Tensor) -> Tensor:
    """material/utils.py:16-17 -> nn.GELU() (exact erf form)."""
    return Fnn.gelu(x)


def lora_effective_weight(W: Tensor, A: Optional[Tensor], B: Optional[Tensor], scaling: float) -> Tensor:
    """loralib.py:209-213: W + (B @ A) * scaling (merged form; mathematically equal to :216-224)."""
    if A is None or B is None:
        return W
    return W + (B @ A) * scaling


def linear_lora(x: Tensor, W: Tensor, A: Optional[Tensor], B: Optional[Tensor], scaling: float) -> Tensor:
    """loralib.py:216-224 un-merged forward."""
    y = Fnn.linear(x, W)
    if A is not None and B is not None:
        y = y + (x @ A.transpose(0, 1) @ B.transpose(0, 1)) * scaling
    return y


def invariants(F: Tensor) -> Tuple[Tensor, Tensor]:
    """meta.py:197-213 — returns (features (N,13), R (N,3,3)); normalize_input=True."""
    I = torch.eye(3, dtype=F.dtype)
    U, sigma, Vh = svd3(F)
    R = U @ Vh
    FtF = F.transpose(-1, -2) @ F
    I1 = sigma - 1.0
    I2 = (FtF - I).reshape(-1, 9)
    I3 = torch.linalg.det(F).unsqueeze(1) - 1.0
    return torch.cat([I1, I2, I3], dim=1), R


def mlp(z: Tensor, weights: Sequence[Tensor], lora: Optional[Sequence[Tuple[Tensor, Tensor]]] = None,
        scaling: float = 1.0) -> Tensor:
    """meta.py:215-217: hidden MLPBlocks (Linear no-bias + GELU), final Linear no-bias."""
    x = z
    n = len(weights)
    for i, W in enumerate(weights):
        A, B = (lora[i] if lora is not None else (None, None))
        x = linear_lora(x, W, A, B, scaling)
        if i < n - 1:
            x = gelu(x)
    return x


def elasticity(F: Tensor, weights, lora=None, scaling: float = 1.0) -> Tensor:
    """meta.py:196-221: stress ('cauchy' as consumed by p2g) = R sym(X) F^T."""
    z, R = invariants(F)
    X = mlp(z, weights, lora, scaling).reshape(-1, 3, 3)
    X = 0.5 * (X.transpose(-1, -2) + X)
    return R @ X @ F.transpose(-1, -2)


def plasticity(F: Tensor, weights, alpha: float, lora=None, scaling: float = 1.0) -> Tensor:
    """meta.py:468-489: F + alpha * R sym(X)."""
    z, R = invariants(F)
    X = mlp(z, weights, lora, scaling).reshape(-1, 3, 3)
    X = 0.5 * (X.transpose(-1, -2) + X)
    return alpha * (R @ X) + F


def compose(fns, sections, F: Tensor) -> Tensor:
    """preset.py:21-27: split by sections, apply, concatenate (empty sections skipped)."""
    outs = []
    for fn, f in zip(fns, torch.split(F, list(sections), dim=0)):
        if f.numel() == 0:
            continue
        outs.append(fn(f))
    return torch.cat(outs, dim=0)
