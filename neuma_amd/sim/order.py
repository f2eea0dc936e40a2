"""Transparent particle re-ordering for the scatter kernels.

The scatter kernels (p2g, g2p adjoint) are ~20x faster when consecutive particles are spatial neighbours, and NeuMA's
data preparation shuffles the particles (/root/reference/modules/tune/utils.py:270-272).  ParticleOrder measures the
order once (fraction of consecutive particles whose stencil origins int(x*G - 0.5) are adjacent); below 80 % it builds
a device-side Hilbert-curve permutation of the stencil origins.  Callers gather their inputs with `perm`, run the engine,
and scatter results back with `inv` - plain differentiable torch indexing, so users never see the internal order.
The permutation is fixed at the first call (particles move a fraction of a cell per substep; the kernels re-sort inside
LDS every launch anyway).
"""
from typing import Optional

import torch
from torch import Tensor


def hilbert_index_torch(cells: Tensor, bits: int) -> Tensor:
    """Device version of synth.hilbert_index (Skilling's transpose algorithm) on an (N,3) int64 tensor."""
    X = [cells[:, 0].clone(), cells[:, 1].clone(), cells[:, 2].clone()]
    Q = 1 << (bits - 1)
    while Q > 1:
        P = Q - 1
        for i in range(3):
            hit = (X[i] & Q) != 0
            t = (X[0] ^ X[i]) & P
            x0_hit = X[0] ^ P
            x0_miss = X[0] ^ t
            xi_miss = X[i] ^ t
            if i == 0:
                X[0] = torch.where(hit, x0_hit, X[0])          # t == 0 for i == 0
            else:
                X[i] = torch.where(hit, X[i], xi_miss)
                X[0] = torch.where(hit, x0_hit, x0_miss)
        Q >>= 1
    X[1] = X[1] ^ X[0]
    X[2] = X[2] ^ X[1]
    t = torch.zeros_like(X[0])
    Q = 1 << (bits - 1)
    while Q > 1:
        t = torch.where((X[2] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    X = [x ^ t for x in X]
    idx = torch.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            idx = (idx << 1) | ((X[i] >> b) & 1)
    return idx


class ParticleOrder(object):
    def __init__(self, num_grids: int, mode="auto", threshold: float = 0.8) -> None:
        self.G, self.mode, self.threshold = int(num_grids), mode, float(threshold)
        self.perm: Optional[Tensor] = None      # None: undecided; False: not needed; tensor: permutation
        self.inv: Optional[Tensor] = None
        self._statics_cache = None

    def quality(self, x: Tensor) -> float:
        base = torch.trunc(x.detach() * self.G - 0.5).clamp_(min=0).long()
        d = (base[1:] - base[:-1]).abs().max(dim=1).values
        return float((d <= 1).float().mean()) if d.numel() else 1.0

    def active(self, x: Tensor) -> bool:
        """Decide (first call, or when the particle count changed) and report whether a permutation is in use."""
        if self.perm is None or (self.perm is not False and self.perm.numel() != x.shape[0]):
            want = self.mode
            if want == "auto":
                want = x.shape[0] > 1 and self.quality(x) < self.threshold
            if not want:
                self.perm = False
            else:
                bits = max(1, (self.G + 1).bit_length())
                base = torch.trunc(x.detach() * self.G - 0.5).clamp_(min=0).long()
                self.perm = torch.argsort(hilbert_index_torch(base, bits), stable=True)
                self.inv = torch.empty_like(self.perm)
                self.inv[self.perm] = torch.arange(self.perm.numel(), device=self.perm.device)
                self._statics_cache = None
        return self.perm is not False

    def statics(self, statics):
        """Permuted copy of an MPMStatics, cached until one of its tensors is modified."""
        from .mpm import MPMStatics
        key = tuple((t.data_ptr(), t._version) for t in (statics.vol, statics.rho, statics.clip_bound, statics.enabled))
        if self._statics_cache is None or self._statics_cache[0] != key:
            st = MPMStatics()
            st.vol, st.rho = statics.vol[self.perm].contiguous(), statics.rho[self.perm].contiguous()
            st.clip_bound, st.enabled = statics.clip_bound[self.perm].contiguous(), statics.enabled[self.perm].contiguous()
            self._statics_cache = (key, st)
        return self._statics_cache[1]
