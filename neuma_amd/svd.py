"""Batched 3x3 SVD operator.  Mirrors /root/reference/modules/nclaw/warp/svd.py:9-101
(SVDFunction / SVD): F (N,3,3) -> U, sigma, Vh with U,V in SO(3), sign carried by sigma[:,2]."""
import torch
import torch.autograd as autograd
import torch.nn as nn
from torch import Tensor

from . import _lib as L


class SVDFunction(autograd.Function):

    @staticmethod
    def forward(ctx, F: Tensor):
        Fc = F.detach().float().contiguous()
        n = Fc.size(0)
        U = torch.empty_like(Fc)
        sigma = torch.empty(n, 3, dtype=torch.float32, device=Fc.device)
        Vh = torch.empty_like(Fc)
        L.check(L.lib().nm_svd3_fwd(n, L.ptr(Fc), L.ptr(U), L.ptr(sigma), L.ptr(Vh), L.stream_ptr(Fc.device)), "nm_svd3_fwd")
        ctx.save_for_backward(U, sigma, Vh)
        return U, sigma, Vh

    @staticmethod
    def backward(ctx, grad_U: Tensor, grad_sigma: Tensor, grad_Vh: Tensor):
        U, sigma, Vh = ctx.saved_tensors
        n = U.size(0)
        gU = None if grad_U is None else grad_U.float().contiguous()
        gs = None if grad_sigma is None else grad_sigma.float().contiguous()
        gV = None if grad_Vh is None else grad_Vh.float().contiguous()
        gF = torch.empty_like(U)
        L.check(L.lib().nm_svd3_bwd(n, L.ptr(U), L.ptr(sigma), L.ptr(Vh), L.ptr(gU), L.ptr(gs), L.ptr(gV), L.ptr(gF),
                                    L.stream_ptr(U.device)), "nm_svd3_bwd")
        return gF


class SVD(nn.Module):
    def forward(self, F: Tensor):
        return SVDFunction.apply(F)
