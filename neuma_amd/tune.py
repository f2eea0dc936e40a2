"""Tuning glue between simulator and renderer.
Mirrors /root/reference/modules/tune/utils.py: diff_rasterization 323-421, compute_bindings_xyz 424-448,
compute_bindings_F 451-472, preprocess_for_rasterization 475-523; modules/nclaw/utils.py:110-118
(denormalize_points_helper_func); modules/d3gs/utils/loss_utils.py:17-24."""
import ctypes as C
from typing import List, Optional

import numpy as np
import torch
import torch.autograd as autograd
from torch import Tensor

from . import _lib as L
from .render import get_rasterizer, deform_cov_by_F


class Bindings(object):
    """CSR (forward) + transposed CSR (backward) copies of the sparse particle->Gaussian binding matrix
    (K x N, <= 10 nnz per row, tune/utils.py:297-317)."""

    def __init__(self, indices: Tensor, values: Tensor, size, device=None):
        device = device if device is not None else values.device
        K, N = int(size[0]), int(size[1])
        rows = indices[0].to("cpu", torch.int64)
        cols = indices[1].to("cpu", torch.int64)
        vals = values.to("cpu", torch.float32)
        self.K, self.N, self.nnz = K, N, int(vals.numel())
        self.rowptr, self.col, self.val = self._csr(rows, cols, vals, K, device)
        self.t_rowptr, self.t_col, self.t_val = self._csr(cols, rows, vals, N, device)

    @staticmethod
    def _csr(rows, cols, vals, nrows, device):
        order = torch.argsort(rows * (int(cols.max()) + 1 if cols.numel() else 1) + cols)
        r, c, v = rows[order], cols[order], vals[order]
        counts = torch.bincount(r, minlength=nrows)
        rowptr = torch.zeros(nrows + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(counts, 0)
        return rowptr.to(device, torch.int32), c.to(device, torch.int32), v.to(device, torch.float32)

    @classmethod
    def from_sparse(cls, B: Tensor, device=None) -> "Bindings":
        B = B.coalesce()
        return cls(B.indices(), B.values(), B.size(), device if device is not None else B.device)

    @classmethod
    def of(cls, B) -> "Bindings":
        if isinstance(B, Bindings):
            return B
        cached = getattr(B, "_nm_bindings", None)
        if cached is None:
            cached = cls.from_sparse(B)
            try:
                B._nm_bindings = cached
            except Exception:
                pass
        return cached


class _SpMM(autograd.Function):
    @staticmethod
    def forward(ctx, X: Tensor, b: Bindings):
        Xc = X.detach().float().contiguous()
        D = Xc.size(1)
        out = torch.empty(b.K, D, dtype=torch.float32, device=Xc.device)
        L.check(L.lib().nm_spmm_csr(b.K, D, L.ptr(b.rowptr), L.ptr(b.col), L.ptr(b.val), L.ptr(Xc), L.ptr(out),
                                    L.stream_ptr(Xc.device)), "nm_spmm_csr")
        ctx.b = b
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        b = ctx.b
        gc = g.float().contiguous()
        D = gc.size(1)
        out = torch.empty(b.N, D, dtype=torch.float32, device=gc.device)
        L.check(L.lib().nm_spmm_csr(b.N, D, L.ptr(b.t_rowptr), L.ptr(b.t_col), L.ptr(b.t_val), L.ptr(gc), L.ptr(out),
                                    L.stream_ptr(gc.device)), "nm_spmm_csr")
        return out, None


def spmm(bindings, X: Tensor) -> Tensor:
    return _SpMM.apply(X, Bindings.of(bindings))


def compute_bindings_xyz(p_curr: Tensor, p_prev: Tensor, k_prev: Tensor, bindings) -> Tensor:
    """tune/utils.py:424-448"""
    delta_x = p_curr - p_prev.detach()
    return k_prev.detach() + spmm(bindings, delta_x)


def compute_bindings_F(deform_grad: Tensor, bindings) -> Tensor:
    """tune/utils.py:451-472"""
    return spmm(bindings, torch.reshape(deform_grad, (-1, 9))).reshape(-1, 3, 3)


def denormalize_points_helper_func(points: Tensor, size, center) -> Tensor:
    """modules/nclaw/utils.py:110-118"""
    if isinstance(size, np.ndarray):
        size = torch.from_numpy(size).to(points)
    if isinstance(center, np.ndarray):
        center = torch.from_numpy(center).to(points)
    return (points.clone() - center) / size


class _PixelLoss(torch.autograd.Function):
    """mean |a-b| (kind 0) or mean (a-b)^2 (kind 1) over a (3,H,W) image, restricted to pixel rows [row0,row1) (the mean
    is still over the full image, so stripes add up): one fused kernel (nm_pixel_loss) produces the value and dL/dimage."""

    @staticmethod
    def forward(ctx, img, gt, kind, row0, row1):
        lib = L.lib()
        a = img.detach().contiguous().float()
        b = gt.detach().contiguous().float()
        h, w = int(a.shape[-2]), int(a.shape[-1])
        loss = torch.zeros((), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        L.check(lib.nm_pixel_loss(int(kind), 1.0, h, w, int(row0), int(row1), L.ptr(a), L.ptr(b), L.ptr(loss),
                                  L.ptr(grad) if grad is not None else None, L.stream_ptr(a.device)), "nm_pixel_loss")
        ctx.grad = grad
        return loss

    @staticmethod
    def backward(ctx, g):
        grad, ctx.grad = ctx.grad, None
        return (grad * g if grad is not None else None), None, None, None, None


def _fused_loss_ok(a, b):
    return a.is_cuda and b.is_cuda and a.dim() == 3 and a.shape == b.shape and a.shape[0] == 3 and not b.requires_grad


def pixel_loss_rows(network_output, gt, kind: int, row0: int = 0, row1: int = 0):
    """Loss of pixel rows [row0,row1) normalised by the FULL image size (multi-GPU stripes sum to the 1-GPU loss)."""
    return _PixelLoss.apply(network_output, gt, kind, row0, row1)


def l1_loss(network_output, gt):
    """loss_utils.py:17-18"""
    if _fused_loss_ok(network_output, gt):
        return _PixelLoss.apply(network_output, gt, 0, 0, 0)
    return torch.abs((network_output - gt)).mean()


def l2_loss(network_output, gt):
    """loss_utils.py:23-24"""
    if _fused_loss_ok(network_output, gt):
        return _PixelLoss.apply(network_output, gt, 1, 0, 0)
    return ((network_output - gt) ** 2).mean()


def diff_rasterization(x: Tensor, deform_grad: Optional[Tensor], gaussians, view_cam, background_color: Tensor,
                       gaussians_active_sh: Optional[int] = None, guassians_cov: Optional[Tensor] = None,
                       gaussians_opa: Optional[Tensor] = None, gaussians_shs: Optional[Tensor] = None,
                       scaling_modifier: Optional[float] = 1., force_mask_data: Optional[bool] = False,
                       tile_rows=None) -> Tensor:
    """tune/utils.py:323-421 (argument names kept, including the reference's `guassians_cov` spelling)."""
    means3D = x
    if gaussians is not None:
        cov3D_precomp = gaussians.get_covariance(scaling_modifier=scaling_modifier)
        opacity = gaussians.get_opacity
        shs = gaussians.get_features
        sh_degree = gaussians.active_sh_degree
    else:
        cov3D_precomp, opacity, shs, sh_degree = guassians_cov, gaussians_opa, gaussians_shs, gaussians_active_sh
    assert means3D.shape[0] == cov3D_precomp.shape[0], \
        f"Shape mismatch: means3D {means3D.shape[0]} cov3D {cov3D_precomp.shape[0]}"
    if deform_grad is not None:
        tensor_F = torch.reshape(deform_grad, (-1, 3, 3))
        assert cov3D_precomp.shape[0] == tensor_F.shape[0], \
            f"Shape mismatch: cov3D {cov3D_precomp.shape[0]} F {tensor_F.shape[0]}"
        cov3D_deformed = deform_cov_by_F(cov3D_precomp.reshape(-1, 6), tensor_F)
    else:
        cov3D_deformed = cov3D_precomp
    means2D = torch.zeros_like(means3D, requires_grad=True)     # (the reference adds 0 to make it a non-leaf it can retain_grad on)
    rasterizer = get_rasterizer(view_cam, sh_degree, debug=False, bg_color=background_color, tile_rows=tile_rows)
    if force_mask_data:
        rendered_image, _ = rasterizer(means3D=means3D, means2D=means2D, shs=None,
                                       colors_precomp=torch.ones(means3D.shape[0], 3, device=means3D.device),
                                       opacities=opacity, scales=None, rotations=None, cov3D_precomp=cov3D_deformed)
    else:
        rendered_image, _ = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=None,
                                       opacities=opacity, scales=None, rotations=None, cov3D_precomp=cov3D_deformed)
    return rendered_image


def preprocess_for_rasterization(obj_gaussians: List, obj_deform_grad: List[Tensor], obj_kernels_prev: List[Tensor],
                                 obj_particles_curr: List[Tensor], obj_particles_prev: List[Tensor],
                                 obj_bindings: List, obj_scalings: List[float]):
    """tune/utils.py:475-523 (multi-object concat)."""
    obj_x = [compute_bindings_xyz(pc, pp, kp, b) for pc, pp, kp, b in
             zip(obj_particles_curr, obj_particles_prev, obj_kernels_prev, obj_bindings)]
    obj_F = [compute_bindings_F(F, b) for F, b in zip(obj_deform_grad, obj_bindings)]
    obj_cov = [g.get_covariance(scaling_modifier=s) for g, s in zip(obj_gaussians, obj_scalings)]
    obj_opa = [g.get_opacity for g in obj_gaussians]
    obj_shs = [g.get_features for g in obj_gaussians]
    return {"means3D": torch.cat(obj_x, 0), "deform_grad": torch.cat(obj_F, 0), "cov3D": torch.cat(obj_cov, 0),
            "opacity": torch.cat(obj_opa, 0), "shs": torch.cat(obj_shs, 0),
            "active_sh_degree": obj_gaussians[0].active_sh_degree}
