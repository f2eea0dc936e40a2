"""Oracle (TEST INFRASTRUCTURE ONLY) — Particle-GS rasterizer, dense per-pixel restatement in PyTorch.

The reference reaches its rasterizer through the un-vendored CUDA extension
`diff_gaussian_rasterization` (graphdeco-inria/gaussian-splatting @ b17ded92..., README.md:56-61;
call sites modules/d3gs/gaussian_renderer/__init__.py:92-119 and modules/tune/utils.py:385-419).
Its source is not under /root/reference and the reference holds no tests or vectors for it:
PARITY UNPINNED.  This file restates the published algorithm (3DGS, Kerbl et al. 2023, and the
constants listed in SURVEY.md App. D); camera / SH conventions are pinned against the reference's
own pure-python modules (tests/golden/camera_sh_golden.npz):
  modules/d3gs/scene/cameras.py:54-57       row-vector convention, matrices arrive transposed
  modules/d3gs/utils/graphics_utils.py:51-71 projection matrix
  modules/d3gs/utils/sh_utils.py:26-108      SH basis constants / evaluation order
  modules/d3gs/utils/general_utils.py:93-139 cov6 order xx,xy,xz,yy,yz,zz; quaternion (r,x,y,z) -> R

Gradients: torch autograd through this forward.  One deliberate deviation from plain autograd to
follow the upstream backward: alpha = min(0.99, o*G) propagates its gradient as if unclamped
(straight-through), as upstream's renderCUDA backward does.

Dense formulation: every pixel evaluates every Gaussian (sorted by (depth, index)), masked by the
Gaussian's 16x16-tile rectangle — O(pixels*K) memory, for test sizes only.
"""
import math
from typing import NamedTuple, Optional

import torch
from torch import Tensor

BLOCK = 16

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class Settings(NamedTuple):
    """Field order of diff_gaussian_rasterization.GaussianRasterizationSettings as filled by
    gaussian_renderer/__init__.py:103-116."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool = False
    debug: bool = False


def eval_sh_color(deg: int, shs: Tensor, means3D: Tensor, campos: Tensor):
    """shs (K, M, 3) with M >= (deg+1)^2; returns (rgb clamped at 0 (K,3), clamped mask)."""
    d = means3D - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6] + SH_C2[3] * xz * shs[:, 7]
                   + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * shs[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * shs[:, 13]
                       + SH_C3[5] * z * (xx - yy) * shs[:, 14] + SH_C3[6] * x * (xx - 3.0 * yy) * shs[:, 15])
    res = res + 0.5
    return torch.clamp_min(res, 0.0), res < 0


def cov6_to_mat(c: Tensor) -> Tensor:
    return torch.stack([torch.stack([c[:, 0], c[:, 1], c[:, 2]], -1),
                        torch.stack([c[:, 1], c[:, 3], c[:, 4]], -1),
                        torch.stack([c[:, 2], c[:, 4], c[:, 5]], -1)], -2)


def preprocess(s: Settings, means3D: Tensor, cov3D: Tensor, opacities: Tensor,
               shs: Optional[Tensor] = None, colors_precomp: Optional[Tensor] = None):
    """Per-Gaussian projection (upstream preprocessCUDA). Returns dict of per-Gaussian tensors."""
    K = means3D.shape[0]
    dt = means3D.dtype
    W, H = s.image_width, s.image_height
    ones = torch.ones(K, 1, dtype=dt)
    hom = torch.cat([means3D, ones], 1)
    p_hom = hom @ s.projmatrix.to(dt)                       # row-vector convention (cameras.py:54-56)
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * p_w[:, None]
    t = (hom @ s.viewmatrix.to(dt))[:, :3]
    depth = t[:, 2]
    visible = depth > 0.2                                    # near-plane cull

    fx = W / (2.0 * s.tanfovx)
    fy = H / (2.0 * s.tanfovy)
    limx, limy = 1.3 * s.tanfovx, 1.3 * s.tanfovy
    tz = torch.where(visible, t[:, 2], torch.ones_like(t[:, 2]))
    # upstream clamps t.xy/t.z to +-1.3 tanfov and, in its backward, treats the clamped value as a constant
    # (x_grad_mul = 0) — also w.r.t. t.z; reproduce that instead of autograd's d(clamp*tz)/dtz
    rx, ry = t[:, 0] / tz, t[:, 1] / tz
    in_x = (rx >= -limx) & (rx <= limx)
    in_y = (ry >= -limy) & (ry <= limy)
    tx = torch.where(in_x, t[:, 0], (torch.clamp(rx, -limx, limx) * tz).detach())
    ty = torch.where(in_y, t[:, 1], (torch.clamp(ry, -limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -(fy * ty) / (tz * tz)], -1)], -2)   # (K,2,3)
    Rv = s.viewmatrix.to(dt)[:3, :3].transpose(0, 1)         # column-vector view rotation
    Sig = cov6_to_mat(cov3D)
    M = J @ Rv[None]
    cov2 = M @ Sig @ M.transpose(-1, -2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    visible = visible & (det != 0)
    det_s = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach())).to(torch.long)
    xy = torch.stack([((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5, ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5], -1)
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK
    xyd = xy.detach()
    rf = radius.to(dt)

    def tr(v):  # C (int) cast
        return torch.trunc(v).to(torch.long)

    rmin_x = torch.clamp(tr((xyd[:, 0] - rf) / BLOCK), 0, gx)
    rmin_y = torch.clamp(tr((xyd[:, 1] - rf) / BLOCK), 0, gy)
    rmax_x = torch.clamp(tr((xyd[:, 0] + rf + BLOCK - 1) / BLOCK), 0, gx)
    rmax_y = torch.clamp(tr((xyd[:, 1] + rf + BLOCK - 1) / BLOCK), 0, gy)
    tiles = (rmax_x - rmin_x) * (rmax_y - rmin_y)
    visible = visible & (tiles > 0)
    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros_like(rgb, dtype=torch.bool)
    else:
        rgb, clamped = eval_sh_color(s.sh_degree, shs, means3D, s.campos.to(dt))
    radius = torch.where(visible, radius, torch.zeros_like(radius))
    tiles = torch.where(visible, tiles, torch.zeros_like(tiles))
    return dict(xy=xy, depth=depth, conic=conic, rgb=rgb, clamped=clamped, radius=radius, visible=visible,
                rect=(rmin_x, rmin_y, rmax_x, rmax_y), tiles=tiles, opacity=opacities.reshape(-1))


def render(s: Settings, means3D: Tensor, cov3D: Tensor, opacities: Tensor,
           shs: Optional[Tensor] = None, colors_precomp: Optional[Tensor] = None,
           row_chunk: int = 16, return_aux: bool = False):
    """Full forward: returns (image (3,H,W), radii (K,) int)."""
    pp = preprocess(s, means3D, cov3D, opacities, shs, colors_precomp)
    dt = means3D.dtype
    W, H = s.image_width, s.image_height
    vis = pp["visible"]
    idx = torch.nonzero(vis).reshape(-1)
    # sort by (depth, index): radix sort on (tile | depth bits) is stable w.r.t. emission order
    dkey = pp["depth"].detach()[idx]
    order = torch.argsort(dkey, stable=True)
    idx = idx[order]
    xy = pp["xy"][idx]
    con = pp["conic"][idx]
    op = pp["opacity"][idx]
    rgb = pp["rgb"][idx]
    rminx, rminy, rmaxx, rmaxy = [r[idx] for r in pp["rect"]]
    bg = s.bg.to(dt)
    rows_out = []
    n_contrib = []
    for y0 in range(0, H, row_chunk):
        y1 = min(H, y0 + row_chunk)
        py, px = torch.meshgrid(torch.arange(y0, y1), torch.arange(W), indexing="ij")
        px = px.reshape(-1)
        py = py.reshape(-1)
        tx_, ty_ = px // BLOCK, py // BLOCK
        in_rect = ((tx_[:, None] >= rminx[None]) & (tx_[:, None] < rmaxx[None]) &
                   (ty_[:, None] >= rminy[None]) & (ty_[:, None] < rmaxy[None]))          # (P,K)
        dx = xy[None, :, 0] - px[:, None].to(dt)
        dy = xy[None, :, 1] - py[:, None].to(dt)
        power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
        a_raw = op[None] * torch.exp(torch.clamp_max(power, 0.0))
        alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()   # straight-through clamp (upstream bwd)
        contrib = in_rect & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
        a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
        T_after = torch.cumprod(1.0 - a_eff, dim=1)
        stop = contrib & (T_after.detach() < 1e-4)
        keep = contrib & (torch.cumsum(stop.to(torch.int32), dim=1) == 0)
        T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], 1)
        wgt = torch.where(keep, alpha * T_before, torch.zeros_like(alpha))
        C = wgt @ rgb                                                                     # (P,3)
        T_fin = torch.prod(torch.where(keep, 1.0 - alpha, torch.ones_like(alpha)), dim=1)
        out = C + T_fin[:, None] * bg[None]
        rows_out.append(out.reshape(y1 - y0, W, 3))
        n_contrib.append(keep.sum(1).reshape(y1 - y0, W))
    img = torch.cat(rows_out, 0).permute(2, 0, 1)
    radii = pp["radius"].to(torch.int32)
    if return_aux:
        return img, radii, dict(pp=pp, n_contrib=torch.cat(n_contrib, 0), D=int(pp["tiles"].sum()))
    return img, radii


def build_cov3D(scales: Tensor, rotations: Tensor, scale_modifier: float = 1.0) -> Tensor:
    """general_utils.py:93-139 + gaussian_model.py:27-31: Sigma = (R S)(R S)^T packed xx,xy,xz,yy,yz,zz.
    rotations are quaternions (r,x,y,z), normalised here as build_rotation does."""
    q = rotations / rotations.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    L = R * (scale_modifier * scales)[:, None, :]
    S = L @ L.transpose(-1, -2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


def deform_cov_by_F(cov6: Tensor, F: Tensor) -> Tensor:
    """modules/d3gs/utils/simulation_utils.py:25-48: Sigma' = F Sigma F^T on packed 6 floats."""
    S = F @ cov6_to_mat(cov6) @ F.transpose(-1, -2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


def bindings_xyz(p_curr: Tensor, p_prev: Tensor, k_prev: Tensor, B: Tensor) -> Tensor:
    """modules/tune/utils.py:424-448 (B: dense or sparse (K,N))."""
    return k_prev.detach() + torch.sparse.mm(B, p_curr - p_prev.detach()) if B.is_sparse \
        else k_prev.detach() + B @ (p_curr - p_prev.detach())


def bindings_F(F: Tensor, B: Tensor) -> Tensor:
    """modules/tune/utils.py:451-472."""
    Ff = F.reshape(-1, 9)
    out = torch.sparse.mm(B, Ff) if B.is_sparse else B @ Ff
    return out.reshape(-1, 3, 3)


def denormalize(points: Tensor, size: Tensor, center: Tensor) -> Tensor:
    """modules/nclaw/utils.py:110-118: (x - center) / size."""
    return (points - center) / size


def l1_loss(a, b):
    """modules/d3gs/utils/loss_utils.py:17-18"""
    return torch.abs(a - b).mean()


def l2_loss(a, b):
    """modules/d3gs/utils/loss_utils.py:23-24"""
    return ((a - b) ** 2).mean()


# ---------------------------------------------------------------- camera helpers (conventions only)
def look_at_camera(eye, target, up, fovx: float, fovy: float, znear=0.01, zfar=100.0, dtype=torch.float32):
    """Build (viewmatrix, projmatrix=full_proj, campos) in the reference convention
    (cameras.py:54-57; graphics_utils.py:38-71): +x right, +y down, +z forward, row-vector."""
    eye = torch.as_tensor(eye, dtype=torch.float64)
    target = torch.as_tensor(target, dtype=torch.float64)
    up = torch.as_tensor(up, dtype=torch.float64)
    fwd = target - eye
    fwd = fwd / fwd.norm()
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    Rw2c = torch.stack([right, down, fwd], 0)                # rows: camera axes in world
    tvec = -Rw2c @ eye
    Rt = torch.eye(4, dtype=torch.float64)
    Rt[:3, :3] = Rw2c
    Rt[:3, 3] = tvec
    world_view = Rt.transpose(0, 1)
    tanx, tany = math.tan(fovx / 2), math.tan(fovy / 2)
    P = torch.zeros(4, 4, dtype=torch.float64)
    P[0, 0] = 1.0 / tanx
    P[1, 1] = 1.0 / tany
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    proj = P.transpose(0, 1)
    full = world_view @ proj
    campos = torch.linalg.inv(world_view)[3, :3]
    return world_view.to(dtype), full.to(dtype), campos.to(dtype)
