"""Particle -> Gaussian binding construction on the GPU.

Counterpart of /root/reference/modules/d3gs/utils/binding_utils.py (gaussian_binding 123-196,
gaussian_binding_with_clip_v1 199-285) and of the binding part of prepare_simulation_data
(modules/tune/utils.py:268-317).  The reference loops over the K Gaussians on the host, tests every particle against each
(K launches, O(K N)) and fills a dense K x N fp32 matrix (hence its `< INT_MAX` assert, :297); here one kernel visits only
the grid cells under each Gaussian's confidence ellipsoid and the result is born sparse.
"""
from pathlib import Path
from typing import Optional, Tuple

import ctypes as C
import numpy as np
import torch
from torch import Tensor

from . import _lib as L


def chi2_threshold(confidence: float) -> float:
    """chi2.ppf(confidence, 3) (binding_utils.py:174)"""
    from scipy.stats import chi2
    return float(chi2.ppf(confidence, 3))


def _grid_for(particles: Tensor, max_cells: int = 1 << 21) -> Tuple[np.ndarray, float, np.ndarray]:
    """Uniform grid over the particles' bounding box: about two particles per cell, at most max_cells cells."""
    lo = particles.min(0).values.double().cpu().numpy()
    hi = particles.max(0).values.double().cpu().numpy()
    ext = np.maximum(hi - lo, 1e-9)
    n = max(int(particles.shape[0]), 1)
    h = float((np.prod(ext) * 2.0 / n) ** (1.0 / 3.0))
    h = max(h, float(ext.max()) / 1024.0, 1e-9)
    while True:
        dims = np.maximum(np.ceil(ext / h).astype(np.int64) + 1, 1)
        if int(np.prod(dims)) <= max_cells:
            break
        h *= 1.26
    return lo.astype(np.float32), h, dims.astype(np.int32)


def build_bindings(means: Tensor, cov6: Tensor, particles: Tensor, confidence: float = 0.95, max_particles: int = 10,
                   return_distances: bool = False):
    """-> (counts (K,) int32, n_inside (K,) int32, cols (K, max_particles) int32 [-1 padded], [pvals])."""
    lib = L.lib()
    dev = means.device
    m = means.detach().float().contiguous()
    c = cov6.detach().float().reshape(-1, 6).contiguous()
    x = particles.detach().float().contiguous()
    K, N = int(m.shape[0]), int(x.shape[0])
    origin, h, dims = _grid_for(x) if N > 0 else (np.zeros(3, np.float32), 1.0, np.ones(3, np.int32))
    ncells = int(np.prod(dims.astype(np.int64)))
    ws_bytes = int(lib.nm_bind_build_workspace(N, ncells))
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
    counts = torch.zeros(K, dtype=torch.int32, device=dev)
    inside = torch.zeros(K, dtype=torch.int32, device=dev)
    cols = torch.full((K, max_particles), -1, dtype=torch.int32, device=dev)
    pv = torch.zeros(K, max_particles, dtype=torch.float32, device=dev) if return_distances else None
    L.check(lib.nm_bind_build(K, N, L.ptr(m), L.ptr(c), L.ptr(x), (C.c_float * 3)(*origin.tolist()), float(h),
                              (C.c_int32 * 3)(*dims.tolist()), chi2_threshold(confidence), int(max_particles), L.ptr(counts),
                              L.ptr(inside), L.ptr(cols), L.ptr(pv) if pv is not None else None, L.ptr(ws), ws_bytes,
                              L.stream_ptr(dev)), "nm_bind_build")
    return (counts, inside, cols, pv) if return_distances else (counts, inside, cols)


def _to_coo(counts: Tensor, cols: Tensor, N: int):
    K, maxp = cols.shape
    valid = cols >= 0
    rows = torch.arange(K, device=cols.device).unsqueeze(1).expand(K, maxp)[valid]
    c = cols[valid].long()
    w = (1.0 / counts.clamp(min=1).float()).unsqueeze(1).expand(K, maxp)[valid]
    return torch.sparse_coo_tensor(torch.stack([rows, c], 0), w, (K, N)).coalesce()


def gaussian_binding(gaussians, particles: Tensor, confidence: float = 0.95, max_particles: int = 10) -> Tensor:
    """binding_utils.py:123-196 as a sparse bool-like matrix: (K x N) COO with value 1 where particle j lies inside
    Gaussian k's confidence ellipsoid (at most max_particles nearest per Gaussian).  `.to_dense().bool()` gives the
    reference's flag_mat; `n_inside` (attribute `_nm_inside`) holds the unclipped counts."""
    counts, inside, cols = build_bindings(gaussians.get_xyz, gaussians.get_covariance(), particles, confidence, max_particles)
    B = _to_coo(counts, cols, int(particles.shape[0]))
    out = torch.sparse_coo_tensor(B.indices(), torch.ones_like(B.values()), B.size()).coalesce()
    out._nm_inside = inside
    return out


def gaussian_binding_with_clip_v1(gaussians, particles: Tensor, confidence: float = 0.95, max_particles: int = 10) -> Tensor:
    """binding_utils.py:199-285 as a sparse COO weight matrix (the reference's dense result .to_sparse_coo(), which is
    what its caller does next, tune/utils.py:298).  Raises like the reference's `assert weight.sum() != 0` when a
    Gaussian ends up without particles."""
    counts, inside, cols = build_bindings(gaussians.get_xyz, gaussians.get_covariance(), particles, confidence, max_particles)
    empty = int((counts == 0).sum())
    if empty:
        raise AssertionError(f"{empty} Gaussians have no particle inside their {confidence:.2f} confidence ellipsoid "
                             "(add their centres as particles first: prepare_bindings does)")
    return _to_coo(counts, cols, int(particles.shape[0]))


def prepare_bindings(gaussians, particles: Tensor, confidence: float = 0.95, max_particles: int = 10,
                     particles_downsample_factor: int = 1, save_dir: Optional[Path] = None, generator=None):
    """Binding part of prepare_simulation_data (tune/utils.py:268-317): optional random down-sampling of the particles,
    a first pass to find Gaussians without any particle, whose centres are appended as extra particles, the final
    clipped binding, and (optionally) particles.ply + bindings.pt in the reference's layout.
    Returns (particles (N',3), bindings sparse COO (K,N'), n_particles (K,))."""
    if particles_downsample_factor > 1:
        perm = torch.randperm(particles.shape[0], generator=generator, device="cpu").to(particles.device)
        particles = particles[perm][::particles_downsample_factor].contiguous()
    pre_counts, _, _ = build_bindings(gaussians.get_xyz, gaussians.get_covariance(), particles, confidence, max_particles)
    lonely = pre_counts == 0
    particles = torch.cat([particles, gaussians.get_xyz[lonely].detach()], 0).contiguous()
    B = gaussian_binding_with_clip_v1(gaussians, particles, confidence, max_particles)
    n_particles = torch.bincount(B.indices()[0], minlength=B.size(0))
    if save_dir is not None:
        from . import io as nio
        save_dir = Path(save_dir)
        nio.save_particles_ply(save_dir / "particles.ply", particles.cpu().numpy())
        nio.save_bindings(save_dir / "bindings.pt", B.indices(), B.values(), B.size(), n_particles)
    return particles, B, n_particles
