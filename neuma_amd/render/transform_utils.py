"""In-place rigid edits of a Gaussian set: /root/reference/modules/d3gs/utils/transform_utils.py:107-137
(translate_gaussians, scale_gaussians).  rotate_gaussians (SH rotation through e3nn Wigner matrices, :12-104) is never
called by the NeuMA drivers and is not provided."""
from typing import Optional, Union

import torch


def _touch(gaussians) -> None:
    cache = getattr(gaussians, "_cov_cache", None)
    if cache is not None:
        cache.clear()          # covariances depend on _scaling


def translate_gaussians(gaussians, translation: torch.Tensor) -> None:
    """transform_utils.py:107-116"""
    assert translation.shape == (3,), f"Translation vector must have shape (3,), but got {translation.shape}."
    gaussians._xyz = gaussians._xyz + translation.unsqueeze(0).to(gaussians._xyz)


def scale_gaussians(gaussians, scale: Union[torch.Tensor, float], origin: Optional[torch.Tensor] = None) -> None:
    """transform_utils.py:119-137: positions scaled about `origin` (default: centroid) AND MOVED TO IT (the reference
    assigns scale * (xyz - origin), without adding the origin back), log-scales shifted by log(scale)."""
    if isinstance(scale, (float, int)):
        scale = torch.tensor(float(scale), device=gaussians.get_xyz.device, dtype=torch.float32)
    elif isinstance(scale, torch.Tensor):
        assert scale.shape == (1,) or scale.shape == (), f"Scale factor must have shape (1,) or (), but got {scale.shape}."
        scale = scale.to(gaussians.get_xyz)
    else:
        raise ValueError(f"Scale factor must be a torch.Tensor or a float, but got {type(scale)}.")
    if origin is None:
        origin = torch.mean(gaussians.get_xyz, dim=0, keepdim=True)
    gaussians._xyz = scale * (gaussians.get_xyz - origin.to(gaussians.get_xyz))
    gaussians._scaling = gaussians._scaling + torch.log(scale)
    _touch(gaussians)
