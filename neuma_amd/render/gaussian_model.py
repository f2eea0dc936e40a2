"""Minimal 3DGS parameter container: the accessors NeuMA's hot path reads.
Mirrors /root/reference/modules/d3gs/scene/gaussian_model.py: activations 26-41, get_* 97-118
(get_xyz, get_features, get_opacity, get_covariance), with the covariance cached per scaling_modifier
(the reference recomputes it on every render call although it is constant).
Densification / optimizer plumbing / simple_knn are out of scope (never reached by NeuMA drivers)."""
import torch
from torch import Tensor

from . import build_cov3D


class GaussianModel(object):
    def __init__(self, sh_degree: int):
        self.active_sh_degree = sh_degree
        self.max_sh_degree = sh_degree
        self._xyz = torch.empty(0)
        self._features_dc = torch.empty(0)
        self._features_rest = torch.empty(0)
        self._scaling = torch.empty(0)
        self._rotation = torch.empty(0)
        self._opacity = torch.empty(0)
        self._cov_cache = {}

    def set_params(self, xyz: Tensor, features_dc: Tensor, features_rest: Tensor, scaling: Tensor, rotation: Tensor,
                   opacity: Tensor) -> "GaussianModel":
        """xyz (K,3); features_dc (K,1,3); features_rest (K,(deg+1)^2-1,3); scaling = log-scales (K,3);
        rotation = quaternions (K,4) (r,x,y,z); opacity = logits (K,1) — the PLY-side parametrisation."""
        self._xyz, self._features_dc, self._features_rest = xyz, features_dc, features_rest
        self._scaling, self._rotation, self._opacity = scaling, rotation, opacity
        self._cov_cache = {}
        return self

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        key = float(scaling_modifier)
        if key not in self._cov_cache:
            with torch.no_grad():
                self._cov_cache[key] = build_cov3D(self.get_scaling, self._rotation, key).contiguous()
        return self._cov_cache[key]
