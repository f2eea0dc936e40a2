"""neuma_amd — MI355X (gfx950) engine behind NeuMA's simulator / constitutive-net / Particle-GS operators.

Python here is plumbing (tensor memory, streams, autograd wiring); the arithmetic lives in
neuma_amd/csrc/*.hip behind the C ABI of include/neuma_hip.h.  Sub-packages mirror the reference layout:

    neuma_amd.sim        <- modules/nclaw/sim        (MPMModelBuilder, MPMModel, MPMState, MPM*Sim ...)
    neuma_amd.svd        <- modules/nclaw/warp/svd.py (SVD)
    neuma_amd.material   <- modules/nclaw/material   (InvariantFullMeta{Elasticity,Plasticity}, ComposeMaterial, LoRA)
    neuma_amd.render     <- diff_gaussian_rasterization + modules/d3gs/gaussian_renderer (GaussianRasterizer ...)
    neuma_amd.tune       <- modules/tune/utils.py    (diff_rasterization, compute_bindings_xyz / _F)
    neuma_amd.rollout    fused S-substep roll-out with checkpointed BPTT (fast path)
"""
from ._lib import NeumaHipError, lib  # noqa: F401

__version__ = "0.1.0"
