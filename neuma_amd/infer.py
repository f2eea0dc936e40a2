"""Forward-only multi-object driver: counterpart of /root/reference/experiments/inference.py:225-362 (and of the
single-object render.py:eval 260-332).

Order of one step, as the reference runs it (inference.py:309-314):
    stress = elasticity(F); state.from_torch(stress=stress)
    x, v, C, F = sim(statics, state)            # MPMForwardSim: in place on `state`
    F = plasticity(F); state.from_torch(F=F)
    statics_initializer.update(statics, step)   # span-based enabling takes effect AFTER the step
then, per object: particle positions (optionally de-normalised per object, nclaw/utils.py:121-135) and F are split by
`sections`, bound to the object's Gaussians with positions of the PREVIOUS rendered frame (tune/utils.py:475-523) and the
concatenated set is rasterised once per view with the camera of the first step.  The first frame is rendered from the
un-deformed Gaussians (deform_grad=None, inference.py:285-298).

Everything numeric runs in the HIP kernels; ComposeMaterial (material/preset.py) dispatches each section to its own net.
"""
from dataclasses import dataclass
from typing import Callable, Dict, Iterator, List, Optional, Sequence

import torch
import torch.nn as nn

from .material import ComposeMaterial
from .render.transform_utils import scale_gaussians, translate_gaussians
from .sim import MPMForwardSim, MPMStateInitializer, MPMStaticsInitializer
from .sim.mpm import MPMInitData, MPMModel
from .tune import denormalize_points_helper_func, diff_rasterization, preprocess_for_rasterization


@dataclass
class SceneObject:
    """One simulated body: particles + statics (MPMInitData), its constitutive pair, its Gaussians and bindings."""
    init_data: MPMInitData
    elasticity: nn.Module
    plasticity: nn.Module
    gaussians: object            # GaussianModel
    bindings: object             # tune.Bindings or a torch sparse COO tensor (K x N_obj)
    scaling: float = 1.0         # scaling_modifier of the covariances (inference.py obj_scalings)


def denormalize_points(points: torch.Tensor, sections: Sequence[int], state_init) -> torch.Tensor:
    """nclaw/utils.py:121-135"""
    out = []
    for gd, gx in zip(state_init.groups, torch.split(points, list(sections), dim=0)):
        out.append(denormalize_points_helper_func(gx, gd.size, gd.center))
    return torch.cat(out, dim=0)


@torch.no_grad()
def simulate_objects(model: MPMModel, objects: List[SceneObject], eval_steps: int, cameras: Sequence, background: torch.Tensor,
                     denormalize: bool = False, on_frame: Optional[Callable[[int, Dict], None]] = None,
                     render: bool = True) -> Iterator[Dict]:
    """Generator over frames 0..eval_steps.  Yields {'step', 'x', 'F', 'means3D', 'images': [per camera]}."""
    device = model.device
    state_initializer = MPMStateInitializer(model)
    statics_initializer = MPMStaticsInitializer(model)
    for o in objects:
        state_initializer.add_group(o.init_data)
        statics_initializer.add_group(o.init_data)
    state, sections = state_initializer.finalize()
    statics = statics_initializer.finalize()
    x, v, C, F, stress = state.to_torch()
    elasticity = ComposeMaterial([o.elasticity for o in objects], sections).to(device).eval()
    plasticity = ComposeMaterial([o.plasticity for o in objects], sections).to(device).eval()
    sim = MPMForwardSim(model)
    gs = [o.gaussians for o in objects]
    scal = [o.scaling for o in objects]
    sec_gaussians = [int(g.get_xyz.shape[0]) for g in gs]
    if not denormalize:                                                # inference.py:275-281
        for g, gd in zip(gs, state_initializer.groups):
            scale_gaussians(g, float(gd.size[0]), torch.zeros(3, device=device))
            translate_gaussians(g, torch.as_tensor(gd.center, dtype=torch.float32, device=device))
    sh_deg = gs[0].active_sh_degree
    first = dict(step=0, x=x.clone(), F=F.clone(), means3D=torch.cat([g.get_xyz for g in gs], 0), images=[])
    if render:
        cov = torch.cat([g.get_covariance(s) for g, s in zip(gs, scal)], 0)
        opa = torch.cat([g.get_opacity for g in gs], 0)
        shs = torch.cat([g.get_features for g in gs], 0)
        for cam in cameras:
            first["images"].append(diff_rasterization(first["means3D"], None, None, cam, background, sh_deg, cov, opa, shs))
    if on_frame:
        on_frame(0, first)
    yield first
    de_x = denormalize_points(x, sections, state_initializer) if denormalize else x
    p_prev = [t.clone().detach() for t in torch.split(de_x, sections, dim=0)]
    k_prev = [g.get_xyz.clone().detach() for g in gs]
    bindings = [o.bindings for o in objects]
    for step in range(1, eval_steps + 1):
        stress = elasticity(F)
        state.from_torch(stress=stress)
        x, v, C, F = sim(statics, state)
        F = plasticity(F)
        state.from_torch(F=F)
        statics_initializer.update(statics, step)
        de_x = denormalize_points(x, sections, state_initializer) if denormalize else x
        p_curr = list(torch.split(de_x, sections, dim=0))
        dgs = list(torch.split(F, sections, dim=0))
        pack = preprocess_for_rasterization(obj_gaussians=gs, obj_deform_grad=dgs, obj_kernels_prev=k_prev, obj_particles_curr=p_curr,
                                            obj_particles_prev=p_prev, obj_bindings=bindings, obj_scalings=scal)
        frame = dict(step=step, x=x.clone(), F=F.clone(), means3D=pack["means3D"], images=[])
        if render:
            for cam in cameras:
                frame["images"].append(diff_rasterization(pack["means3D"], pack["deform_grad"], None, cam, background,
                                                          pack["active_sh_degree"], pack["cov3D"], pack["opacity"], pack["shs"]))
        if on_frame:
            on_frame(step, frame)
        yield frame
        p_prev = [t.clone().detach() for t in torch.split(de_x, sections, dim=0)]
        k_prev = [t.clone().detach() for t in torch.split(pack["means3D"], sec_gaussians, dim=0)]
