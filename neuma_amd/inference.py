"""`python -m neuma_amd.inference -c <config.yaml> -vn <name> [-s N] [-dv view ...] [-sp folder]` - the config-driven multi-object
forward roll-out + renderer, counterpart of /root/reference/experiments/inference.py (args 48-84, `eval` 87-377, main 380-386)
for the `experiments/configs/demo/*.yaml` schema (multiobj-bb-cc.yaml, generalize-*.yaml):

    seeds, device, background (white only if video_data.data.white_background)                              :91-113
    <result_root>/inference/images_<video_name>/, <result_root>/inference_states/states_<save_particles>/  :117-125
    one MPMModel for the scene (cfg.sim), CameraDataset (cameras only; --dataset_path / --debug_views)      :129-143
    per entry of cfg.objects                                                                                :159-254
        assets in <assets_root>/<sim_data_name>/ (prepare_simulation_data when particles_path / mesh_path is given)
        bindings.pt, kernels.ply, scaling_modifier; the constitutive pair from `pretrained_ckpt` (+ `constitution.load_lora`
        with lora.{r, alpha}); MPMInitData with span [0, eval_steps] and `particle_data.vel.{lin_vel, ang_vel}`
    ComposeMaterial over the sections, Gaussians mapped into the simulation box unless `denormalize`        :256-281
    frame 0 from the un-deformed kernels, then per step the loop of infer.simulate_objects                  :283-362
    -> <view>_<step:03d>.png per debug view, <first_step + step:03d>.ply per step when --save_particles

The numeric work is infer.simulate_objects (the operators of SURVEY 8 f3) on the HIP kernels.  Not reproduced: packing the
frames into an mp4 (mediapy; --skip_frames / --remove_images are accepted, the latter removes the frames) and the random
initial velocity of an object without `particle_data.vel` - the reference's own fallback `sample_vel(seed=42)` raises (it
needs a cfg with lin_vel_bound / ang_vel_bound, nclaw/utils.py:15-30): here such an object samples when its `particle_data`
carries those bounds and raises otherwise."""
import argparse
import random
import shutil
import sys
from pathlib import Path

import numpy as np
import torch

from . import io as nio
from .config import Cfg, load_config
from .dataset import CameraDataset
from .evaluate import save_image
from .infer import SceneObject, simulate_objects
from .material import InvariantFullMetaElasticity, InvariantFullMetaPlasticity
from .prepare import prepare_simulation_data
from .sim import MPMInitData, MPMModelBuilder

RESULT = "results"


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--config", "-c", type=str, required=True, help="Path to the config file.")
    p.add_argument("--eval_steps", "-s", type=int, default=600, help="Number of simulation steps.")
    p.add_argument("--skip_frames", "-f", type=int, default=1, help="Number of skip frames when packing the video.")
    p.add_argument("--remove_images", "-ri", action="store_true", help="Whether to remove images after packing video.")
    p.add_argument("--video_name", "-vn", type=str, required=True, help="Save video name.")
    p.add_argument("--debug_views", "-dv", nargs="+", default=[], help="Views for rendering.")
    p.add_argument("--save_particles", "-sp", type=str, default=None, help="Specify the folder name for saving simulated particles.")
    p.add_argument("--dataset_path", type=str, default=None, help="Rewrite video dataset path.")
    p.add_argument("--result_root", type=str, default=RESULT)
    return p.parse_args(argv)


def sample_vel(cfg, seed=None):
    """nclaw/utils.py:15-30: a random downward linear velocity of magnitude in cfg.lin_vel_bound and an angular velocity with
    components in cfg.ang_vel_bound, from numpy's PCG64 stream of `seed`."""
    if seed is None:
        seed = cfg.seed
    rng = np.random.Generator(np.random.PCG64(seed))
    lin_dir = rng.uniform(-1, 1, size=3)
    if lin_dir[1] > 0:
        lin_dir[1] = -lin_dir[1]
    lin_dir /= np.linalg.norm(lin_dir)
    lin_vel = lin_dir * rng.uniform(*cfg.lin_vel_bound)
    ang_vel = rng.uniform(*cfg.ang_vel_bound, size=3)
    return lin_vel, ang_vel


def load_object(obj_cfg: Cfg, assets: Path, eval_steps: int, device) -> SceneObject:
    """One entry of cfg.objects -> SceneObject (inference.py:159-254)."""
    data_root = assets / obj_cfg.sim_data_name
    print(f"\nLoad data for {obj_cfg.sim_data_name} ...")
    pd, gc = obj_cfg.particle_data, obj_cfg.gaussian
    if pd.get("particles_path") is not None or pd.get("mesh_path") is not None:
        data_root.mkdir(parents=True, exist_ok=True)
        common = dict(save_dir=data_root, kernels_path=Path(gc.kernels_path), sh_degree=gc.sh_degree, opacity_thres=gc.opacity_thres,
                      confidence=gc.confidence, max_particles=gc.max_particles, device=device)
        if pd.get("particles_path") is not None:
            prepare_simulation_data(particles_path=Path(pd.particles_path), particles_downsample_factor=pd.downsample_factor, **common)
        else:
            prepare_simulation_data(mesh_path=Path(pd.mesh_path), mesh_sample_mode=pd.mesh_sample_mode,
                                    mesh_sample_resolution=pd.mesh_sample_resolution, particles_downsample_factor=1, **common)
    bindings, n_particles = nio.load_bindings(data_root / "bindings.pt", device=device)
    print(f"#Gaussians with particle bindings: {int((n_particles > 0).sum())}")
    print(f"#Avg particles: {float(n_particles.mean())}")
    print(f"#Max particles: {float(n_particles.max())}, index: {int(torch.argmax(n_particles))}")
    gaussians = nio.load_gaussians_ply(data_root / "kernels.ply", gc.sh_degree, device=device)
    cc = obj_cfg.constitution
    elasticity = InvariantFullMetaElasticity(cc.elasticity).to(device)
    plasticity = InvariantFullMetaPlasticity(cc.plasticity).to(device)
    pretrained = torch.load(obj_cfg.pretrained_ckpt, map_location=device)
    elasticity.load_state_dict(pretrained["elasticity"])
    plasticity.load_state_dict(pretrained["plasticity"])
    print(f"Loaded pretrained weights from {obj_cfg.pretrained_ckpt}")
    if cc.get("load_lora") is not None:
        elasticity.init_lora_layers(r=cc.lora.r, lora_alpha=cc.lora.alpha)
        plasticity.init_lora_layers(r=cc.lora.r, lora_alpha=cc.lora.alpha)
        lora = torch.load(cc.load_lora, map_location=device)
        elasticity.load_state_dict(lora["elasticity"], strict=False)
        plasticity.load_state_dict(lora["plasticity"], strict=False)
        elasticity.to(device); plasticity.to(device)
        print(f"Loaded lora weights from {cc.load_lora}")
    pd.span = [0, eval_steps]                                   # NOTE: manually setting (inference.py:234)
    pd.shape.name = obj_cfg.sim_data_name + "/particles"        # NOTE: manually setting (inference.py:235)
    if pd.shape.get("asset_root") is None:
        pd.shape.asset_root = str(assets)
    init_data = MPMInitData.get(pd)
    if pd.get("vel") is not None:
        print(f"Use initial velocity: {dict(pd.vel)} ...")
        lin_vel, ang_vel = np.array(pd.vel.lin_vel, dtype=np.float64), np.array(pd.vel.ang_vel, dtype=np.float64)
    elif pd.get("lin_vel_bound") is not None and pd.get("ang_vel_bound") is not None:
        print("Randomly sample initial velocity ...")
        lin_vel, ang_vel = sample_vel(pd, seed=42)
    else:
        raise ValueError(f"object {obj_cfg.sim_data_name!r}: particle_data.vel (lin_vel, ang_vel) is required - the reference's fallback "
                         "sample_vel(seed=42) needs lin_vel_bound / ang_vel_bound (nclaw/utils.py:15-30), which this entry does not carry")
    init_data.set_lin_vel(lin_vel)
    init_data.set_ang_vel(ang_vel)
    return SceneObject(init_data=init_data, elasticity=elasticity, plasticity=plasticity, gaussians=gaussians, bindings=bindings,
                       scaling=float(gc.get("scaling_modifier", 1.0)))


@torch.no_grad()
def inference(cfg: Cfg, on_frame=None):
    """inference.py:87-377 (`eval`).  Returns the image folder."""
    seed = cfg.seed
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    device = torch.device(f"cuda:{cfg.gpu}")
    torch.cuda.set_device(device)
    background = torch.tensor([1.0, 1.0, 1.0] if cfg.video_data.data.get("white_background", False) else [0.0, 0.0, 0.0], device=device)
    root = Path(cfg.get("result_root", RESULT))
    image_root = root / "inference" / f"images_{cfg.video_name}"
    image_root.mkdir(exist_ok=True, parents=True)
    (root / "inference_videos").mkdir(exist_ok=True)
    state_root = None
    if cfg.get("save_particles") is not None:
        state_root = root / "inference_states" / f"states_{cfg.save_particles}"
        state_root.mkdir(parents=True, exist_ok=True)
    eval_steps = int(cfg.eval_steps)
    model = MPMModelBuilder().parse_cfg(cfg.sim).finalize(device, False)          # the YAML's own sim.eps (no override here)
    if cfg.get("dataset_path") is not None:
        cfg.video_data.data.path = cfg.dataset_path
        print(f"Rewrite video dataset path to\n\t{cfg.dataset_path}")
    debug_views = list(cfg.get("debug_views") or [])
    if len(debug_views) > 0:
        cfg.video_data.data.used_views = debug_views
    cfg.video_data.device = str(device)
    dataset = CameraDataset(cfg.video_data)
    first_step = dataset.steps[0]
    assets = Path(cfg.get("assets_root", "experiments/assets"))
    objects = [load_object(o, assets, eval_steps, device) for o in cfg.objects]
    views = [vw for vw in dataset.views if vw in debug_views]
    cameras = [dataset.getCameras(vw, first_step) for vw in views]               # the camera of the FIRST step throughout
    for frame in simulate_objects(model, objects, eval_steps, cameras, background, denormalize=bool(cfg.get("denormalize", False))):
        step = frame["step"]
        for vw, img in zip(views, frame["images"]):
            save_image(img, image_root / f"{vw}_{step:03d}.png")
        if state_root is not None and step > 0:
            nio.save_particles_ply(state_root / f"{first_step + step:03d}.ply", frame["x"].detach().cpu().numpy())
        if on_frame is not None:
            on_frame(step, frame)
    if cfg.get("remove_images"):
        shutil.rmtree(image_root, ignore_errors=True)
    return image_root


def main(argv=None):
    args = parse_args(argv)
    cfg = load_config(args.config)
    for k, val in vars(args).items():               # cfg.update(vars(args)), inference.py:383: the command line wins, also an empty -dv
        if k != "config":
            cfg[k] = val
    inference(cfg)


if __name__ == "__main__":
    sys.exit(main())
