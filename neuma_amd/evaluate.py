"""Forward-only roll-out + rendering of a fine-tuned experiment: counterpart of /root/reference/experiments/render.py:109-347
(`eval`), reachable as `python -m neuma_amd.render -c <config.yaml> -vn <name> [-es N] [-l 0100_lora.pt] [-dv view ...]`.

Per step (render.py:304-332): stress = E(F) -> state.from_torch(stress) -> in-place MPMForwardSim -> F = P(F) ->
state.from_torch(F) -> statics_initializer.update(statics, step) -> de-normalise, bind with the PREVIOUS frame's positions,
render the debug views with the camera of the FIRST step -> <result>/<name>/images_<video_name>/<view>_<frame:03d>.png.
The YAML's own sim.eps is used here (the reference only overrides it in finetune.py).  Packing the frames into an mp4
(mediapy) is left to external tools."""
import argparse
import random
from pathlib import Path

import numpy as np
import torch

from . import io as nio
from .config import Cfg, load_config
from .finetune import particle_init_data, setup
from .sim import MPMForwardSim, MPMModelBuilder, MPMStateInitializer, MPMStaticsInitializer
from .tune import compute_bindings_F, compute_bindings_xyz, denormalize_points_helper_func, diff_rasterization

RESULT = "experiments/results"


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--config", "-c", type=str, required=True)
    p.add_argument("--eval_steps", "-es", type=int, default=400)
    p.add_argument("--init_frame", "-if", type=int, default=None)
    p.add_argument("--skip_frames", "-sf", type=int, default=1)
    p.add_argument("--load_lora", "-l", type=str, default=None)
    p.add_argument("--video_name", "-vn", type=str, required=True)
    p.add_argument("--sim_dt", "-dt", type=float, default=None)
    p.add_argument("--debug_views", "-dv", nargs="+", default=[])
    p.add_argument("--save_particles", "-sp", type=str, default=None)
    p.add_argument("--change_base_model", "-cbm", type=str, default=None)
    p.add_argument("--dataset_path", type=str, default=None)
    p.add_argument("--transform_file", type=str, default=None)
    p.add_argument("--alpha", type=float, default=None)
    p.add_argument("--result_root", type=str, default=RESULT)
    return p.parse_args(argv)


def save_image(img: torch.Tensor, path) -> None:
    """torchvision.utils.save_image for one (3,H,W) image in [0,1].  No image leaves the GPU unchecked: a render whose bin lists
    overflowed (incomplete image) raises here (render.flush_pending)."""
    from .render import flush_pending
    flush_pending()
    from PIL import Image
    a = (img.detach().clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    Image.fromarray(a, "RGB").save(path)


@torch.no_grad()
def evaluate(cfg: Cfg, on_frame=None):
    seed = cfg.seed
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    device = torch.device(f"cuda:{cfg.gpu}")
    torch.cuda.set_device(device)
    background = torch.tensor([1.0, 1.0, 1.0] if cfg.video_data.data.get("white_background", False) else [0.0, 0.0, 0.0], device=device)
    if cfg.get("alpha") is not None:
        cfg.constitution.lora.alpha = cfg.alpha
    exp_root = Path(cfg.root) / cfg.name
    assert exp_root.exists(), f"Experiment {exp_root} does not exist."
    tune_root = exp_root / "finetune"
    image_root = Path(cfg.get("result_root", RESULT)) / cfg.name / f"images_{cfg.video_name}"
    image_root.mkdir(exist_ok=True, parents=True)
    state_root = None
    if cfg.get("save_particles") is not None:
        state_root = Path(cfg.get("result_root", RESULT)) / cfg.name / f"states_{cfg.save_particles}"
        state_root.mkdir(parents=True, exist_ok=True)
    cfg.video_data.data.init_frame = cfg.get("init_frame")               # NOTE: manually setting (render.py:186-188)
    if cfg.get("debug_views"):
        cfg.video_data.data.used_views = list(cfg.debug_views)
    if cfg.get("dataset_path") is not None:
        cfg.video_data.data.path = cfg.dataset_path
    if cfg.get("transform_file") is not None:
        cfg.video_data.data.transformsfile = cfg.transform_file
    env = setup(cfg, device, for_eval=True)
    dataset, gaussians, bindings = env["dataset"], env["gaussians"], env["bindings"]
    E, P = env["elasticity"].eval().requires_grad_(False), env["plasticity"].eval().requires_grad_(False)
    first_step = dataset.steps[0]
    ix, iv = nio.load_init_state(tune_root / "init.pt")
    dataset.set_init_x_and_v(init_x=ix, init_v=iv)
    lora_name = cfg.get("load_lora") or cfg.constitution.get("load_lora")
    if lora_name is not None:
        E.init_lora_layers(r=cfg.constitution.lora.r, lora_alpha=cfg.constitution.lora.alpha)
        P.init_lora_layers(r=cfg.constitution.lora.r, lora_alpha=cfg.constitution.lora.alpha)
        lora = torch.load(tune_root / lora_name, map_location=device)
        E.load_state_dict(lora["elasticity"], strict=False)
        P.load_state_dict(lora["plasticity"], strict=False)
        E.to(device); P.to(device)
        print(f"Loaded lora weights from {tune_root / lora_name}")
    eval_steps = int(cfg.eval_steps)
    if cfg.get("sim_dt") is not None:
        cfg.sim.dt = cfg.sim_dt
    model = MPMModelBuilder().parse_cfg(cfg.sim).finalize(device, False)
    sim = MPMForwardSim(model)
    state_initializer, statics_initializer = MPMStateInitializer(model), MPMStaticsInitializer(model)
    init_data = particle_init_data(cfg, eval_steps)
    state_initializer.add_group(init_data); statics_initializer.add_group(init_data)
    state, _ = state_initializer.finalize()
    statics = statics_initializer.finalize()
    assert init_data.pos.shape[0] == dataset.get_init_x.shape[0], \
        f"Shape mismatch: init_data {init_data.pos.shape[0]} dataset {dataset.get_init_x.shape[0]}"
    x, v, C, F, _ = dataset.get_init_material_data()
    state.from_torch(x=x, v=v, C=C, F=F)
    views = [vw for vw in dataset.views if vw in cfg.get("debug_views", [])]
    scal = cfg.gaussian.get("scaling_modifier", 1.0)
    for vw in views:                                                       # first frame: un-deformed kernels (render.py:292-297)
        render = diff_rasterization(gaussians.get_xyz, None, gaussians, dataset.getCameras(vw, first_step), background, scaling_modifier=scal)
        save_image(render, image_root / f"{vw}_{first_step:03d}.png")
    de_x = denormalize_points_helper_func(x, init_data.size, init_data.center)
    de_x_prev, g_prev = de_x.clone().detach(), gaussians.get_xyz.clone().detach()
    for step in range(1, eval_steps + 1):
        stress = E(F)
        state.from_torch(stress=stress)
        x, v, C, F = sim(statics, state)
        F = P(F)
        state.from_torch(F=F)
        statics_initializer.update(statics, step)
        de_x = denormalize_points_helper_func(x, init_data.size, init_data.center)
        means3D = compute_bindings_xyz(de_x, de_x_prev, g_prev, bindings)
        deform_grad = compute_bindings_F(F, bindings)
        images = {}
        for vw in views:
            images[vw] = diff_rasterization(means3D, deform_grad, gaussians, dataset.getCameras(vw, first_step), background, scaling_modifier=scal)
            save_image(images[vw], image_root / f"{vw}_{first_step + step:03d}.png")
        if state_root is not None:
            nio.save_particles_ply(state_root / f"{first_step + step:03d}.ply", x.detach().cpu().numpy())
        if on_frame is not None:
            on_frame(step, dict(x=x, F=F, means3D=means3D, images=images))
        de_x_prev, g_prev = de_x.clone().detach(), means3D.clone().detach()
    return image_root


def main(argv=None):
    args = parse_args(argv)
    cfg = load_config(args.config)
    for k, val in vars(args).items():
        if k != "config":
            cfg[k] = val
    evaluate(cfg)
