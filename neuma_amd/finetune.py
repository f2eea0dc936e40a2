"""`python -m neuma_amd.finetune -c <config.yaml>` - the fine-tuning entry point, driven by the reference's YAML schema and
on-disk dataset layout.  Counterpart of /root/reference/experiments/finetune.py:491-663 (finetune) around the two stages
implemented in neuma_amd.train:

    seeds, device, background (white only if video_data.data.white_background; mask data forces black)      :496-522
    <root>/<name>/ with config.yaml + finetune/ ; assets in <assets_root>/<sim_data_name>/                 :524-545
    prepare_simulation_data -> kernels.ply, particles.ply, bindings.pt                                      :549-579
    VideoDataset, bindings, GaussianModel.load_ply, nets + `pretrained_ckpt`                                 :581-623
    init.pt found in the asset folder is linked into finetune/                                              :627-633
    stage A  optimize_init_velocity  (cfg.velocity)     -> finetune/init.pt                                 :635-647
    stage B  finetune_constitutive   (cfg.constitution) -> finetune/{epoch:04d}_lora.pt                     :651-663

Keys consumed: SURVEY.md App. F.  `assets_root` (default experiments/assets, the reference's ASSETS_PATH) is the one
addition.  Logging to tensorboard / debug image dumps are not reproduced."""
import argparse
import random
import sys
from pathlib import Path

import numpy as np
import torch

from . import io as nio
from .config import Cfg, load_config, save_config
from .dataset import VideoDataset
from .harness import DiskRuntime
from .material import InvariantFullMetaElasticity, InvariantFullMetaPlasticity
from .prepare import prepare_simulation_data
from .sim import MPMInitData
from .train import finetune_constitutive, optimize_init_velocity

EPS = 6e-7          # finetune.py:47


def parse_args(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    p.add_argument("--config", "-c", type=str, required=True, help="Path to the config file.")
    return p.parse_args(argv)


def setup(cfg: Cfg, device, for_eval: bool = False):
    """Everything both drivers build before their loops: asset preparation, dataset, bindings, Gaussians, the nets with
    the base checkpoint, the particle set.  Returns a dict."""
    assets = Path(cfg.get("assets_root", "experiments/assets"))
    data_root = assets / cfg.sim_data_name
    data_root.mkdir(parents=True, exist_ok=True)
    pd = cfg.particle_data
    common = dict(save_dir=data_root, kernels_path=Path(cfg.gaussian.kernels_path), sh_degree=cfg.gaussian.sh_degree,
                  opacity_thres=cfg.gaussian.opacity_thres, confidence=cfg.gaussian.confidence, max_particles=cfg.gaussian.max_particles,
                  device=device)
    if pd.get("particles_path") is not None:
        prepare_simulation_data(particles_path=Path(pd.particles_path), particles_downsample_factor=pd.get("downsample_factor", 3), **common)
    elif pd.get("mesh_path") is not None:
        prepare_simulation_data(mesh_path=Path(pd.mesh_path), mesh_sample_mode=pd.get("mesh_sample_mode", "volumetric"),
                                mesh_sample_resolution=pd.get("mesh_sample_resolution", 30), particles_downsample_factor=1, **common)
    elif not for_eval:
        raise ValueError("Either 'particles_path' or 'mesh_path' must be provided in configuration.")
    cfg.video_data.device = str(device)                                      # finetune.py:582
    dataset = VideoDataset(cfg.video_data)
    bindings, n_particles = nio.load_bindings(data_root / "bindings.pt", device=device)
    print(f"Exp name [{cfg.name}]\nUsing data name [{cfg.sim_data_name}]")
    print(f"#Gaussians with particle bindings: {int((n_particles > 0).sum())}")
    print(f"#Avg particles: {float(n_particles.mean())}\n#Max particles: {float(n_particles.max())}")
    gaussians = nio.load_gaussians_ply(data_root / "kernels.ply", cfg.gaussian.sh_degree, device=device)
    elasticity = InvariantFullMetaElasticity(cfg.constitution.elasticity).to(device)
    plasticity = InvariantFullMetaPlasticity(cfg.constitution.plasticity).to(device)
    ckpt_path = cfg.get("change_base_model") or cfg.pretrained_ckpt
    pretrained = torch.load(ckpt_path, map_location=device)
    elasticity.load_state_dict(pretrained["elasticity"])
    plasticity.load_state_dict(pretrained["plasticity"])
    print(f"Loaded pretrained weights from {ckpt_path}")
    return dict(data_root=data_root, dataset=dataset, bindings=bindings, gaussians=gaussians, elasticity=elasticity,
                plasticity=plasticity)


def particle_init_data(cfg: Cfg, nsteps: int) -> MPMInitData:
    cfg.particle_data.span = [0, nsteps]                                     # NOTE: manually setting (finetune.py:109, 274)
    cfg.particle_data.shape.name = cfg.sim_data_name + "/particles"
    if cfg.particle_data.shape.get("asset_root") is None:
        cfg.particle_data.shape.asset_root = cfg.get("assets_root", "experiments/assets")
    return MPMInitData.get(cfg.particle_data)


def _views(dataset, spec):
    return sorted(dataset.views if spec in (None, "all") else spec)


def finetune(cfg: Cfg, log=print):
    seed = cfg.seed
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    device = torch.device(f"cuda:{cfg.gpu}")
    torch.cuda.set_device(device)
    force_mask_data = bool(cfg.video_data.data.get("read_mask_only", False))
    if force_mask_data:
        cfg.video_data.data.white_background = False
        print("[Warning] Force to use black background when loading mask data")
    background = torch.tensor([1.0, 1.0, 1.0] if cfg.video_data.data.get("white_background", False) else [0.0, 0.0, 0.0], device=device)
    exp_root = Path(cfg.root) / cfg.name
    if exp_root.exists() and not (cfg.get("resume") or cfg.get("overwrite")):
        raise FileExistsError(f"{exp_root} exists (set resume: true or overwrite: true)")
    if exp_root.exists() and cfg.get("overwrite") and not cfg.get("resume"):
        # nclaw/utils.py:42-47 mkdir(overwrite=True) removes the old experiment: a stale finetune/init.pt would make stage A
        # skip with the old velocity, stale NNNN_lora.pt files would win the keep-newest-3 rotation
        import shutil
        shutil.rmtree(exp_root, ignore_errors=True)
    exp_root.mkdir(parents=True, exist_ok=True)
    save_config(cfg, exp_root / "config.yaml")
    tune_root = exp_root / "finetune"
    tune_root.mkdir(exist_ok=True)
    env = setup(cfg, device)
    data_root, dataset = env["data_root"], env["dataset"]
    if (data_root / "init.pt").exists() and not (tune_root / "init.pt").exists():
        (tune_root / "init.pt").symlink_to((data_root / "init.pt").resolve())
        print(f"Found initial velocity from {data_root / 'init.pt'}.")
    scal = cfg.gaussian.get("scaling_modifier", 1.0)

    # ---- stage A: initial velocity (finetune.py:63-231)
    vc = cfg.velocity
    init_data = particle_init_data(cfg, vc.num_frames * vc.substeps)
    if (tune_root / "init.pt").exists():
        ix, iv = nio.load_init_state(tune_root / "init.pt")
        dataset.set_init_x_and_v(init_x=ix, init_v=iv)
        log("Loading initial velocity from checkpoint ...")
    else:
        dataset.set_init_x_and_v(init_x=init_data.pos)
        assert init_data.pos.shape[0] == dataset.get_init_x.shape[0]
        rt = DiskRuntime(cfg.sim, dataset, env["gaussians"], env["bindings"], init_data, env["elasticity"], env["plasticity"], background,
                         device, vc.substeps, _views(dataset, vc.get("views", "all")), scaling_modifier=scal,
                         pixel_loss=vc.get("pixel_loss", "l2"), fused=bool(cfg.get("fused_rollout", True)), eps=EPS)
        rt.force_mask_data = force_mask_data
        stage = dict(num_epochs=vc.num_epochs, num_frames=vc.num_frames, lr=vc.lr, scheduler=vc.scheduler, lambda_reg=vc.get("lambda_reg"),
                     reg_all=vc.get("reg_all", False), pixel_loss=vc.get("pixel_loss", "l2"), steps=dataset.steps)
        v_fit, _ = optimize_init_velocity(rt, rt.ground_truth(vc.num_frames), stage, tune_root=tune_root, log=log)
        dataset.set_init_x_and_v(init_x=rt.x0, init_v=rt.v0)
        del rt
    log(f"Initial velocity obtained: {dataset.get_init_v.mean(0).tolist()}.")

    # ---- stage B: constitutive adaptors (finetune.py:234-488)
    cc = cfg.constitution
    init_data = particle_init_data(cfg, cc.num_frames * cc.substeps)
    assert init_data.pos.shape[0] == dataset.get_init_x.shape[0], \
        f"Shape mismatch: init_data {init_data.pos.shape[0]} dataset {dataset.get_init_x.shape[0]}"
    E, P = env["elasticity"], env["plasticity"]
    E.init_lora_layers(r=cc.lora.r, lora_alpha=cc.lora.alpha)
    P.init_lora_layers(r=cc.lora.r, lora_alpha=cc.lora.alpha)
    rt = DiskRuntime(cfg.sim, dataset, env["gaussians"], env["bindings"], init_data, E, P, background, device, cc.substeps,
                     _views(dataset, cc.get("views", "all")), scaling_modifier=scal, pixel_loss=cc.get("pixel_loss", "l2"),
                     fused=bool(cfg.get("fused_rollout", True)), eps=EPS)
    rt.force_mask_data = force_mask_data
    stage = {k: cc[k] for k in ("elasticity_lr", "elasticity_wd", "elasticity_grad_max_norm", "elasticity_scheduler", "plasticity_lr",
                                "plasticity_wd", "plasticity_grad_max_norm", "plasticity_scheduler", "warmup_step", "decay_init",
                                "decay_final", "decay_steps", "lambda_max_decay", "num_epochs", "num_frames")}
    stage.update(exclude_steps=tuple(cc.get("exclude_steps", ())), num_lora_ckpts=cc.get("num_lora_ckpts", 3), resume=bool(cfg.get("resume")),
                 steps=dataset.steps)
    losses = finetune_constitutive(rt, rt.ground_truth(cc.num_frames), stage, tune_root=tune_root, log=log)
    log("Finetuning ends.")
    return losses


def main(argv=None):
    args = parse_args(argv)
    finetune(load_config(args.config))


if __name__ == "__main__":
    sys.exit(main())
