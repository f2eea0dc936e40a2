"""GPU: the config-driven entry points (`python -m neuma_amd.finetune`, `python -m neuma_amd.render`) on a tiny experiment
written to disk in the reference's layout - YAML in the schema of experiments/configs/synthetic/finetune-bb.yaml, a
NeuMASynthetic dataset (data_dynamic.json + RGBA frames), a 3DGS point_cloud.ply, a particle cloud, a base checkpoint - with
asset preparation (opacity pruning, binding construction), stage A, stage B, checkpoint rotation, resume and the forward
renderer.  Counterpart of experiments/finetune.py:491-663 and render.py:109-347."""
import json

import numpy as np
import pytest
import torch
import yaml

from gpu_util import dev

pytestmark = pytest.mark.gpu


def _write_experiment(tmp_path, frames=3):
    from PIL import Image
    from neuma_amd import io as nio, synth
    from neuma_amd.config import load_config
    from neuma_amd.dataset import VideoDataset
    from neuma_amd.finetune import particle_init_data, setup
    from neuma_amd.harness import DiskRuntime
    from neuma_amd.render.gaussian_model import GaussianModel
    from neuma_amd.train import simulate_video
    d = dev()
    scene = synth.make_scene("tiny")
    W, H = scene.cfg["W"], scene.cfg["H"]
    raw = tmp_path / "raw"
    raw.mkdir()
    # reconstructed kernels (with a few nearly transparent ones that preparation must prune) and the particle cloud
    gm = GaussianModel(scene.cfg["sh"])
    sh = torch.tensor(scene.g_sh)
    op = torch.tensor(scene.g_opacity_logit).clone()
    op[:50] = -8.0                                              # sigmoid(-8) < opacity_thres
    gm.set_params(torch.tensor(scene.g_xyz), sh[:, :1].contiguous(), sh[:, 1:].contiguous(), torch.tensor(scene.g_logscale),
                  torch.tensor(scene.g_rot), op)
    nio.save_gaussians_ply(gm, raw / "point_cloud.ply")
    nio.save_particles_ply(raw / "pcd.ply", scene.x0)
    w = synth.load_base_weights("jelly")
    keys = ("layers.0.fc.weight", "layers.1.fc.weight", "final_layer.fc.weight")
    torch.save({t: {k: torch.tensor(a) for k, a in zip(keys, w[s])} for t, s in (("elasticity", "e"), ("plasticity", "p"))},
               raw / "jelly_0300.pt")
    data = tmp_path / "dataset"
    (data / "data_dynamic").mkdir(parents=True)
    cams = synth.ring_cameras(2, W, H)
    entries = []
    for vi, cam in enumerate(cams):
        c2w = np.linalg.inv(cam.world_view_transform.double().numpy().T)
        c2w[:3, 1:3] *= -1
        fx, fy = nio.fov2focal(cam.FoVx, W), nio.fov2focal(cam.FoVy, H)
        for step in range(frames + 1):
            entries.append({"file_path": f"./data_dynamic/r_{vi}_{step:03d}.png", "c2w": c2w[:3].tolist(),
                            "intrinsic": [[fx, 0, W / 2], [0, fy, H / 2], [0, 0, 1]]})
            Image.fromarray(np.full((H, W, 4), 255, dtype=np.uint8), "RGBA").save(data / entries[-1]["file_path"])
    (data / "data_dynamic.json").write_text(json.dumps(entries))
    net = dict(layer_widths=[64, 64], norm=None, nonlinearity="gelu", no_bias=True, normalize_input=True)
    sched = dict(type="cos", max_steps=4, learning_rate_alpha=0.1)
    cfg = dict(gpu=0, seed=42, debug=False, debug_views=["r_0"], resume=False, overwrite=False, root=str(tmp_path / "logs"),
               assets_root=str(tmp_path / "assets"), sim_data_name="tinyball", name="tinyball-v1", pretrained_ckpt=str(raw / "jelly_0300.pt"),
               gaussian=dict(sh_degree=3, opacity_thres=0.02, confidence=0.95, max_particles=10, kernels_path=str(raw / "point_cloud.ply")),
               video_data=dict(eval=False, camera_type="NeuMASynthetic",
                               data=dict(path=str(data), transformsfile="data_dynamic.json", white_background=True, exclude_steps=[-1],
                                         used_views=["r_0", "r_1"]), camera=dict(resolution=1, data_device="cpu")),
               sim=dict(gravity=[0.0, -9.8, 0.0], bc="noslip", num_grids=32, dt=1e-3, bound=1, eps=0.0, skip_frame=1),
               particle_data=dict(shape=dict(asset_root=None, sort=None, ori_bounds=[[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]],
                                             sim_bounds=[[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]]), rho=1000.0, clip_bound=0.1,
                                  particles_path=str(raw / "pcd.ply"), downsample_factor=1),
               constitution=dict(elasticity=dict(net), plasticity=dict(net, alpha=1e-3), elasticity_lr=0.02, elasticity_wd=0.0,
                                 elasticity_grad_max_norm=1.0, elasticity_scheduler=sched, plasticity_lr=0.002, plasticity_wd=0.0,
                                 plasticity_grad_max_norm=1.0, plasticity_scheduler=sched, warmup_step=0, decay_init=0.5, decay_final=1.0,
                                 decay_steps=2, lambda_max_decay=0.33, lora=dict(r=16, alpha=16), num_epochs=2, substeps=4,
                                 num_frames=frames, views=["r_0", "r_1"], num_lora_ckpts=3),
               velocity=dict(num_epochs=3, num_frames=2, substeps=4, lambda_reg=0.005, views=["r_0"], lr=0.2, scheduler=sched))
    path = tmp_path / "finetune-tiny.yaml"
    path.write_text(yaml.safe_dump(cfg, sort_keys=False))
    # ---- ground-truth video: prepare the assets exactly as the driver will, then roll a "true" material out and overwrite
    #      the placeholder frames
    c = load_config(path)
    env = setup(c, d)
    init_data = particle_init_data(c, frames * 4)
    ds = env["dataset"]
    ds.set_init_x_and_v(init_x=init_data.pos, init_v=np.tile(np.array([[0.3, -0.6, 0.1]], np.float32), (init_data.pos.shape[0], 1)))
    E, P = env["elasticity"], env["plasticity"]
    for n in (E, P):
        n.init_lora_layers(16, 16)
        for lin in (n.layers[0].fc, n.layers[1].fc, n.final_layer.fc):
            lin.lora_B.data.normal_(0, 0.05)
    rt = DiskRuntime(c.sim, ds, env["gaussians"], env["bindings"], init_data, E.to(d), P.to(d), torch.ones(3, device=d), d, 4, ["r_0", "r_1"])
    video = simulate_video(rt, frames)
    first = [rt.render_view(rt.gaussians.get_xyz, None, vi, cov=rt._cov) for vi in range(2)]
    for step, imgs in enumerate([first] + video):
        for vi, img in enumerate(imgs):
            a = (img.clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
            Image.fromarray(np.concatenate([a, np.full((H, W, 1), 255, np.uint8)], -1), "RGBA").save(data / f"data_dynamic/r_{vi}_{step:03d}.png")
    return path, scene


def test_finetune_and_render_entry_points(tmp_path):
    from neuma_amd import io as nio
    from neuma_amd.config import load_config
    from neuma_amd.evaluate import main as render_main
    from neuma_amd.finetune import finetune, main as finetune_main
    path, scene = _write_experiment(tmp_path)
    assets = tmp_path / "assets" / "tinyball"
    # asset preparation happened through the reference's rules
    K = nio.load_gaussians_ply(assets / "kernels.ply", 3).get_xyz.shape[0]
    assert K == scene.g_xyz.shape[0] - 50                                   # 50 transparent kernels pruned
    B, n_p = nio.load_bindings(assets / "bindings.pt")
    N = nio.load_particles_ply(assets / "particles.ply").shape[0]
    assert N >= scene.x0.shape[0] and (B.K, B.N) == (K, N) and float(n_p.min()) >= 1 and float(n_p.max()) <= 10
    assert (assets / "particles.npz").exists()                              # MPMInitData cache written like the reference's
    # ---- the whole driver through its command line
    finetune_main(["-c", str(path)])
    exp = tmp_path / "logs" / "tinyball-v1"
    tune = exp / "finetune"
    saved = yaml.safe_load((exp / "config.yaml").read_text())
    assert saved["name"] == "tinyball-v1" and saved["constitution"]["lora"] == {"r": 16, "alpha": 16}      # finetune.py:529
    init = torch.load(tune / "init.pt")
    assert set(init) == {"init_x", "init_v"} and init["init_v"].shape == (N, 3)
    v_fit = init["init_v"][0]
    assert torch.isfinite(v_fit).all() and float(v_fit.abs().max()) > 0 and float((init["init_v"] - v_fit).abs().max()) == 0
    assert sorted(p.name for p in tune.glob("*_lora.pt")) == ["0001_lora.pt", "0002_lora.pt"]
    ck = torch.load(tune / "0002_lora.pt")
    assert set(ck) == {"elasticity", "plasticity", "loss"} and np.isfinite(ck["loss"])
    assert sorted(ck["elasticity"]) == sorted(f"{p}.fc.lora_{ab}" for p in ("layers.0", "layers.1", "final_layer") for ab in "AB")
    # ---- an existing experiment is refused unless resume / overwrite; resume reuses init.pt and the newest adaptor
    with pytest.raises(FileExistsError):
        finetune(load_config(path))
    cfg = load_config(path)
    cfg.resume = True
    cfg.constitution.num_epochs = 1
    logs = []
    losses = finetune(cfg, log=logs.append)
    assert any("Loading initial velocity from checkpoint" in l for l in logs)
    assert len(losses) == 1 and abs(losses[0] - ck["loss"]) < 0.5 * ck["loss"] + 1e-9
    # ---- forward renderer with the fine-tuned adaptor
    render_main(["-c", str(path), "-vn", "check", "-es", "30", "-dt", "0.004", "-l", "0002_lora.pt", "-dv", "r_0", "-sp", "run",
                 "--result_root", str(tmp_path / "results")])
    out = tmp_path / "results" / "tinyball-v1"
    imgs = sorted(p.name for p in (out / "images_check").glob("*.png"))
    assert imgs == [f"r_0_{i:03d}.png" for i in range(31)]
    assert sorted(p.name for p in (out / "states_run").glob("*.ply")) == [f"{i:03d}.ply" for i in range(1, 31)]
    from PIL import Image
    a0 = np.array(Image.open(out / "images_check" / "r_0_000.png")).astype(np.float64)
    gt0 = np.array(Image.open(tmp_path / "dataset" / "data_dynamic" / "r_0_000.png").convert("RGB")).astype(np.float64)
    assert np.abs(a0 - gt0).max() <= 1.0                                      # first frame = un-deformed kernels = the GT's first frame
    a30 = np.array(Image.open(out / "images_check" / "r_0_030.png")).astype(np.float64)
    assert np.abs(a30 - a0).max() > 5                                         # the body fell (0.12 s of gravity)
    x30 = nio.load_particles_ply(out / "states_run" / "030.ply")
    assert x30.shape == (N, 3) and np.isfinite(x30).all() and x30[:, 1].mean() < nio.load_particles_ply(assets / "particles.ply")[:, 1].mean() - 0.03


def test_inference_entry_point_two_objects(tmp_path):
    """`python -m neuma_amd.inference -c demo.yaml` (experiments/inference.py:48-84, 87-377, 380-386): a two-object scene in the
    schema of experiments/configs/demo/multiobj-bb-cc.yaml - per-object asset folders, base checkpoints (jelly / plasticine),
    a LoRA adaptor for one of them, initial velocities, boxes one above the other, Gaussians mapped into the simulation box -
    rolled out and rendered from the command line; the frames and particle states on disk are those of the operator-level
    driver (infer.simulate_objects) called by hand on the same objects."""
    from PIL import Image
    from neuma_amd import io as nio, synth
    from neuma_amd.config import load_config
    from neuma_amd.inference import inference, load_object, main as inference_main
    from neuma_amd.infer import simulate_objects
    from neuma_amd.material import InvariantFullMetaElasticity
    from neuma_amd.sim import MPMModelBuilder
    path, scene = _write_experiment(tmp_path, frames=1)
    base = yaml.safe_load(path.read_text())
    raw, assets = tmp_path / "raw", tmp_path / "assets"
    w = synth.load_base_weights("plasticine")
    keys = ("layers.0.fc.weight", "layers.1.fc.weight", "final_layer.fc.weight")
    torch.save({t: {k: torch.tensor(a) for k, a in zip(keys, w[s])} for t, s in (("elasticity", "e"), ("plasticity", "p"))}, raw / "plasticine_0300.pt")
    # a LoRA adaptor file in the layout finetune writes (NNNN_lora.pt: the lora_A / lora_B entries of both nets)
    lora = {}
    for tag in ("elasticity", "plasticity"):
        net = InvariantFullMetaElasticity(base["constitution"]["elasticity"])
        net.init_lora_layers(r=16, lora_alpha=4)
        g = torch.Generator().manual_seed(5)
        lora[tag] = {k: (0.05 * torch.randn(v.shape, generator=g)) for k, v in net.state_dict().items() if "lora_" in k}
    torch.save(dict(lora, loss=0.0), raw / "0100_lora.pt")
    # the second object uses the same prepared asset folder under its own name (a copy: kernels.ply, particles.ply, bindings.pt)
    import shutil
    shutil.copytree(assets / "tinyball", assets / "tinycat")
    for f in (assets / "tinycat").glob("particles.npz"):
        f.unlink()              # the MPMInitData cache is rebuilt from particles.ply, as for a fresh asset folder

    def obj(name, ckpt, lo, vel, lora_path=None):
        c = dict(sim_data_name=name, pretrained_ckpt=str(raw / ckpt), gaussian=dict(sh_degree=3),
                 particle_data=dict(shape=dict(asset_root=None, sort=None, ori_bounds=[[0.0, 0.0, 0.0], [1.0, 1.0, 1.0]],
                                               sim_bounds=[[0.25, lo, 0.25], [0.75, lo + 0.5, 0.75]]),
                                    vel=dict(lin_vel=vel, ang_vel=[0.0, 0.0, 0.0]), rho=1000.0, clip_bound=0.1),
                 constitution=dict(elasticity=base["constitution"]["elasticity"], plasticity=base["constitution"]["plasticity"], views=["r_0"]))
        if lora_path:
            c["constitution"].update(load_lora=str(lora_path), lora=dict(r=16, alpha=4))
        return c

    cfg = dict(gpu=0, seed=42, debug=True, debug_views=["r_1"], resume=False, overwrite=False, denormalize=False, assets_root=str(assets),
               video_data=dict(base["video_data"], data=dict(base["video_data"]["data"], init_frame=0, used_views=["r_1"])),
               sim=dict(base["sim"], num_grids=32, eps=6e-7),
               objects=[obj("tinyball", "jelly_0300.pt", 0.45, [0.0, -0.5, 0.0], raw / "0100_lora.pt"),
                        obj("tinycat", "plasticine_0300.pt", 0.02, [0.0, -0.5, 0.0])])
    demo = tmp_path / "multiobj-tiny.yaml"
    demo.write_text(yaml.safe_dump(cfg, sort_keys=False))
    steps = 12
    inference_main(["-c", str(demo), "-s", str(steps), "-vn", "pair", "-dv", "r_0", "-sp", "pair", "--result_root", str(tmp_path / "results")])
    img_root = tmp_path / "results" / "inference" / "images_pair"
    assert sorted(p.name for p in img_root.glob("*.png")) == [f"r_0_{i:03d}.png" for i in range(steps + 1)]      # -dv overrides the YAML's r_1
    st_root = tmp_path / "results" / "inference_states" / "states_pair"
    assert sorted(p.name for p in st_root.glob("*.ply")) == [f"{i:03d}.ply" for i in range(1, steps + 1)]
    assert (tmp_path / "results" / "inference_videos").is_dir()
    # ---- the same scene through the operators, by hand
    c = load_config(demo)
    d = dev()
    objects = [load_object(o, assets, steps, d) for o in c.objects]
    n0, n1 = (o.init_data.num_particles for o in objects)
    assert objects[0].elasticity.layers[0].fc.r == 16 and not getattr(objects[1].elasticity.layers[0].fc, "r", 0)      # adaptor on the first object only
    assert abs(objects[0].init_data.pos[:, 1].mean() - objects[1].init_data.pos[:, 1].mean() - 0.43) < 0.02         # one box above the other
    model = MPMModelBuilder().parse_cfg(c.sim).finalize(d, False)
    from neuma_amd.dataset import CameraDataset
    c.video_data.data.used_views = ["r_0"]
    ds = CameraDataset(c.video_data)
    cam = ds.getCameras("r_0", ds.steps[0])
    frames = list(simulate_objects(model, objects, steps, [cam], torch.ones(3, device=d)))
    x_last = nio.load_particles_ply(st_root / f"{steps:03d}.ply")
    assert x_last.shape == (n0 + n1, 3)
    # two runs of one scene differ by the order of the scatters' float atomics: positions to 1e-6, 8-bit frames to one level
    from gpu_util import measured
    assert measured(np.abs(x_last - frames[-1]["x"].cpu().numpy()).max(), "x after 12 steps: command line vs operators (abs)") < 2e-6
    for i in (0, steps):
        a = np.array(Image.open(img_root / f"r_0_{i:03d}.png")).astype(np.int32)
        b = (frames[i]["images"][0].clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).cpu().numpy().astype(np.int32)
        assert measured(np.abs(a - b).max(), f"frame {i}: 8-bit levels, command line vs operators") <= 1
    a0 = np.array(Image.open(img_root / "r_0_000.png")).astype(np.int32)
    aN = np.array(Image.open(img_root / f"r_0_{steps:03d}.png")).astype(np.int32)
    assert (a0 < 250).any(axis=2).sum() > 50 and np.abs(aN - a0).max() > 5          # both bodies visible on white, and they moved
    # both bodies fall; the first section is the upper one
    x0 = frames[0]["x"].cpu().numpy()
    assert (x_last[:n0, 1].mean() < x0[:n0, 1].mean() - 5e-3) and (x_last[n0:, 1].mean() < x0[n0:, 1].mean() - 1e-3)
    # an object without an initial velocity is refused with the reason
    bad = yaml.safe_load(demo.read_text())
    del bad["objects"][1]["particle_data"]["vel"]
    (tmp_path / "bad.yaml").write_text(yaml.safe_dump(bad))
    with pytest.raises(ValueError, match="particle_data.vel"):
        inference_main(["-c", str(tmp_path / "bad.yaml"), "-s", "2", "-vn", "bad", "--result_root", str(tmp_path / "results")])
