"""GPU parity: multi-object forward driver (neuma_amd/infer.py; inference.py:225-362)
against the fp64 oracle chain with per-section weights and span-based enabling."""
import math

import numpy as np
import pytest
import torch

from oracle import material as omat
from oracle import mpm as om
from oracle import raster as orr
from gpu_util import dev, rel_max, abs_max

pytestmark = pytest.mark.gpu


def _nets(material, device):
    from neuma_amd import synth
    from neuma_amd.harness import make_material_cfg
    from neuma_amd.material import InvariantFullMetaElasticity, InvariantFullMetaPlasticity
    w = synth.load_base_weights(material)
    E, P = InvariantFullMetaElasticity(make_material_cfg()).to(device), InvariantFullMetaPlasticity(make_material_cfg()).to(device)
    for net, tag in ((E, "e"), (P, "p")):
        net.layers[0].fc.weight.data.copy_(torch.tensor(w[tag][0]))
        net.layers[1].fc.weight.data.copy_(torch.tensor(w[tag][1]))
        net.final_layer.fc.weight.data.copy_(torch.tensor(w[tag][2]))
    return E.eval(), P.eval(), w


def _object(center, n, G, material, span, device, seed, K=300):
    from scipy.spatial import cKDTree
    from neuma_amd import synth
    from neuma_amd.infer import SceneObject
    from neuma_amd.render.gaussian_model import GaussianModel
    from neuma_amd.sim.mpm import MPMInitData
    from neuma_amd.tune import Bindings
    x = synth.ball_particles(n, G, (center,), seed).astype(np.float64)
    dx = 1.0 / G
    init = MPMInitData(rho=1000.0, clip_bound=0.1, span=span, num_particles=x.shape[0], vol=float((dx / 2) ** 3), pos=x,
                       lin_vel=np.array([0.0, -0.5, 0.0]), center=np.zeros(3), size=np.ones(3))
    E, P, w = _nets(material, device)
    rng = np.random.default_rng(seed + 7)
    pick = rng.integers(0, x.shape[0], size=K)
    gx = (x[pick] + rng.normal(0, 0.5 * dx, size=(K, 3))).astype(np.float32)
    gm = GaussianModel(0)
    q = rng.normal(size=(K, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    gm.set_params(torch.tensor(gx, device=device), torch.tensor(rng.normal(0, 0.3, size=(K, 1, 3)).astype(np.float32), device=device),
                  torch.zeros(K, 0, 3, device=device),
                  torch.tensor(rng.uniform(math.log(0.5 * dx), math.log(1.5 * dx), size=(K, 3)).astype(np.float32), device=device),
                  torch.tensor(q.astype(np.float32), device=device), torch.tensor(rng.normal(2, 1, size=(K, 1)).astype(np.float32), device=device))
    _, idx = cKDTree(x).query(gx, k=4)
    rows = torch.arange(K).repeat_interleave(4)
    b = Bindings(torch.stack([rows, torch.tensor(idx.reshape(-1))], 0), torch.full((K * 4,), 0.25), (K, x.shape[0]), device)
    return SceneObject(init, E, P, gm, b, 1.0), w, idx


def test_two_objects_with_spans_match_the_oracle_chain():
    from neuma_amd import synth
    from neuma_amd.infer import simulate_objects
    from neuma_amd.sim import MPMModelBuilder
    G, steps = 32, 5
    d = dev()
    model = MPMModelBuilder().parse_cfg(dict(gravity=[0.0, -9.8, 0.0], bc="noslip", num_grids=G, dt=1e-3, bound=1, eps=6e-7)).finalize(d)
    o1, w1, idx1 = _object((0.3, 0.5, 0.3), 700, G, "jelly", (0, 10 ** 9), d, seed=0)
    o2, w2, idx2 = _object((0.7, 0.5, 0.7), 500, G, "sand", (3, 10 ** 9), d, seed=1)     # joins the simulation at step 3
    cams = synth.ring_cameras(2, 96, 64, device=d)
    bg = torch.ones(3, device=d)
    k0 = [o1.gaussians.get_xyz.clone(), o2.gaussians.get_xyz.clone()]
    frames = list(simulate_objects(model, [o1, o2], steps, cams, bg, denormalize=True))
    assert [f["step"] for f in frames] == list(range(steps + 1)) and all(len(f["images"]) == 2 for f in frames)

    # ---- fp64 oracle chain with per-section weights and the reference's enable rule (mpm.py:67-72; update AFTER the step)
    n1, n2 = o1.init_data.num_particles, o2.init_data.num_particles
    const = om.MPMConstant(G, 1e-3, 1, (0.0, -9.8, 0.0), 6e-7, "noslip")
    x = torch.tensor(np.concatenate([o1.init_data.pos, o2.init_data.pos])).double()
    v = torch.tensor(np.tile([[0.0, -0.5, 0.0]], (n1 + n2, 1))).double()
    C = torch.zeros(n1 + n2, 3, 3, dtype=torch.float64)
    F = torch.eye(3, dtype=torch.float64).repeat(n1 + n2, 1, 1)
    vol = torch.full((n1 + n2,), o1.init_data.vol, dtype=torch.float64)
    rho = torch.full((n1 + n2,), 1000.0, dtype=torch.float64)
    clip = torch.full((n1 + n2,), 0.1, dtype=torch.float64)
    W = [[torch.tensor(a).double() for a in w["e"]] for w in (w1, w2)], [[torch.tensor(a).double() for a in w["p"]] for w in (w1, w2)]

    def enabled(step):
        e = torch.ones(n1 + n2, dtype=torch.int32)
        e[n1:] = 1 if step >= 3 else 0
        return e

    en = enabled(0)
    for step in range(1, steps + 1):
        stress = torch.cat([omat.elasticity(F[:n1], W[0][0]), omat.elasticity(F[n1:], W[0][1])])
        x, v, C, F = om.step(const, vol, rho, clip, en, x, v, C, F, stress)
        F = torch.cat([omat.plasticity(F[:n1], W[1][0], 1e-3), omat.plasticity(F[n1:], W[1][1], 1e-3)])
        en = enabled(step)
        fr = frames[step]
        assert abs_max(fr["x"], x) < 5e-7 and abs_max(fr["F"], F) < 7e-7, step      # measured 1.1e-07
    # object 2 did not move before it was enabled (its first enabled step is step 4: update(step=3) happens after step 3)
    assert abs_max(frames[3]["x"][n1:], torch.tensor(o2.init_data.pos)) == 0.0
    assert abs_max(frames[5]["x"][n1:], torch.tensor(o2.init_data.pos)) > 1e-5

    # ---- Gaussian positions: k_t = k_{t-1} + B (p_t - p_{t-1}) per object, concatenated
    p_prev = torch.tensor(np.concatenate([o1.init_data.pos, o2.init_data.pos])).double()
    k = torch.cat(k0).double().cpu()
    for step in range(1, steps + 1):
        p = frames[step]["x"].double().cpu()
        dk1 = (p[:n1] - p_prev[:n1])[torch.tensor(idx1)].mean(1)
        dk2 = (p[n1:] - p_prev[n1:])[torch.tensor(idx2)].mean(1)
        k = k + torch.cat([dk1, dk2])
        assert abs_max(frames[step]["means3D"], k) < 5e-7      # measured 1.0e-07
        p_prev = p

    # ---- images: frame 0 = un-deformed Gaussians (deform_grad None); last frame against the oracle rasterizer
    for fi in (0, steps):
        fr = frames[fi]
        cov = torch.cat([orr.build_cov3D(torch.exp(g._scaling.double().cpu()), g._rotation.double().cpu()) for g in (o1.gaussians, o2.gaussians)])
        if fi > 0:
            Fk = torch.cat([frames[fi]["F"].double().cpu()[:n1][torch.tensor(idx1)].mean(1), frames[fi]["F"].double().cpu()[n1:][torch.tensor(idx2)].mean(1)])
            cov = orr.deform_cov_by_F(cov, Fk)
        op = torch.cat([g.get_opacity.double().cpu() for g in (o1.gaussians, o2.gaussians)])
        sh = torch.cat([g.get_features.double().cpu() for g in (o1.gaussians, o2.gaussians)])
        cam = cams[0]
        s = orr.Settings(64, 96, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3, dtype=torch.float64), 1.0,
                         cam.world_view_transform.double().cpu(), cam.full_proj_transform.double().cpu(), 0, cam.camera_center.double().cpu())
        ref = orr.render(s, fr["means3D"].double().cpu(), cov, op, shs=sh)[0]
        assert abs_max(fr["images"][0], ref) < 2e-6      # measured 6.3e-07


def test_scene_round_trips_through_the_on_disk_formats(tmp_path):
    """kernels.ply / particles.ply / bindings.pt / init.pt / data_dynamic.json written in the reference's layouts, read back
    by neuma_amd.io and rendered: same image as the in-memory scene."""
    import json
    from PIL import Image
    from neuma_amd import io as nio, synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F, diff_rasterization
    d = dev()
    scene = synth.make_scene("tiny")
    rt = SceneRuntime(scene, d)
    root = tmp_path / "data"
    (root / "data_dynamic").mkdir(parents=True)
    nio.save_gaussians_ply(rt.gaussians, root / "kernels.ply")
    nio.save_particles_ply(root / "particles.ply", scene.x0)
    # (the runtime keeps its kernels in its own - spatial - order, rt.gaussian_perm; the files are written in that order too)
    order = np.arange(scene.bind_idx.shape[0]) if rt.gaussian_perm is None else rt.gaussian_perm.numpy()
    bind_idx, bind_w = scene.bind_idx[order], scene.bind_w[order]
    K, nb = bind_idx.shape
    ind = torch.stack([torch.arange(K).repeat_interleave(nb), torch.tensor(bind_idx.reshape(-1))], 0)
    nio.save_bindings(root / "bindings.pt", ind, torch.tensor(bind_w.reshape(-1)), (K, rt.N), torch.full((K,), nb))
    torch.save({"init_x": torch.tensor(scene.x0), "init_v": torch.tensor(scene.v0)}, root / "init.pt")
    entries = []
    W, H = scene.cfg["W"], scene.cfg["H"]
    for vi, cam in enumerate(rt.cameras):
        w2c = cam.world_view_transform.double().cpu().numpy().T
        c2w = np.linalg.inv(w2c)
        c2w[:3, 1:3] *= -1                      # COLMAP -> OpenGL axes (the reader flips them back)
        fx, fy = nio.fov2focal(cam.FoVx, W), nio.fov2focal(cam.FoVy, H)
        name = f"./data_dynamic/r_{vi}_000.png"
        Image.fromarray(np.full((H, W, 4), 255, dtype=np.uint8), "RGBA").save(root / name)
        entries.append({"file_path": name, "c2w": c2w[:3].tolist(), "intrinsic": [[fx, 0, W / 2], [0, fy, H / 2], [0, 0, 1]]})
    (root / "data_dynamic.json").write_text(json.dumps(entries))

    g2 = nio.load_gaussians_ply(root / "kernels.ply", scene.cfg["sh"], device=d)
    b2, n_p = nio.load_bindings(root / "bindings.pt", device=d)
    x_disk = torch.tensor(nio.load_particles_ply(root / "particles.ply"), dtype=torch.float32, device=d)
    init_x, init_v = nio.load_init_state(root / "init.pt")
    cams = nio.read_neuma_synthetic_cameras(str(root), "data_dynamic.json", True, load_images=True)
    assert float(n_p.min()) == nb and abs_max(x_disk, rt.x0) == 0.0 and abs_max(init_v, rt.v0) == 0.0
    assert cams["views"] == [f"r_{i}" for i in range(rt.V)] and cams["cam_infos"][0].image.shape == (H, W, 3)
    with torch.no_grad():
        x, v, C, F = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
        for vi in range(rt.V):
            ref = rt.render_view(compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings), compute_bindings_F(F, rt.bindings), vi)
            cam = nio.DiskCamera(cams["cam_infos"][vi], device=d)
            m3 = compute_bindings_xyz(x, x_disk, g2.get_xyz, b2)
            img = diff_rasterization(m3, compute_bindings_F(F, b2), g2, cam, rt.background)
            assert abs_max(img, ref) < 1e-4      # measured 3.1e-05
