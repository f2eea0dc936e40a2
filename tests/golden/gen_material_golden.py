#!/usr/bin/env python3
"""Generate golden vectors for the constitutive nets by IMPORTING the reference's own classes.

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/gen_material_golden.py

`warp` and `omegaconf` are absent, so identity-decorator stubs are injected; the reference's
`SVD` module (a Warp kernel) is replaced by torch.linalg.svd + the det/sign rule of
modules/nclaw/warp/svd.py:76-92.  Everything else (MLPBlock, LinearLoRA, invariants, composition)
is the reference's code, executed in fp64 on CPU.

Outputs (data only — inputs and expected outputs):
    tests/golden/base_models.npz              the three shipped checkpoints as plain arrays (+ the same file as package
                                              data neuma_amd/data/base_models.npz, which synth / bench load)
    tests/golden/material_<name>.npz          F, LoRA A/B, stress, F_p, and autograd gradients
    tests/golden/camera_sh_golden.npz         view/proj matrices, SH evaluations, l1/l2 loss values
    tests/golden/scheduler_golden.npz         learning-rate curves of the reference's cosine / exponential schedulers
"""
import os
import sys
import types
import math
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
OUT = Path(os.environ.get("NEUMA_GOLDEN_OUT", HERE))      # (tests/test_golden_regen.py regenerates into a temporary directory)


def install_stubs():
    wp = types.ModuleType("warp")

    def ident(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    wp.kernel = ident
    wp.struct = ident
    wp.func = ident

    class _T:  # placeholder types
        def __init__(self, *a, **k):
            pass

    for n in ["vec3", "mat33", "float32", "int8", "Tape"]:
        setattr(wp, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
    wp.array = lambda *a, **k: None
    wp.context = types.SimpleNamespace(Devicelike=object)
    wp.codegen = types.SimpleNamespace(StructInstance=_T)
    wp.types = types.SimpleNamespace(array=_T)
    sys.modules["warp"] = wp

    oc = types.ModuleType("omegaconf")

    class DictConfig(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    oc.DictConfig = DictConfig
    oc.OmegaConf = object
    sys.modules["omegaconf"] = oc
    return DictConfig


class TorchSVD(torch.nn.Module):
    """Stand-in for modules/nclaw/warp/svd.py SVD: same outputs/sign rule, torch autograd adjoint."""

    def forward(self, F):
        U, s, Vh = torch.linalg.svd(F)
        fu = torch.where(torch.linalg.det(U) < 0, -1.0, 1.0).to(F)
        fv = torch.where(torch.linalg.det(Vh) < 0, -1.0, 1.0).to(F)
        one = torch.ones_like(fu)
        U = U * torch.stack([one, one, fu], -1)[:, None, :]
        Vh = Vh * torch.stack([one, one, fv], -1)[:, :, None]
        s = s * torch.stack([one, one, fu * fv], -1)
        return U, s, Vh


def make_F(n, seed, spread):
    g = torch.Generator().manual_seed(seed)
    F = torch.eye(3, dtype=torch.float64)[None] + spread * torch.randn(n, 3, 3, generator=g, dtype=torch.float64)
    # a few reflected matrices (negative sigma_2 branch of svd.py:76-92)
    F[-4:, :, 2] *= -1.0
    return F


def main():
    DictConfig = install_stubs()
    sys.path.insert(0, str(REF))
    import modules.nclaw.material as material  # noqa: the reference package

    cfg = DictConfig(layer_widths=[64, 64], norm=None, nonlinearity="gelu", no_bias=True,
                     normalize_input=True, alpha=1e-3)
    base = {}
    for name in ["jelly", "plasticine", "sand"]:
        ckpt = torch.load(REF / "experiments" / "base_models" / f"{name}_0300.pt", map_location="cpu")
        E = material.InvariantFullMetaElasticity(cfg).double()
        P = material.InvariantFullMetaPlasticity(cfg).double()
        print(name, E.load_state_dict({k: v.double() for k, v in ckpt["elasticity"].items()}),
              P.load_state_dict({k: v.double() for k, v in ckpt["plasticity"].items()}))
        E.svd = TorchSVD()
        P.svd = TorchSVD()
        for net, tag in [(ckpt["elasticity"], "e"), (ckpt["plasticity"], "p")]:
            base[f"{name}_{tag}_w0"] = net["layers.0.fc.weight"].numpy().astype(np.float32)
            base[f"{name}_{tag}_w1"] = net["layers.1.fc.weight"].numpy().astype(np.float32)
            base[f"{name}_{tag}_w2"] = net["final_layer.fc.weight"].numpy().astype(np.float32)

        F = make_F(68, seed=7, spread=0.08)
        out = {"F": F.numpy()}
        # --- plain (no LoRA) forward
        with torch.no_grad():
            out["stress_plain"] = E(F).numpy()
            out["Fp_plain"] = P(F).numpy()
        # --- LoRA r=16 alpha=16 (finetune-bb.yaml:100-102), un-merged path as finetune.py runs it
        torch.manual_seed(0)
        E.init_lora_layers(r=16, lora_alpha=16)
        P.init_lora_layers(r=16, lora_alpha=16)
        E.double(); P.double()
        g = torch.Generator().manual_seed(1)
        for net, tag in [(E, "e"), (P, "p")]:
            for li, lin in enumerate([net.layers[0].fc, net.layers[1].fc, net.final_layer.fc]):
                lin.lora_B.data = 0.01 * torch.randn(lin.lora_B.shape, generator=g, dtype=torch.float64)
                out[f"{tag}_A{li}"] = lin.lora_A.detach().numpy().copy()
                out[f"{tag}_B{li}"] = lin.lora_B.detach().numpy().copy()
                out[f"{tag}_scaling"] = np.float64(lin.scaling)
        E.freeze_all_except_lora(); P.freeze_all_except_lora()
        E.train(); P.train()
        Fg = F.clone().requires_grad_(True)
        stress = E(Fg)
        gs = torch.randn(stress.shape, generator=g, dtype=torch.float64)
        (stress * gs).sum().backward()
        out["stress_lora"] = stress.detach().numpy()
        out["gout_e"] = gs.numpy()
        out["gF_e"] = Fg.grad.numpy().copy()
        for li, lin in enumerate([E.layers[0].fc, E.layers[1].fc, E.final_layer.fc]):
            out[f"e_gA{li}"] = lin.lora_A.grad.numpy().copy()
            out[f"e_gB{li}"] = lin.lora_B.grad.numpy().copy()
        Fg = F.clone().requires_grad_(True)
        Fp = P(Fg)
        gp = torch.randn(Fp.shape, generator=g, dtype=torch.float64)
        (Fp * gp).sum().backward()
        out["Fp_lora"] = Fp.detach().numpy()
        out["gout_p"] = gp.numpy()
        out["gF_p"] = Fg.grad.numpy().copy()
        for li, lin in enumerate([P.layers[0].fc, P.layers[1].fc, P.final_layer.fc]):
            out[f"p_gA{li}"] = lin.lora_A.grad.numpy().copy()
            out[f"p_gB{li}"] = lin.lora_B.grad.numpy().copy()
        # --- merged (eval) path, loralib.py:199-214
        E.eval(); P.eval()
        with torch.no_grad():
            out["stress_lora_merged"] = E(F).numpy()
            out["Fp_lora_merged"] = P(F).numpy()
        np.savez_compressed(OUT / f"material_{name}.npz", **out)
    np.savez_compressed(OUT / "base_models.npz", **base)
    np.savez_compressed((HERE.parent.parent / "neuma_amd" / "data" if OUT == HERE else OUT / "data") / "base_models.npz", **base)

    # ---- camera / SH / loss conventions (pure torch/numpy reference modules, loaded by file path)
    import importlib.util

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    gu = load(REF / "modules/d3gs/utils/graphics_utils.py", "ref_graphics_utils")
    sh = load(REF / "modules/d3gs/utils/sh_utils.py", "ref_sh_utils")
    lu = load(REF / "modules/d3gs/utils/loss_utils.py", "ref_loss_utils")
    rng = np.random.default_rng(3)
    cam = {}
    # a rotation + translation in the reference's (R, T) convention: cameras.py:54-57
    ang = 0.7
    R = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    T = np.array([0.1, -0.2, 2.0])
    wv = torch.tensor(gu.getWorld2View2(R, T, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
    fovx, fovy = 0.8, 0.6
    proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    cam.update(R=R, T=T, fovx=fovx, fovy=fovy, world_view=wv.numpy(), proj=proj.numpy(), full_proj=full.numpy(),
               center=wv.inverse()[3, :3].numpy())
    dirs = rng.normal(size=(32, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    coef = rng.normal(size=(32, 3, 16))
    for deg in range(4):
        cam[f"sh_deg{deg}"] = sh.eval_sh(deg, torch.tensor(coef), torch.tensor(dirs)).numpy()
    cam.update(sh_dirs=dirs, sh_coef=coef)
    a = torch.tensor(rng.random((3, 8, 9))); b = torch.tensor(rng.random((3, 8, 9)))
    cam.update(loss_a=a.numpy(), loss_b=b.numpy(), l1=lu.l1_loss(a, b).numpy(), l2=lu.l2_loss(a, b).numpy())
    np.savez_compressed(OUT / "camera_sh_golden.npz", **cam)

    # ---- LR schedulers (modules/tune/scheduler/__init__.py:29-118) evaluated through torch LambdaLR
    sch = load(REF / "modules/tune/scheduler/__init__.py", "ref_scheduler")
    out = {}
    for tag, cfgd, lr0 in [("cos", dict(type="cos", max_steps=1000, learning_rate_alpha=0.025), 0.008),
                           ("cos_warm", dict(type="cos", max_steps=200, learning_rate_alpha=0.01, warm_up_end=20), 1.0),
                           ("exp", dict(type="exp", lr_final=1e-4, max_steps=500, warmup_steps=10), 0.01)]:
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=lr0)
        sc = sch.fetch_scheduler(DictConfig(cfgd)).get_scheduler(opt, lr0)
        lrs = []
        for _ in range(int(cfgd["max_steps"]) + 5):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step(); sc.step()
        out[tag] = np.array(lrs)
    np.savez_compressed(OUT / "scheduler_golden.npz", **out)
    print("done")


if __name__ == "__main__":
    main()
