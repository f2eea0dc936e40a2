"""Oracle vs golden vectors produced by the reference's own material classes
(tests/golden/gen_material_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import material as om

NAMES = ["jelly", "plasticine", "sand"]


def _load(golden_dir, name):
    g = np.load(golden_dir / f"material_{name}.npz")
    b = np.load(golden_dir / "base_models.npz")
    W = {t: [torch.tensor(b[f"{name}_{t}_w{i}"], dtype=torch.float64) for i in range(3)] for t in "ep"}
    return g, W


@pytest.mark.parametrize("name", NAMES)
def test_plain_forward_matches_reference(golden_dir, name):
    g, W = _load(golden_dir, name)
    F = torch.tensor(g["F"])
    s = om.elasticity(F, W["e"])
    fp = om.plasticity(F, W["p"], 1e-3)
    scale = np.abs(g["stress_plain"]).max()
    assert np.abs(s.numpy() - g["stress_plain"]).max() <= 1e-10 * scale
    assert np.abs(fp.numpy() - g["Fp_plain"]).max() <= 1e-12


@pytest.mark.parametrize("name", NAMES)
def test_lora_forward_backward_matches_reference(golden_dir, name):
    g, W = _load(golden_dir, name)
    F = torch.tensor(g["F"], requires_grad=True)
    for tag, fn, key, gout in [("e", lambda F_, lo, sc: om.elasticity(F_, W["e"], lo, sc), "stress_lora", "gout_e"),
                               ("p", lambda F_, lo, sc: om.plasticity(F_, W["p"], 1e-3, lo, sc), "Fp_lora", "gout_p")]:
        lora = [(torch.tensor(g[f"{tag}_A{i}"], requires_grad=True), torch.tensor(g[f"{tag}_B{i}"], requires_grad=True))
                for i in range(3)]
        sc = float(g[f"{tag}_scaling"])
        Fg = torch.tensor(g["F"], requires_grad=True)
        out = fn(Fg, lora, sc)
        scale = np.abs(g[key]).max()
        assert np.abs(out.detach().numpy() - g[key]).max() <= 1e-10 * scale
        (out * torch.tensor(g[gout])).sum().backward()
        ref = g[f"gF_{tag}"]
        assert np.abs(Fg.grad.numpy() - ref).max() <= 1e-8 * np.abs(ref).max()
        for i in range(3):
            for nm, t in [("A", lora[i][0]), ("B", lora[i][1])]:
                ref = g[f"{tag}_g{nm}{i}"]
                assert np.abs(t.grad.numpy() - ref).max() <= 1e-8 * max(np.abs(ref).max(), 1e-30)
        # merged path == un-merged path (loralib.py:199-214)
        Wm = [om.lora_effective_weight(W[tag][i], lora[i][0].detach(), lora[i][1].detach(), sc) for i in range(3)]
        merged = om.elasticity(torch.tensor(g["F"]), Wm) if tag == "e" else om.plasticity(torch.tensor(g["F"]), Wm, 1e-3)
        mk = "stress_lora_merged" if tag == "e" else "Fp_lora_merged"
        assert np.abs(merged.numpy() - g[mk]).max() <= 1e-10 * np.abs(g[mk]).max()


def test_svd_convention():
    torch.manual_seed(0)
    F = torch.eye(3, dtype=torch.float64) + 0.3 * torch.randn(64, 3, 3, dtype=torch.float64)
    F[-8:, :, 0] *= -1
    U, s, Vh = om.svd3(F)
    assert torch.allclose(U @ torch.diag_embed(s) @ Vh, F, atol=1e-12)
    assert torch.allclose(torch.linalg.det(U), torch.ones(64, dtype=torch.float64))
    assert torch.allclose(torch.linalg.det(Vh), torch.ones(64, dtype=torch.float64))
    assert (s[:, 0] >= s[:, 1]).all() and (s[:, 1] >= s[:, 2].abs()).all()
    assert (torch.sign(s[:, 2]) == torch.sign(torch.linalg.det(F))).all()


def test_svd_adjoint_matches_autograd():
    torch.manual_seed(1)
    F = (torch.eye(3, dtype=torch.float64) + 0.3 * torch.randn(32, 3, 3, dtype=torch.float64)).requires_grad_(True)
    U, s, Vh = om.svd3(F)
    gU, gs, gVh = torch.randn_like(U), torch.randn_like(s), torch.randn_like(Vh)
    (gF,) = torch.autograd.grad((U * gU).sum() + (s * gs).sum() + (Vh * gVh).sum(), F)
    mine = om.svd3_adjoint(U.detach(), s.detach(), Vh.detach(), gU, gs, gVh)
    assert torch.allclose(mine, gF, rtol=1e-9, atol=1e-9)
