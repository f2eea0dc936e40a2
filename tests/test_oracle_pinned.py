"""The oracle's MPM / SVD-rule / cov-deform restatements against fixtures produced by EXECUTING the reference's own
kernel bodies (tests/golden/gen_mpm_golden.py: /root/reference/modules/nclaw/sim/mpm.py, warp/svd.py,
d3gs/utils/simulation_utils.py run under a scalar numpy stand-in for warp).  fp64 fixtures pin the algorithm to
round-off; fp32 fixtures are the reference's arithmetic in its own precision (sequential atomics order)."""
import numpy as np
import pytest
import torch

from oracle import mpm as om
from oracle import material as omat
from oracle import raster as orr

STEP_TAGS = ["n64_g16_noslip", "n64_g16_freeslip", "n2048_g32_noslip", "n2048_g32_freeslip"]
# MPMModelBuilder.parse_cfg stores gravity as np.float32 (mpm.py:512) - also when the kernels then run in fp64
GRAVITY = (0.0, float(np.float32(-9.8)), 0.0)
GRAD_TAGS = ["n64_g16_noslip", "n64_g16_freeslip", "n384_g16_noslip", "n384_g16_freeslip"]


def load_case(z, dtype=torch.float64):
    G = int(z["in_G"])
    const = lambda bc: om.MPMConstant(num_grids=G, dt=float(z["in_dt"]), bound=1, gravity=GRAVITY, eps=6e-7, bc=bc)  # noqa: E731
    t = lambda k: torch.from_numpy(z["in_" + k]).to(dtype)  # noqa: E731
    return const, t("vol"), t("rho"), t("clip_bound"), torch.from_numpy(z["in_enabled"]), t("x"), t("v"), t("C"), t("F"), t("stress")


@pytest.mark.parametrize("tag", STEP_TAGS)
def test_step_matches_reference_fp64(golden_dir, tag):
    z = np.load(golden_dir / f"mpm_step_{tag}.npz")
    mk, vol, rho, clip, en, x, v, C, F, S = load_case(z)
    const = mk(tag.split("_")[-1])
    (nx, nv, nC, nF), (mv, m, gv) = om.step(const, vol, rho, clip, en, x, v, C, F, S, return_grid=True)
    # grid: every node, including the untouched ones (m == 0 => v = g dt, BC-masked) - mpm.py:382-385 / 411-414
    assert np.abs(m.numpy() - z["f64_m"]).max() <= 1e-13 * np.abs(z["f64_m"]).max()
    assert np.abs(mv.numpy() - z["f64_mv"]).max() <= 1e-12 * np.abs(z["f64_mv"]).max()
    assert np.abs(gv.numpy() - z["f64_gv"]).max() <= 1e-11 * np.abs(z["f64_gv"]).max()
    assert (z["f64_m"] == 0).any() and (z["f64_m"] > 0).any()
    e = en.numpy() != 0
    for name, got in [("x", nx), ("v", nv), ("C", nC), ("F", nF)]:
        ref = z["f64_" + name]
        assert np.abs(got.numpy()[e] - ref[e]).max() <= 1e-11 * max(1.0, np.abs(ref[e]).max()), name
        assert (ref[~e] == -7.0).all(), "reference leaves disabled particles' next state untouched"
    # ... which, for the fresh model.state() an out-of-place sim hands over, means zeros / identity (fixture rows fresh_*)
    fx, fv, fC, fF = om.step(const, vol, rho, clip, en, x, v, C, F, S, fresh=True)
    for name, got in zip(["x", "v", "C", "F"], [fx, fv, fC, fF]):
        assert np.array_equal(got.numpy()[~e], z["fresh_" + name][~e].astype(np.float64)), name
        assert np.array_equal(z["fresh_" + name][e], z["f32_" + name][e]), name
    # the fixture really exercises the special cases
    base = np.trunc(z["in_x"] * int(z["in_G"]) - 0.5)
    assert (base == 0).any() and ((z["in_x"] * int(z["in_G"]) - 0.5) < 0).any(), "int() truncation case present"
    lo = float(clip.max()) / int(z["in_G"])
    assert (np.isclose(z["f64_x"][e], lo) | np.isclose(z["f64_x"][e], 1 - lo)).any(), "position clamp fired"


@pytest.mark.parametrize("tag", STEP_TAGS)
def test_step_fp32_oracle_vs_reference_fp32(golden_dir, tag):
    """Same arithmetic in fp32: differences are summation-order round-off only."""
    z = np.load(golden_dir / f"mpm_step_{tag}.npz")
    mk, vol, rho, clip, en, x, v, C, F, S = load_case(z, torch.float32)
    const = mk(tag.split("_")[-1])
    nx, nv, nC, nF = om.step(const, vol, rho, clip, en, x, v, C, F, S)
    e = en.numpy() != 0
    tol = dict(x=5e-7, v=2e-5, C=5e-5, F=5e-6)          # DESIGN.md §2 single-substep tolerances (v, C relative)
    for name, got in [("x", nx), ("v", nv), ("C", nC), ("F", nF)]:
        ref = z["f32_" + name][e]
        scale = max(1.0, np.abs(ref).max()) if name in ("v", "C") else 1.0
        assert np.abs(got.numpy()[e] - ref).max() <= tol[name] * scale, name


@pytest.mark.parametrize("tag", GRAD_TAGS)
def test_adjoint_matches_central_differences_of_reference(golden_dir, tag):
    z = np.load(golden_dir / f"mpm_grad_{tag}.npz")
    mk, vol, rho, clip, en, x, v, C, F, S = load_case(z)
    const = mk(tag.split("_")[-1])
    leaves = dict(x=x, v=v, C=C, F=F, stress=S)
    for t in leaves.values():
        t.requires_grad_(True)
    nx, nv, nC, nF = om.step(const, vol, rho, clip, en, x, v, C, F, S)
    e = en != 0
    W = {k: torch.from_numpy(z["W_" + k]) for k in ["x", "v", "C", "F"]}
    L = (W["x"][e] * nx[e]).sum() + (W["v"][e] * nv[e]).sum() + (W["C"][e] * nC[e]).sum() + (W["F"][e] * nF[e]).sum()
    grads = torch.autograd.grad(L, list(leaves.values()))
    for (name, _), g in zip(leaves.items(), grads):
        for d, fd, fd4 in zip(z["dir_" + name], z["fd_" + name], z["fd4_" + name]):
            assert abs(fd - fd4) <= 2e-5 * abs(fd), "fixture self-consistency (two step sizes)"
            an = float((g * torch.from_numpy(d)).sum())
            assert abs(an - fd) <= 2e-5 * abs(fd) + 1e-9, (name, an, fd)


def _stress(F, mu, lam):
    J = torch.linalg.det(F)
    I = torch.eye(3, dtype=F.dtype)
    return mu * (F @ F.transpose(1, 2) - I) + lam * torch.log(J)[:, None, None] * I


def test_rollout_forward_inplace_and_extra(golden_dir):
    """12 steps of the reference's MPMForwardSim with span enabling (render.py:304-310 order) and MPMExtraSim."""
    z = np.load(golden_dir / "mpm_rollout.npz")
    const = om.MPMConstant(num_grids=int(z["G"]), dt=float(z["dt"]), bound=1, gravity=GRAVITY, eps=6e-7, bc="noslip")
    t = lambda k: torch.from_numpy(z[k]).double()  # noqa: E731
    x, v = t("x0"), t("v0")
    N = x.shape[0]
    C = torch.zeros(N, 3, 3, dtype=torch.float64)
    F = torch.eye(3, dtype=torch.float64).repeat(N, 1, 1)
    vol, rho, clip = t("vol"), t("rho"), t("clip_bound")
    en = torch.from_numpy(z["enabled0"]).clone()
    sections, spans = z["sections"], z["spans"]
    xe = t("xe0")
    for step in range(1, 13):
        S = _stress(F, float(z["mu"]), float(z["lam"]))
        if step in (3, 9):
            mv, m = om.p2g(const, vol, rho, en, x, v, C, S)
            gv = om.grid_op(const, mv, m)
            ne = xe.shape[0]
            xe, _, _, _ = om.g2p(const, torch.full((ne,), 0.1, dtype=torch.float64), torch.ones(ne, dtype=torch.int32), xe,
                                 torch.eye(3, dtype=torch.float64).repeat(ne, 1, 1), gv)
            assert np.abs(xe.numpy() - z[f"f64_xe_{step}"]).max() < 1e-13
        x, v, C, F = om.step(const, vol, rho, clip, en, x, v, C, F, S)
        off = 0
        for n, (a, b) in zip(sections, spans):                      # mpm.py:67-72 after the step
            en[off:off + n] = 1 if a <= step < b else 0
            off += n
        if step in (1, 5, 12):
            assert (en.numpy() == z[f"f64_enabled_{step}"]).all()
            for name, got in [("x", x), ("v", v), ("C", C), ("F", F)]:
                ref = z[f"f64_{name}_{step}"]
                assert np.abs(got.numpy() - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (name, step)
    assert float(x[:, 1].min()) < 0.1 and np.abs(z["f64_F_12"] - np.eye(3)).max() > 0.1     # contact + real deformation
    # fp32 execution of the reference stays within the documented roll-out tolerance of the fp64 one
    assert np.abs(z["f32_x_12"] - z["f64_x_12"]).max() < 5e-6


def test_svd_sign_rule(golden_dir):
    """svd.py:61-96 executed on numpy factors (both raw conventions) vs oracle.svd3: sigma, R = U Vh, reconstruction."""
    z = np.load(golden_dir / "svd_rule.npz")
    A = torch.from_numpy(z["A"])
    U, s, Vh = omat.svd3(A)
    for mode in ["numpy", "rot"]:
        rU, rs, rVh = (torch.from_numpy(z[f"{mode}_f64_{k}"]) for k in ["U", "sigma", "Vh"])
        assert torch.allclose(s, rs, atol=1e-13)
        assert (rs[:, 2] < 0).sum() >= 16 and (rs[:, 0] >= rs[:, 1]).all() and (rs[:, 1] >= rs[:, 2].abs()).all()
        assert torch.allclose(torch.linalg.det(rU), torch.ones(len(A), dtype=torch.float64), atol=1e-12)
        assert torch.allclose(torch.linalg.det(rVh), torch.ones(len(A), dtype=torch.float64), atol=1e-12)
        assert torch.allclose(rU @ torch.diag_embed(rs) @ rVh, A, atol=1e-13)
        distinct = ((rs[:, 0] - rs[:, 1]).abs() > 1e-3) & ((rs[:, 1] - rs[:, 2].abs()).abs() > 1e-3)
        # factors are unique up to joint sign flips of column pairs when sigma is simple: compare sign-invariant forms
        assert torch.allclose((U @ Vh)[distinct], (rU @ rVh)[distinct], atol=1e-10)
        for i in range(3):
            Pi = U[:, :, i, None] * Vh[:, None, i, :]
            rPi = rU[:, :, i, None] * rVh[:, None, i, :]
            assert torch.allclose(Pi[distinct], rPi[distinct], atol=1e-9)


def test_cov_deform(golden_dir):
    z = np.load(golden_dir / "cov_deform.npz")
    out = orr.deform_cov_by_F(torch.from_numpy(z["cov6"]), torch.from_numpy(z["F"]))
    assert np.abs(out.numpy() - z["f64_out"]).max() <= 1e-15
    out32 = orr.deform_cov_by_F(torch.from_numpy(z["cov6"]).float(), torch.from_numpy(z["F"]).float())
    assert np.abs(out32.numpy() - z["f32_out"]).max() <= 2e-7 * np.abs(z["f32_out"]).max()
