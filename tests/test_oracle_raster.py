"""Oracle rasterizer: conventions pinned against the reference's pure-python camera / SH / loss modules
(tests/golden/camera_sh_golden.npz) plus analytic checks."""
import math

import numpy as np
import torch

from oracle import raster as orr


def test_camera_convention_matches_reference(golden_dir):
    g = np.load(golden_dir / "camera_sh_golden.npz")
    R, T = g["R"], g["T"]
    # reference: world_view = getWorld2View2(R,T).T ; Rt[:3,:3] = R.T ; Rt[:3,3] = T
    Rt = np.eye(4)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = T
    assert np.allclose(Rt.T, g["world_view"], atol=1e-6)
    eye = -R @ T   # camera centre: C = -R_w2c^T t, R_w2c = R.T
    assert np.allclose(eye, g["center"], atol=1e-5)
    # look_at_camera reproduces a reference camera given its pose
    fwd = R[:, 2]
    wv, full, campos = orr.look_at_camera(eye, eye + fwd, -R[:, 1], float(g["fovx"]), float(g["fovy"]), dtype=torch.float64)
    assert np.allclose(wv.numpy(), g["world_view"], atol=1e-6)
    assert np.allclose(full.numpy(), g["full_proj"], atol=1e-5)
    assert np.allclose(campos.numpy(), g["center"], atol=1e-5)


def test_sh_matches_reference(golden_dir):
    g = np.load(golden_dir / "camera_sh_golden.npz")
    dirs = torch.tensor(g["sh_dirs"])
    coef = torch.tensor(g["sh_coef"])            # (32, 3, 16) reference layout [..., C, coeffs]
    shs = coef.permute(0, 2, 1)                  # rasterizer layout (K, 16, 3)
    for deg in range(4):
        rgb, _ = orr.eval_sh_color(deg, shs, dirs, torch.zeros(3, dtype=torch.float64))
        ref = np.maximum(g[f"sh_deg{deg}"] + 0.5, 0.0)
        assert np.allclose(rgb.numpy(), ref, atol=1e-12)


def test_losses_match_reference(golden_dir):
    g = np.load(golden_dir / "camera_sh_golden.npz")
    a, b = torch.tensor(g["loss_a"]), torch.tensor(g["loss_b"])
    assert abs(orr.l1_loss(a, b).item() - float(g["l1"])) < 1e-14
    assert abs(orr.l2_loss(a, b).item() - float(g["l2"])) < 1e-14


def _simple_scene(dtype=torch.float64):
    W = H = 64
    fov = 2 * math.atan(0.5)
    wv, full, campos = orr.look_at_camera([0, 0, -2.0], [0, 0, 0], [0, -1, 0], fov, fov, dtype=dtype)
    s = orr.Settings(H, W, math.tan(fov / 2), math.tan(fov / 2), torch.tensor([0.2, 0.3, 0.4], dtype=dtype), 1.0,
                     wv, full, 0, campos)
    return s


def test_single_gaussian_analytic():
    s = _simple_scene()
    W = s.image_width
    means = torch.tensor([[0.0, 0.0, 0.0]], dtype=torch.float64)
    sig = 0.05
    cov = torch.tensor([[sig * sig, 0, 0, sig * sig, 0, sig * sig]], dtype=torch.float64)
    op = torch.tensor([[0.8]], dtype=torch.float64)
    col = torch.tensor([[1.0, 0.5, 0.25]], dtype=torch.float64)
    img, radii, aux = orr.render(s, means, cov, op, colors_precomp=col, return_aux=True)
    f = W / (2 * s.tanfovx)
    var = (f * sig / 2.0) ** 2 + 0.3
    cx = (W - 1) / 2
    assert int(radii[0]) == math.ceil(3 * math.sqrt(var))
    for (px, py) in [(31, 31), (32, 32), (35, 30), (40, 40)]:
        d2 = (cx - px) ** 2 + (cx - py) ** 2
        a = min(0.99, 0.8 * math.exp(-0.5 * d2 / var))
        if a < 1 / 255:
            a = 0.0
        exp = a * np.array([1.0, 0.5, 0.25]) + (1 - a) * np.array([0.2, 0.3, 0.4])
        assert np.allclose(img[:, py, px].numpy(), exp, atol=1e-12)
    # far corner untouched (outside tile rect) = background
    assert np.allclose(img[:, 0, 0].numpy(), [0.2, 0.3, 0.4])


def test_front_to_back_order_and_termination():
    s = _simple_scene()
    # two opaque-ish gaussians on the axis; nearer one must dominate
    means = torch.tensor([[0.0, 0.0, 0.5], [0.0, 0.0, -0.5]], dtype=torch.float64)
    cov = torch.tensor([[0.01, 0, 0, 0.01, 0, 0.01]] * 2, dtype=torch.float64)
    op = torch.tensor([[0.9], [0.9]], dtype=torch.float64)
    col = torch.tensor([[1.0, 0, 0], [0, 1.0, 0]], dtype=torch.float64)
    img, _ = orr.render(s, means, cov, op, colors_precomp=col)
    c = img[:, 32, 32]
    assert c[1] > c[0]   # green (z=-0.5 is nearer to the camera at z=-2)
    # 60 stacked gaussians: T-termination keeps the image finite and T >= 0
    K = 60
    means = torch.zeros(K, 3, dtype=torch.float64)
    means[:, 2] = torch.linspace(-0.5, 0.5, K)
    img, _, aux = orr.render(s, means, cov[:1].expand(K, 6), torch.full((K, 1), 0.7, dtype=torch.float64),
                             colors_precomp=torch.ones(K, 3, dtype=torch.float64), return_aux=True)
    assert aux["n_contrib"][32, 32] < K
    assert torch.isfinite(img).all()


def test_deform_cov_and_build_cov():
    torch.manual_seed(0)
    sc = torch.rand(5, 3, dtype=torch.float64) + 0.1
    q = torch.randn(5, 4, dtype=torch.float64)
    c6 = orr.build_cov3D(sc, q, 1.5)
    S = orr.cov6_to_mat(c6)
    assert torch.allclose(S, S.transpose(-1, -2))
    ev = torch.linalg.eigvalsh(S)
    assert torch.allclose(ev, torch.sort((1.5 * sc) ** 2, dim=1).values, atol=1e-12)
    F = torch.eye(3, dtype=torch.float64) + 0.2 * torch.randn(5, 3, 3, dtype=torch.float64)
    d6 = orr.deform_cov_by_F(c6, F)
    assert torch.allclose(orr.cov6_to_mat(d6), F @ S @ F.transpose(-1, -2))
