"""Every BASELINE.json configuration at FULL SIZE on the GPU (bb, jd, sf, burger, stress), through size-independent
properties - conservation, particle-order invariance, fused == per-operator, stripe re-assembly, background linearity - plus
one whole frame forward + backward; and oracle-sized parity cases with the settings those configurations differ by
(sand plasticity, SH degree 0 on a black background, 64^3 grid)."""
import numpy as np
import pytest
import torch

from oracle import mpm as om
from gpu_util import dev, rel_max, abs_max, measured, mpm_case, build_model, build_statics

pytestmark = pytest.mark.gpu

CONFIGS = ["bb", "jd", "sf", "burger", "stress"]


@pytest.fixture(scope="module", params=CONFIGS)
def rt(request):
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    scene = synth.make_scene(request.param)
    r = SceneRuntime(scene, dev(), fused=True)
    yield r
    del r
    torch.cuda.empty_cache()


def test_config_matches_baseline_json(rt):
    c = rt.scene.cfg
    expect = {"bb": (8_000, 64, 16_000, 256, 256, 3, "jelly"), "jd": (50_000, 128, 100_000, 800, 800, 3, "jelly"),
              "sf": (100_000, 128, 100_000, 800, 800, 3, "sand"), "burger": (80_000, 128, 200_000, 1920, 1080, 0, "jelly"),
              "stress": (1_000_000, 256, 500_000, 1920, 1080, 3, "jelly")}[rt.scene.name]
    assert (c["N"], c["G"], c["K"], c["W"], c["H"], c["sh"], c["mat"]) == expect
    assert float(rt.background.sum()) == (0.0 if rt.scene.name == "burger" else 3.0)       # burger renders on black


def test_p2g_conserves_mass_and_momentum(rt):
    from neuma_amd.sim import MPMDiffSim
    N = rt.N
    v = torch.randn(N, 3, generator=torch.Generator().manual_seed(0)).to(dev())
    zero = torch.zeros(N, 3, 3, device=dev())
    with torch.no_grad():
        MPMDiffSim(rt.model)(rt.statics, rt.x0, v, zero, rt.F0, zero)
    mv, m, _ = rt.model.grid_export()
    pm = (rt.statics.vol * rt.statics.rho).double()
    assert abs(float(m.double().sum()) - float(pm.sum())) < 1e-5 * float(pm.sum())
    assert float((mv.double().sum((0, 1, 2)) - (pm[:, None] * v.double()).sum(0)).abs().max()) < 1e-4 * float(pm.sum())
    nb, nm = rt.model.grid_stats()
    assert nm == int((m > 0).sum()) and nb * 64 >= nm and nm < 0.02 * rt.scene.cfg["G"] ** 3      # ~1 % of the grid


def test_substep_is_invariant_under_particle_order(rt):
    from neuma_amd.sim import MPMDiffSim
    N = rt.N
    g = torch.Generator().manual_seed(1)
    v = (0.3 * torch.randn(N, 3, generator=g)).to(dev())
    C = (0.5 * torch.randn(N, 3, 3, generator=g)).to(dev())
    F = (torch.eye(3) + 0.05 * torch.randn(N, 3, 3, generator=g)).to(dev())
    S = (100.0 * torch.randn(N, 3, 3, generator=g)).to(dev())
    with torch.no_grad():
        a = [t.clone() for t in MPMDiffSim(rt.model, reorder=False)(rt.statics, rt.x0, v, C, F, S)]
        perm = torch.randperm(N, generator=g).to(dev())
        b = MPMDiffSim(rt.model, reorder="auto")(rt.statics, *[t[perm].contiguous() for t in (rt.x0, v, C, F, S)])
    for x, y, tol in zip(a, b, [2e-7, 2e-7, 7e-7, 7e-7]):      # measured 6e-8 | 3e-8 | 2.1e-7 | 2.0e-7
        assert abs_max(x[perm], y) / max(1.0, float(x.abs().max())) < tol


def test_constitutive_nets_frame_indifference_and_plastic_flow(rt):
    N = min(rt.N, 200_000)
    g = torch.Generator().manual_seed(2)
    F = (torch.eye(3) + 0.08 * torch.randn(N, 3, 3, generator=g)).to(dev())
    Q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    if torch.linalg.det(Q) < 0:
        Q[:, 0] *= -1
    Q = Q.float().to(dev())
    with torch.no_grad():
        s, sq = rt.elasticity(F), rt.elasticity(Q @ F)
        p, pq = rt.plasticity(F), rt.plasticity(Q @ F)
    assert rel_max(sq, Q @ s @ Q.T) < 5e-6      # measured 1.1e-06
    assert abs_max(pq, Q @ p) < 1.5e-6      # measured 3.6e-07
    assert float((p - F).abs().max()) > 1e-6          # the plasticity net is not the identity (sand: sf)


def test_fused_rollout_matches_per_operator_path(rt):
    """States and ALL gradients (inputs and LoRA factors) of the fused node against the per-operator classes, at full size.
    The comparison runs from a deformed state under random output weights: from the rest state with a plain sum as the loss
    dL/dF, dL/dC are residues of node sums that cancel, and two runs of the SAME path differ by up to 15 % there (atomics
    order; tools/exp_gradcheck_full.py) - with a well-conditioned loss both paths agree to 5e-5, and that is what is asked."""
    S = min(rt.S, 3)
    old = rt.S
    rt.S = rt.sim_fused.substeps = S
    g = torch.Generator().manual_seed(4)
    F0 = (torch.eye(3) + 0.05 * torch.randn(rt.N, 3, 3, generator=g)).to(dev())
    wts = [torch.randn(sh, generator=g).to(dev()) for sh in ((rt.N, 3), (rt.N, 3), (rt.N, 3, 3), (rt.N, 3, 3))]
    try:
        res = {}
        for fused in (True, False):
            rt.fused = fused
            for p in rt.parameters():
                p.grad = None
            ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
            out = rt.rollout(*ins)
            sum((o * w).sum() for o, w in zip(out, wts)).backward()
            res[fused] = ([o.detach().clone() for o in out], [t.grad.clone() for t in ins + rt.parameters()])
        for x, y, tol in zip(res[True][0], res[False][0], [2e-7, 3e-6, 2e-5, 3e-6]):      # measured 6e-8 | 9.8e-7 | <= 5.4e-6 over four runs | 9.8e-7
            assert abs_max(x, y) / max(1.0, float(y.abs().max())) < tol
        for a, b in zip(res[True][1], res[False][1]):
            assert torch.isfinite(a).all() and float(b.abs().max()) > 0 and rel_max(a, b) < 3e-4
        # rest state, plain sum: the parameter gradients (sums over all particles) are well conditioned there too
        for fused in (True, False):
            rt.fused = fused
            for p in rt.parameters():
                p.grad = None
            out = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
            (out[0].sum() + (out[3] ** 2).sum()).backward()
            res[fused] = [p.grad.clone() for p in rt.parameters()]
        for a, b in zip(res[True], res[False]):
            assert rel_max(a, b) < 1e-3          # (gradients of 1e-6 at rest: the stated gradient tolerance)
    finally:
        rt.S = rt.sim_fused.substeps = old
        rt.fused = True


def test_render_stripes_background_linearity_and_gradients(rt):
    from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F
    with torch.no_grad():
        means = compute_bindings_xyz(rt.x0 + 0.001, rt.x0, rt.gaussians.get_xyz, rt.bindings)
        dg = compute_bindings_F(rt.F0, rt.bindings)
    m = means.clone().requires_grad_(True)
    full = rt.render_view(m, dg, 0)
    H, rows = full.shape[1], rt.tile_rows
    assert tuple(full.shape) == (3, rt.scene.cfg["H"], rt.scene.cfg["W"])
    gw = torch.randn(full.shape, generator=torch.Generator().manual_seed(3)).to(dev())
    (gfull,) = torch.autograd.grad((full * gw).sum(), m)
    acc, gacc = torch.zeros_like(full), torch.zeros_like(gfull)
    for r0, r1 in [(0, rows // 3), (rows // 3, rows // 2), (rows // 2, rows)]:
        part = rt.render_view(m, dg, 0, tile_rows=(r0, r1))
        y0, y1 = r0 * 16, min(H, r1 * 16)
        # (bit for bit when the stripe and the full view are composited with the same plan; a few-tile view is split into
        # list segments whose length depends on the stripe: transmittance products in another order, 2e-6)
        assert abs_max(part[:, y0:y1], full[:, y0:y1]) < 3e-6
        acc += part.detach()
        gacc += torch.autograd.grad((part * gw).sum(), m)[0]
    assert abs_max(acc, full.detach()) < 3e-6 and rel_max(gacc, gfull) < 5e-5      # measured 1.4e-05
    bg0 = rt.background
    imgs = {}
    for name, val in (("black", 0.0), ("grey", 0.5), ("white", 1.0)):
        rt.background = torch.full((3,), val, device=dev())
        imgs[name] = rt.render_view(means, dg, 0)
    rt.background = bg0
    assert abs_max(imgs["grey"], 0.5 * (imgs["black"] + imgs["white"])) < 5e-7        # out = C + T_final * bg      # measured 1.2e-07
    assert float(full.min()) >= 0.0 and torch.isfinite(full).all() and torch.isfinite(gfull).all()
    assert float((imgs["white"] - imgs["black"]).max()) > 0.5                           # some background shows: the body does not fill the frame


def test_whole_frame_forward_backward(rt):
    """S substeps + V renders + loss, forward and backward, from the rest state and from a deformed one.  (From rest a
    single-substep frame has an exactly zero weight gradient: F = I makes all 13 invariants zero, the nets have no bias, so
    every activation and with it every dL/dW vanishes - the S = 1 configurations only train through the later frames of a
    video.  The deformed start is where the gradient must be alive.)"""
    for kind in ("rest", "deformed"):
        rt.set_start_state(kind)
        rt.make_ground_truth()
        for p in rt.parameters():
            p.grad = None
        res = rt.frame()
        grads = [p.grad for p in rt.parameters()]
        assert torch.isfinite(res.loss) and float(res.loss) > 0
        assert all(g is not None and torch.isfinite(g).all() for g in grads)
        assert torch.isfinite(res.x).all() and torch.isfinite(res.F).all()
        if kind == "deformed" or rt.S > 1:
            assert any(float(g.abs().max()) > 0 for g in grads)
    rt.set_start_state("rest")


# ---------------------------------------------------------------- oracle-sized parity with the configurations' settings
def test_sand_plasticity_rollout_matches_oracle_chain():
    """sf: the sand checkpoint (the plasticity return mapping does real work) through the fused roll-out vs the fp64 chain."""
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from test_gpu_rollout import _oracle_rollout
    S = 3
    rt = SceneRuntime(synth.make_scene("tiny", override=dict(S=S, mat="sand")), dev(), fused=True)
    g = torch.Generator().manual_seed(4)
    F0 = (torch.eye(3) + 0.05 * torch.randn(rt.N, 3, 3, generator=g)).to(dev())
    ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
    outs = rt.rollout(*ins)
    gws = [torch.randn(o.shape, generator=g) for o in outs]
    grads = torch.autograd.grad(sum((o * w.to(dev())).sum() for o, w in zip(outs, gws)), ins)
    oins = [t.detach().cpu().double().requires_grad_(True) for t in ins]
    oouts, We, Wp = _oracle_rollout(rt, S, *oins)
    for nme, a, b, tol in zip("xvCF", outs, oouts, [3e-7, 3e-6, 5e-6, 1.5e-6]):      # measured 8.5e-8 | 9.5e-7 | 1.5e-6 | 3.8e-7
        assert abs_max(a, b) / max(1.0, float(b.abs().max())) < tol, nme
    og = torch.autograd.grad(sum((o * w.double()).sum() for o, w in zip(oouts, gws)), oins)
    for nme, a, b in zip("xvCF", grads, og):
        assert rel_max(a, b) < 5e-4, nme      # measured 1.3e-04
    with torch.no_grad():       # sand really flows: the plastic correction is far above round-off
        assert float((rt.plasticity(F0) - F0).abs().max()) > 1e-5


def test_sh_degree_0_on_black_background_matches_oracle_image():
    """burger: SH degree 0 (one coefficient per channel), black background."""
    import math
    from neuma_amd import synth
    from neuma_amd.harness import SceneRuntime
    from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F
    from oracle import raster as orr
    rt = SceneRuntime(synth.make_scene("tiny", override=dict(sh=0, bg="black")), dev(), fused=True)
    assert float(rt.background.sum()) == 0.0 and rt._shs.shape[1] == 1
    with torch.no_grad():
        x, v, C, F = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
        means3D = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
        dg = compute_bindings_F(F, rt.bindings)
    m = means3D.clone().requires_grad_(True)
    img = rt.render_view(m, dg, 1)
    cam = rt.cameras[1]
    s = orr.Settings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), rt.background.cpu().double(), 1.0,
                     cam.world_view_transform.cpu().double(), cam.full_proj_transform.cpu().double(), 0, cam.camera_center.cpu().double())
    om_ = means3D.cpu().double().requires_grad_(True)
    cov = orr.deform_cov_by_F(rt._cov.cpu().double(), dg.cpu().double())
    oimg, _ = orr.render(s, om_, cov, rt._opacity.cpu().double(), shs=rt._shs.cpu().double())
    assert abs_max(img, oimg) < 2e-6      # measured 6.0e-07
    assert measured((img.detach().cpu().double() - oimg.detach()).abs().mean(), "mean abs image error") < 2e-7      # measured 2.5e-08
    gw = torch.randn(img.shape, generator=torch.Generator().manual_seed(9))
    (g,) = torch.autograd.grad((img * gw.to(dev())).sum(), m)
    (og,) = torch.autograd.grad((oimg * gw.double()).sum(), om_)
    assert rel_max(g, og) < 7e-6      # measured 2.1e-06


@pytest.mark.parametrize("bc", ["noslip", "freeslip"])
def test_substep_on_a_64_grid_vs_oracle(bc):
    """bb: 64^3 grid."""
    from neuma_amd.sim import MPMDiffSim
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=8192, G=64, bc=bc)
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    ins = [t.float().to(dev()).requires_grad_(True) for t in (x, v, C, F, S)]
    outs = MPMDiffSim(model, reorder=False)(st, *ins)
    oins = [t.detach().cpu().double().requires_grad_(True) for t in ins]
    oo = om.step(const, vol, rho, clip, en, *oins)
    e = en != 0
    for a, b, tol, rel in zip(outs, oo, [2e-7, 7e-7, 1.5e-6, 1e-6], [False, True, True, False]):      # measured 3e-8 | 2.0e-7 | 3.4e-7 | 2.4e-7
        err = rel_max(a[e], b[e]) if rel else abs_max(a[e], b[e])
        assert err < tol
    gws = [torch.randn(o.shape, generator=torch.Generator().manual_seed(3)) for o in outs]
    grads = torch.autograd.grad(sum((o * w.to(dev())).sum() for o, w in zip(outs, gws)), ins)
    og = torch.autograd.grad(sum((o * w.double()).sum() for o, w in zip(oo, gws)), oins)
    for a, b in zip(grads, og):
        assert rel_max(a, b) < 1.5e-6      # measured 3.4e-07
