"""GPU parity: MLS-MPM substep (p2g / grid_op / g2p and adjoints) through the C ABI vs the fp64 oracle.
Tolerances from SURVEY.md §8d: single step atol x 5e-7, v 2e-6 (scaled by |v|max), C 5e-5*|C|scale, F 5e-7;
gradients max-norm relative 2e-3."""
import numpy as np
import pytest
import torch

from oracle import mpm as om
from gpu_util import dev, rel_max, abs_max, mpm_case, build_model, build_statics, parity

pytestmark = pytest.mark.gpu


def _gpu_step(model, st, x, v, C, F, S, sim=None):
    from neuma_amd.sim import MPMDiffSim
    sim = sim or MPMDiffSim(model, reorder=False)       # the kernels see the particles in exactly the order given here
    ins = [t.float().to(dev()).requires_grad_(True) for t in (x, v, C, F, S)]
    outs = sim(st, *ins)
    return ins, outs


@pytest.mark.parametrize("bc", ["noslip", "freeslip"])
@pytest.mark.parametrize("N,G", [(4096, 32), (3001, 16)])
def test_forward_one_step(bc, N, G):
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=N, G=G, bc=bc)
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    ins, outs = _gpu_step(model, st, x, v, C, F, S)
    # the fp32 inputs are what the GPU saw: run the oracle on exactly those values
    xi, vi, Ci, Fi, Si = [t.detach().cpu().double() for t in ins]
    (ox, ov, oC, oF), (gmv, gm, gv) = om.step(const, vol, rho, clip, en, xi, vi, Ci, Fi, Si, return_grid=True)
    mv, m, vg = model.grid_export()
    # node velocities are compared mass-weighted: at nodes whose mass nearly cancels (negative B-spline lobes of
    # particles within half a cell of the wall) v = mv/m amplifies fp32 summation-order noise, but such nodes carry
    # no weight in g2p
    assert rel_max(vg * m[..., None], gv * gm[..., None]) < 7e-7      # measured 1.8e-07
    e = (en != 0)
    case = f"substep N={N} G={G} {bc} vs fp64 oracle"
    parity(case, "x", abs_max(outs[0][e], ox[e]), 1e-7)            # measured 3.0e-8 (bounds: 3x measured, round 4)
    parity(case, "v", rel_max(outs[1][e], ov[e]), 7e-7)            # 2.1e-7
    parity(case, "C", rel_max(outs[2][e], oC[e]), 1e-6)            # 2.8e-7
    parity(case, "F", abs_max(outs[3][e], oF[e]), 7e-7)            # 2.3e-7
    parity(case, "grid m", rel_max(m, gm), 1e-7)                  # 3.0e-8
    parity(case, "grid mv", rel_max(mv, gmv), 4e-7)               # 1.2e-7
    # disabled particles: the out-of-place sim returns its fresh next state for them (zeros, F = I), as the reference does
    d = ~e
    assert float(outs[0].detach().cpu()[d].abs().max()) == 0.0 and float(outs[1].detach().cpu()[d].abs().max()) == 0.0
    assert torch.equal(outs[3].detach().cpu()[d], torch.eye(3).expand(int(d.sum()), 3, 3))
    nb, nm = model.grid_stats()
    assert nm == int((gm > 0).sum())
    assert nb * 64 >= nm


@pytest.mark.parametrize("bc", ["noslip", "freeslip"])
def test_backward_one_step(bc):
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=4096, G=32, bc=bc)
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    ins, outs = _gpu_step(model, st, x, v, C, F, S)
    torch.manual_seed(11)
    gws = [torch.randn(o.shape) for o in outs]
    loss = sum((o * g.to(dev())).sum() for o, g in zip(outs, gws))
    grads = torch.autograd.grad(loss, ins)
    oins = [t.detach().cpu().double().requires_grad_(True) for t in ins]
    oo = om.step(const, vol, rho, clip, en, *oins)
    ol = sum((o * g.double()).sum() for o, g in zip(oo, gws))
    og = torch.autograd.grad(ol, oins)
    names = ["x", "v", "C", "F", "stress"]
    for nme, a, b in zip(names, grads, og):
        parity(f"one-step adjoint N=4096 G=32 {bc} vs fp64 autograd of the oracle", f"dL/d{nme} (rel)", rel_max(a, b), 8e-6)      # measured: dL/dx 2.5e-6, the others <= 4e-7
        assert torch.isfinite(a).all()
    assert (grads[0].cpu()[en == 0] == 0).all() and (grads[4].cpu()[en == 0] == 0).all()


def test_unsorted_particles_take_the_fallback_path_with_same_result():
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=8192, G=64, near_wall=False, disabled=False)
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    _, outs = _gpu_step(model, st, x, v, C, F, S)
    perm = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(5))
    ins_p, outs_p = _gpu_step(model, st, x[perm], v[perm], C[perm], F[perm], S[perm])
    for a, b in zip(outs, outs_p):
        assert abs_max(a[perm.to(dev())], b) / max(1.0, float(a.abs().max())) < 1e-6      # measured 3.2e-07
    g = torch.autograd.grad(outs_p[1].sum() + outs_p[3].sum(), ins_p)
    assert all(torch.isfinite(t).all() for t in g)


@pytest.mark.parametrize("nclusters", [1, 2, 3, 8, 24])
def test_scatter_modes_on_chunks_made_of_disjoint_clusters(nclusters):
    """Every 256-particle workgroup chunk is made of `nclusters` compact clusters far apart from each other: 1 -> plain
    tile, 2-3 -> axis-compressed tile, 8 -> per-wave boxes, 24 -> per-wave boxes with several passes (+ leftovers through
    global atomics).  Forward grid, next state and the gradients (g2p adjoint scatter uses the same code) vs the oracle."""
    G = 64
    const = om.MPMConstant(num_grids=G, dt=1e-3, bound=1, gravity=(0.0, -9.8, 0.0), eps=6e-7, bc="noslip")
    g = torch.Generator().manual_seed(nclusters)
    nchunks, per = 6, 256 // nclusters
    centers = 0.15 + 0.7 * torch.rand(nchunks * nclusters, 3, generator=g, dtype=torch.float64)
    xs = []
    for c in range(nchunks):
        for k in range(nclusters):
            xs.append(centers[c * nclusters + k] + (1.5 / G) * (torch.rand(per, 3, generator=g, dtype=torch.float64) - 0.5))
        if per * nclusters < 256:      # pad the chunk to 256 so that chunks stay aligned with workgroups
            xs.append(centers[c * nclusters] + (1.5 / G) * (torch.rand(256 - per * nclusters, 3, generator=g, dtype=torch.float64) - 0.5))
    x = torch.cat(xs)
    N = x.shape[0]
    v = torch.randn(N, 3, generator=g, dtype=torch.float64)
    C = torch.randn(N, 3, 3, generator=g, dtype=torch.float64)
    F = torch.eye(3, dtype=torch.float64)[None] + 0.05 * torch.randn(N, 3, 3, generator=g, dtype=torch.float64)
    S = 20.0 * torch.randn(N, 3, 3, generator=g, dtype=torch.float64)
    vol = torch.full((N,), (const.dx / 2) ** 3, dtype=torch.float64)
    rho = torch.full((N,), 1000.0, dtype=torch.float64)
    clip = torch.full((N,), 0.1, dtype=torch.float64)
    en = torch.ones(N, dtype=torch.int32)
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    ins, outs = _gpu_step(model, st, x, v, C, F, S)
    xi, vi, Ci, Fi, Si = [t.detach().cpu().double().requires_grad_(True) for t in ins]
    (ox, ov, oC, oF), (gmv, gm, gv) = om.step(const, vol, rho, clip, en, xi, vi, Ci, Fi, Si, return_grid=True)
    mv, m, vg = model.grid_export()
    assert rel_max(m, gm) < 5e-7 and rel_max(mv, gmv) < 5e-7      # measured 1.2e-07
    assert abs_max(outs[0], ox) < 2e-7 and rel_max(outs[1], ov) < 1e-6 and rel_max(outs[2], oC) < 1e-6      # measured 3.0e-08
    w = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in outs]
    go = torch.autograd.grad(sum((o * wi).sum() for o, wi in zip((ox, ov, oC, oF), w)), [xi, vi, Ci, Fi, Si])
    gg = torch.autograd.grad(sum((o * wi.float().to(dev())).sum() for o, wi in zip(outs, w)), ins)
    for a, b in zip(gg, go):
        assert rel_max(a, b) < 1.5e-6      # measured 3.5e-07
    nb, nm = model.grid_stats()
    assert nm == int((gm > 0).sum())


def test_per_operator_sims_reorder_shuffled_particles_transparently():
    """MPMDiffSim / MPMCacheDiffSim (what finetune.py drives, on particles prepare_simulation_data has shuffled) run on an
    internally Hilbert-sorted copy: same outputs and gradients, in the caller's order, as with re-ordering switched off."""
    from neuma_amd.sim import MPMCacheDiffSim, MPMDiffSim
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=6000, G=32, near_wall=False)
    perm = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(9))
    x, v, C, F, S, en = x[perm], v[perm], C[perm], F[perm], S[perm], en[perm].contiguous()
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    ref_in, ref_out = _gpu_step(model, st, x, v, C, F, S)                       # reorder=False
    w = [torch.randn(o.shape, generator=torch.Generator().manual_seed(1)).to(dev()) for o in ref_out]
    g_ref = torch.autograd.grad(sum((o * wi).sum() for o, wi in zip(ref_out, w)), ref_in)
    for sim, args in ((MPMDiffSim(model), ()), (MPMCacheDiffSim(model, 4), (2,))):
        ins = [t.float().to(dev()).requires_grad_(True) for t in (x, v, C, F, S)]
        outs = sim(st, *args, *ins)
        assert torch.is_tensor(sim.order.perm)                                   # shuffled input: permutation in use
        for a, b in zip(outs, ref_out):
            assert abs_max(a, b) / max(1.0, float(b.abs().max())) < 1.5e-6      # measured 3.5e-07
        g = torch.autograd.grad(sum((o * wi).sum() for o, wi in zip(outs, w)), ins)
        for a, b in zip(g, g_ref):
            assert rel_max(a, b) < 1e-6      # measured 2.9e-07
    e = (en != 0).to(dev())
    assert int((~e).sum()) > 0 and float(outs[0][~e].abs().max()) == 0.0         # disabled particles: the fresh next state's rows


def test_forward_sim_reorders_shuffled_particles_in_place():
    """MPMForwardSim (inference path, in place on the caller's state): shuffled particles are stepped on a sorted internal
    copy and written back into the same buffers."""
    from neuma_amd.sim import MPMForwardSim
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=5000, G=32, near_wall=False)
    perm = torch.randperm(x.shape[0], generator=torch.Generator().manual_seed(11))
    x, v, C, F, S, en = x[perm], v[perm], C[perm], F[perm], S[perm], en[perm].contiguous()
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    res = {}
    for tag, reorder in (("plain", False), ("sorted", "auto")):
        sim = MPMForwardSim(model, reorder=reorder)
        state = model.state(x.shape[0])
        state.from_torch(x=x.float().to(dev()), v=v.float().to(dev()), C=C.float().to(dev()), F=F.float().to(dev()),
                         stress=S.float().to(dev()))
        held = state.to_torch()[0]                       # a handle the caller keeps: must see the update
        for _ in range(3):
            outs = sim(st, state)
        assert outs[0].data_ptr() == held.data_ptr()
        assert torch.is_tensor(sim.order.perm) == (reorder == "auto")
        res[tag] = [o.clone() for o in outs]
    for a, b in zip(res["sorted"], res["plain"]):
        assert abs_max(a, b) / max(1.0, float(b.abs().max())) < 3e-6      # measured 9.2e-07


def test_in_place_forward_sim_and_extra():
    from neuma_amd.sim import MPMForwardSim, MPMExtraSim
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=2048, G=32, disabled=False)
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    state = model.state(x.shape[0])
    state.from_torch(x=x.float().to(dev()), v=v.float().to(dev()), C=C.float().to(dev()), F=F.float().to(dev()),
                     stress=S.float().to(dev()))
    xi, vi, Ci, Fi, Si = [t.clone().cpu().double() for t in state.to_torch()]
    ox, ov, oC, oF = om.step(const, vol, rho, clip, en, xi, vi, Ci, Fi, Si)
    # passive extra particles sampled on the same grid (mpm.py:260-277) — before the in-place step mutates state
    M = 300
    xe = x[:M] + 0.3 * const.dx
    st_e = build_statics(model, vol[:M], rho[:M], clip[:M], en[:M], dev())
    state_e = model.state(M)
    state_e.from_torch(x=xe.float().to(dev()))
    Fe = state_e.particle.F.clone().cpu().double()
    x_extra = MPMExtraSim(model)(st, state, st_e, state_e)
    gmv, gm = om.p2g(const, vol, rho, en, xi, vi, Ci, Si)
    gv = om.grid_op(const, gmv, gm)
    oxe, _, _, _ = om.g2p(const, clip[:M], en[:M], xe.float().double(), Fe, gv)
    assert abs_max(x_extra, oxe) < 2e-7      # measured 3.0e-08
    x2, v2, C2, F2 = MPMForwardSim(model)(st, state)
    assert abs_max(x2, ox) < 2e-7 and rel_max(v2, ov) < 7e-7 and rel_max(C2, oC) < 1.5e-6 and abs_max(F2, oF) < 7e-7      # measured 3.0e-08


def test_fifty_step_rollout_drift_vs_fp64():
    """Contact-free roll-out with a smooth neo-Hookean-like stress; horizon tolerance of SURVEY §8d."""
    from neuma_amd.sim import MPMForwardSim
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=4096, G=32, near_wall=False, disabled=False)
    v = 0.1 * v
    C = torch.zeros_like(C)
    F = torch.eye(3, dtype=torch.float64).repeat(x.shape[0], 1, 1)
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    state = model.state(x.shape[0])
    state.from_torch(x=x.float().to(dev()), v=v.float().to(dev()), C=C.float().to(dev()), F=F.float().to(dev()))
    sim = MPMForwardSim(model)
    xo, vo, Co, Fo = [t.clone().cpu().double() for t in state.to_torch()[:4]]

    def stress_of(Fm):   # mu (F F^T - I)
        return 200.0 * (Fm @ Fm.transpose(-1, -2) - torch.eye(3, dtype=Fm.dtype, device=Fm.device))

    for _ in range(50):
        state.from_torch(stress=stress_of(state.particle.F))
        sim(st, state)
        xo, vo, Co, Fo = om.step(const, vol, rho, clip, en, xo, vo, Co, Fo, stress_of(Fo))
    xg, vg, Cg, Fg, _ = state.to_torch()
    assert abs_max(xg, xo) < 2e-6 and abs_max(vg, vo) < 2e-6 and abs_max(Cg, Co) < 3e-5 and abs_max(Fg, Fo) < 1e-5      # measured 6.6e-07


def test_error_paths():
    from neuma_amd.sim import MPMModelBuilder
    from neuma_amd import NeumaHipError
    cfg = dict(gravity=[0, 0, 0], bc="sticky", num_grids=16, dt=1e-3, bound=1, eps=0.0)
    with pytest.raises(ValueError):
        MPMModelBuilder().parse_cfg(cfg).finalize(dev())                 # mpm.py:550
    with pytest.raises(RuntimeError):
        MPMModelBuilder().finalize(dev())                                # abstract.py:83-84
    cfg["bc"] = "noslip"
    model = MPMModelBuilder().parse_cfg(cfg).finalize(dev())
    st = model.statics(0)
    s0 = model.state(0)
    model.forward(st, s0, s0)                                             # empty input is a no-op
    model_cpu = MPMModelBuilder().parse_cfg(cfg).finalize("cpu")
    with pytest.raises(NeumaHipError):
        model_cpu.forward(model_cpu.statics(4), model_cpu.state(4), model_cpu.state(4))


def test_per_operator_grid_tape_matches_recompute():
    """MPMSimFunction carries a grid cache record in the `tape` slot: gradients equal the reference-style recompute, also
    when the record overflows (capacity 1 block -> transparent fallback)."""
    from neuma_amd.sim import MPMCacheDiffSim
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=4096, G=32)
    res = {}
    for tag, cache in (("recompute", 0), ("auto", "auto"), ("overflow", 1)):
        model = build_model(const, dev())
        model.grid_cache = cache
        st = build_statics(model, vol, rho, clip, en, dev())
        sim = MPMCacheDiffSim(model, 3, reorder=False)
        ins = [t.float().to(dev()).requires_grad_(True) for t in (x, v, C, F, S)]
        a = sim(st, 0, *ins)
        b = sim(st, 1, a[0], a[1], a[2], a[3], ins[4])          # second substep: the auto capacity is known by now
        if cache == "auto":
            assert model._cache_blocks == int(1.5 * model.grid_stats()[0]) + 64 or model._cache_blocks > 64
        res[tag] = torch.autograd.grad(b[0].sum() + (b[3] * b[3]).sum() + b[1].sum(), ins)
    for tag in ("auto", "overflow"):
        for g, r in zip(res[tag], res["recompute"]):
            assert rel_max(g, r) < 1.5e-6, tag      # measured 3.4e-07


@pytest.mark.parametrize("mode", ["sort", "f64"])
def test_both_scatter_paths_give_the_same_substep_and_gradients(mode, monkeypatch):
    """NEUMA_SCATTER (read when a model is created): "f64" (default) = fp64 LDS atomics into the workgroup tile, "sort" = round
    3's counting sort + barrier-separated pushes, kept for A/B measurements.  Both against the fp64 oracle, forward and adjoint,
    incl. a disabled span and particles at the walls."""
    monkeypatch.setenv("NEUMA_SCATTER", mode)
    const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=5000, G=32, bc="freeslip")
    model = build_model(const, dev())
    st = build_statics(model, vol, rho, clip, en, dev())
    ins, outs = _gpu_step(model, st, x, v, C, F, S)
    xi, vi, Ci, Fi, Si = [t.detach().cpu().double().requires_grad_(True) for t in ins]
    oo = om.step(const, vol, rho, clip, en, xi, vi, Ci, Fi, Si)
    e = (en != 0)
    case = f"substep N=5000 G=32 freeslip, NEUMA_SCATTER={mode}, vs fp64 oracle"
    parity(case, "x", abs_max(outs[0][e], oo[0][e]), 1e-7)
    parity(case, "v", rel_max(outs[1][e], oo[1][e]), 7e-7)
    parity(case, "C", rel_max(outs[2][e], oo[2][e]), 1e-6)
    parity(case, "F", abs_max(outs[3][e], oo[3][e]), 7e-7)
    torch.manual_seed(3)
    gws = [torch.randn(o.shape) for o in outs]
    grads = torch.autograd.grad(sum((o * g.to(dev())).sum() for o, g in zip(outs, gws)), ins)
    og = torch.autograd.grad(sum((o * g.double()).sum() for o, g in zip(oo, gws)), [xi, vi, Ci, Fi, Si])
    for nme, a, b in zip(["x", "v", "C", "F", "stress"], grads, og):
        parity(case, f"dL/d{nme} (rel)", rel_max(a, b), 3e-6)      # measured <= 5.8e-7 (fp64 autograd of the oracle)
