"""GPU parity against fixtures produced by EXECUTING the reference's own kernel code (tests/golden/gen_mpm_golden.py):
MPM substep + adjoints, in-place forward roll-out with span enabling, passive extra set, SVD det/sign rule,
deform_cov_by_F.  Everything goes through the C ABI (libneuma_hip.so).  fp32 tolerances of DESIGN.md §2."""
import numpy as np
import pytest
import torch

from oracle import mpm as om
from gpu_util import dev, build_model, build_statics, parity, measured

pytestmark = pytest.mark.gpu

STEP_TAGS = ["n64_g16_noslip", "n64_g16_freeslip", "n2048_g32_noslip", "n2048_g32_freeslip"]
GRAD_TAGS = ["n64_g16_noslip", "n64_g16_freeslip", "n384_g16_noslip", "n384_g16_freeslip"]
GRAVITY = (0.0, float(np.float32(-9.8)), 0.0)
# bounds of the comparisons that used to be bare asserts (round 6: logged through gpu_util.parity like the rest; values in DESIGN.md §2)
# measured on MI355X (gpurun_out r06b, profiles/r06_parity_table.md): v * m 4.5e-7; |an - fd| at most 7.2e-3 of the round-5 budget
# (2e-3 |fd| + 2e-6 sum |g||d|) - the budget is now 1/20 of that; x_extra 6.8e-8; sigma 3.8e-7, det 3.6e-7, reconstruction 4.9e-7,
# rotation 2.1e-7, rank-one pieces 8.8e-7; cov 5.3e-8.  Fixture comparisons are deterministic up to the scatters' atomics order.
GV_BOUND = 3e-6
FD_REL, FD_ABS = 1e-4, 1e-7
XE_BOUND = 4e-7
SVD_BOUND = dict(sigma=1.5e-6, det=1.5e-6, recon=2e-6, rot=1e-6, rank1=4e-6)
COV_BOUND = 3e-7


def _case(z, bc):
    const = om.MPMConstant(num_grids=int(z["in_G"]), dt=float(z["in_dt"]), bound=1, gravity=GRAVITY, eps=6e-7, bc=bc)
    t = lambda k: torch.from_numpy(z["in_" + k])  # noqa: E731
    model = build_model(const, dev())
    st = build_statics(model, t("vol"), t("rho"), t("clip_bound"), torch.from_numpy(z["in_enabled"]), dev())
    ins = [t(k).float().to(dev()) for k in ["x", "v", "C", "F", "stress"]]
    return const, model, st, ins


@pytest.mark.parametrize("reorder", [False, "auto"])
@pytest.mark.parametrize("tag", STEP_TAGS)
def test_substep_vs_reference_run(golden_dir, tag, reorder):
    from neuma_amd.sim import MPMDiffSim
    z = np.load(golden_dir / f"mpm_step_{tag}.npz")
    const, model, st, ins = _case(z, tag.split("_")[-1])
    outs = MPMDiffSim(model, reorder=reorder)(st, *ins)
    e = z["in_enabled"] != 0
    assert (~e).sum() > 0
    # Bounds (round 4: <= 3x what was measured on MI355X, DESIGN.md §2 has the table).  Against the reference's fp64 run the
    # error IS the reference's own fp32-vs-fp64 distance on the same fixture (the GPU sees fp32-rounded inputs, like the
    # reference's fp32 run does): SURVEY §8d's figure (x 2e-7, v 2e-6 rel, C 5e-6 rel, F 5e-7) or three times that distance,
    # whichever is larger - on these fixtures the distance is v 7.5e-7..3.2e-6, C 2.1e-6..1.4e-5, F 2.7e-7..3.4e-7.  Against
    # the reference's fp32 run only the summation order differs: measured x 6e-8, v 1.6e-7, C 3e-7, F 2.4e-7.
    floor = {"f64": dict(x=2e-7, v=2e-6, C=5e-6, F=5e-7), "f32": dict(x=2e-7, v=5e-7, C=1e-6, F=7.5e-7)}
    for ref_key in ("f64", "f32"):
        tol = {}
        for name, got in zip(["x", "v", "C", "F"], outs):
            # every row: the rows of disabled particles are what the reference's MPMDiffSim returns for them - its fresh
            # model.state() untouched by g2p (mpm.py:84-93, 443-444; fixture rows `fresh_*`: zeros, F = identity), bit for bit
            ref = z[f"{ref_key}_{name}"].astype(np.float64)
            ref[~e] = z[f"fresh_{name}"][~e]
            scale = max(1.0, np.abs(ref).max()) if name in ("v", "C") else 1.0
            g = got.detach().cpu().double().numpy()
            noise = np.abs(z[f"f32_{name}"].astype(np.float64)[e] - z[f"f64_{name}"].astype(np.float64)[e]).max() / scale
            tol[name] = max(floor[ref_key][name], 3.0 * noise) if ref_key == "f64" else floor[ref_key][name]
            parity(f"substep {tag} vs reference {ref_key} run (reorder={reorder})", name, np.abs(g - ref).max() / scale, tol[name], noise)
            assert np.array_equal(g[~e], z[f"fresh_{name}"][~e].astype(np.float64)), (ref_key, name)
    if reorder is False:
        mv, m, gv = (a.cpu().double().numpy() for a in model.grid_export())
        gn = lambda k: np.abs(z["f32_" + k].astype(np.float64) - z["f64_" + k]).max() / np.abs(z["f64_" + k]).max()  # noqa: E731
        parity(f"substep {tag} grid vs reference f64 run", "m (rel)", np.abs(m - z["f64_m"]).max() / z["f64_m"].max(), max(2e-7, 3 * gn("m")), gn("m"))
        parity(f"substep {tag} grid vs reference f64 run", "mv (rel)", np.abs(mv - z["f64_mv"]).max() / np.abs(z["f64_mv"]).max(),
               max(5e-7, 3 * gn("mv")), gn("mv"))
        w = z["f64_m"][..., None]
        parity(f"substep {tag} grid vs reference f64 run", "v * m (rel)", np.abs(gv * m[..., None] - z["f64_gv"] * w).max() / np.abs(z["f64_gv"] * w).max(), GV_BOUND)
        # untouched nodes: the reference's dense sweep leaves v = BC(g dt) there; the block-sparse grid must export the same
        un = z["f64_m"] == 0
        parity(f"substep {tag} grid vs reference f64 run", "v of untouched nodes (abs)", np.abs(gv[un] - z["f64_gv"][un]).max(), 2e-9)      # (8.6e-10: g dt in fp32 against fp64)


@pytest.mark.parametrize("tag", GRAD_TAGS)
def test_adjoints_vs_central_differences_of_the_reference(golden_dir, tag):
    from neuma_amd.sim import MPMDiffSim
    z = np.load(golden_dir / f"mpm_grad_{tag}.npz")
    const, model, st, ins = _case(z, tag.split("_")[-1])
    ins = [t.requires_grad_(True) for t in ins]
    outs = MPMDiffSim(model, reorder=False)(st, *ins)
    e = torch.from_numpy(z["in_enabled"] != 0).to(dev())
    W = [torch.from_numpy(z["W_" + k]).float().to(dev()) for k in ["x", "v", "C", "F"]]
    L = sum((w[e] * o[e]).sum() for w, o in zip(W, outs))
    grads = torch.autograd.grad(L, ins)
    worst = {}
    for name, g in zip(["x", "v", "C", "F", "stress"], grads):
        gd = g.double().cpu()
        for d, fd in zip(z["dir_" + name], z["fd_" + name]):
            d = torch.from_numpy(d)
            an = float((gd * d).sum())
            # fp32 error budget of an inner product: relative to |g|.|d| rather than to the (possibly cancelling) result
            budget = FD_REL * abs(fd) + FD_ABS * float(gd.abs().mul(d.abs()).sum())
            worst[name] = max(worst.get(name, 0.0), abs(an - fd) / budget)
    # |analytic - fd| over its budget (FD_REL |fd| + FD_ABS sum |g||d|), worst direction per input: logged, <= 1
    for name, wv in worst.items():
        parity(f"adjoint {tag} vs central differences of the reference forward", f"dL/d{name}: |an - fd| / budget", wv, 1.0)


def _stress(F, mu, lam):
    J = torch.linalg.det(F)
    I = torch.eye(3, dtype=F.dtype, device=F.device)
    return mu * (F @ F.transpose(1, 2) - I) + lam * torch.log(J)[:, None, None] * I


def test_inplace_forward_rollout_spans_and_extra_vs_reference_run(golden_dir):
    """render.py:304-310 order on the mirrored classes: from_torch(stress) -> MPMForwardSim (in place) ->
    statics_initializer.update(statics, step); MPMExtraSim at steps 3 and 9."""
    from neuma_amd.sim import MPMModelBuilder, MPMInitData, MPMStateInitializer, MPMStaticsInitializer, MPMForwardSim, MPMExtraSim
    z = np.load(golden_dir / "mpm_rollout.npz")
    cfg = dict(gravity=[0.0, -9.8, 0.0], bc="noslip", num_grids=int(z["G"]), dt=float(z["dt"]), bound=1, eps=6e-7)
    model = MPMModelBuilder().parse_cfg(cfg).finalize(dev(), requires_grad=False)
    sec = [int(s) for s in z["sections"]]
    x0, v0 = z["x0"], z["v0"]
    si, sti = MPMStateInitializer(model), MPMStaticsInitializer(model)
    off = 0
    for n, span in zip(sec, z["spans"]):
        g = MPMInitData(rho=float(z["rho"][off]), clip_bound=float(z["clip_bound"][off]), span=(int(span[0]), int(span[1])),
                        num_particles=n, vol=float(z["vol"][off]), pos=x0[off:off + n])
        g.set_ind_vel(v0[off:off + n])
        si.add_group(g); sti.add_group(g)
        off += n
    state, sections = si.finalize()
    statics = sti.finalize()
    assert np.array_equal(statics.enabled.cpu().numpy(), z["enabled0"])
    sim, extra = MPMForwardSim(model, reorder=False), MPMExtraSim(model, reorder=False)
    ne = z["xe0"].shape[0]
    st_e = model.statics(ne)
    st_e.enabled.fill_(1); st_e.clip_bound.fill_(0.1)
    state_e = model.state(ne)
    state_e.particle.x.copy_(torch.from_numpy(z["xe0"]).float())
    # measured (MI355X, round 4) x | v | C | F: step 1 5.3e-8 | 2.4e-7 | 4.9e-6 | 6.0e-8, step 5 1.2e-7 | 3.1e-7 | 1.1e-6 | 3.1e-7,
    # step 12 1.9e-7 | 4.1e-7 | 7.1e-7 | 5.6e-7 - at or below the reference's own fp32-vs-fp64 distance on this roll-out
    tol = {1: dict(x=2e-7, v=1e-6, C=1.5e-5, F=2e-7), 5: dict(x=4e-7, v=1.5e-6, C=6e-6, F=1.1e-6), 12: dict(x=6e-7, v=2e-6, C=3e-6, F=2e-6)}
    for step in range(1, 13):
        F = state.particle.F
        state.from_torch(stress=_stress(F.double(), float(z["mu"]), float(z["lam"])).float())
        if step in (3, 9):
            xe = extra(statics, state, st_e, state_e)
            parity(f"in-place roll-out, step {step}, passive extra set vs reference f64 run", "x_extra", np.abs(xe.cpu().double().numpy() - z[f"f64_xe_{step}"]).max(), XE_BOUND)
        x, v, C, Fn = sim(statics, state)
        sti.update(statics, step)
        if step in tol:
            assert np.array_equal(statics.enabled.cpu().numpy(), z[f"f64_enabled_{step}"])
            for name, got in zip(["x", "v", "C", "F"], (x, v, C, Fn)):
                ref = z[f"f64_{name}_{step}"]
                scale = max(1.0, np.abs(ref).max()) if name in ("v", "C") else 1.0
                err = np.abs(got.cpu().double().numpy() - ref).max()
                noise = np.abs(z[f"f32_{name}_{step}"].astype(np.float64) - ref).max() / scale if f"f32_{name}_{step}" in z else None
                parity(f"in-place roll-out, step {step}, vs reference f64 run", name, err / scale, tol[step][name], noise)


def test_svd_vs_reference_sign_rule(golden_dir):
    from neuma_amd.svd import SVD
    z = np.load(golden_dir / "svd_rule.npz")
    A = torch.from_numpy(z["A"]).float().to(dev())
    U, s, Vh = (t.double().cpu() for t in SVD()(A))
    rU, rs, rVh = (torch.from_numpy(z[f"numpy_f64_{k}"]) for k in ["U", "sigma", "Vh"])
    case = "SVD vs numpy f64 factors under the reference's sign rule"
    parity(case, "sigma", (s - rs).abs().max(), SVD_BOUND["sigma"])
    parity(case, "det U - 1", (torch.linalg.det(U) - 1).abs().max(), SVD_BOUND["det"])
    parity(case, "det Vh - 1", (torch.linalg.det(Vh) - 1).abs().max(), SVD_BOUND["det"])
    parity(case, "U diag(s) Vh - A", (U @ torch.diag_embed(s) @ Vh - torch.from_numpy(z["A"])).abs().max(), SVD_BOUND["recon"])
    distinct = ((rs[:, 0] - rs[:, 1]).abs() > 1e-2) & ((rs[:, 1] - rs[:, 2].abs()).abs() > 1e-2)
    assert int(distinct.sum()) > 60
    parity(case, "U Vh (rotation), distinct sigma", ((U @ Vh) - (rU @ rVh))[distinct].abs().max(), SVD_BOUND["rot"])
    for i in range(3):      # rank-one pieces u_i v_i^T are sign-invariant: same factors up to the joint column flips
        P = U[:, :, i, None] * Vh[:, None, i, :]
        rP = rU[:, :, i, None] * rVh[:, None, i, :]
        parity(case, f"u_{i} v_{i}^T, distinct sigma", (P - rP)[distinct].abs().max(), SVD_BOUND["rank1"])


def test_cov_deform_vs_reference_run(golden_dir):
    from neuma_amd.render import deform_cov_by_F
    z = np.load(golden_dir / "cov_deform.npz")
    out = deform_cov_by_F(torch.from_numpy(z["cov6"]).float().to(dev()), torch.from_numpy(z["F"]).float().to(dev()))
    ref = z["f64_out"]
    parity("deform_cov_by_F vs reference f64 run", "cov (rel)", np.abs(out.cpu().double().numpy() - ref).max() / np.abs(ref).max(), COV_BOUND)
