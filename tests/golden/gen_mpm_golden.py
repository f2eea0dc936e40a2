#!/usr/bin/env python3
"""Generate golden vectors for the MPM substep, the SVD det/sign rule, deform_cov_by_F and the statics / state
initializers by EXECUTING THE REFERENCE'S OWN CODE.

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/gen_mpm_golden.py

warp-lang is absent, so `tests/golden/warp_scalar.py` (a scalar numpy stand-in for the few `wp.*` names the kernels
use) is registered as `warp`; then the reference's modules are imported unmodified and their code is run:

  modules/nclaw/sim/mpm.py     MPMModelBuilder.parse_cfg/finalize, MPMModel.forward / forward_extra (clear, p2g,
                               grid_op_{noslip,freeslip}, g2p: lines 260-498), MPMStatics.update_*, MPMStateInitializer,
                               MPMStaticsInitializer (554-776)
  modules/nclaw/sim/interface.py  MPMForwardSim, MPMExtraSim (126-147) - the in-place forward drivers
  modules/nclaw/warp/svd.py    SVDFunction.batch_svd (61-96) with numpy's SVD standing in for wp.svd3
  modules/d3gs/utils/simulation_utils.py  deform_cov_by_F (25-48)

Every kernel body runs once per thread id in Python, in fp64 (exact-arithmetic pin of the ALGORITHM) and in fp32 (the
reference's own precision, atomics applied in thread order).  Gradient KATs are central differences of the reference
forward in fp64 along random directions (Warp's generated adjoints are not in the tree).

Outputs (data only - inputs and expected outputs):
    tests/golden/mpm_step_<tag>.npz      one substep: inputs, grid mv/m/v, next state (fp64 + fp32 runs)
    tests/golden/mpm_grad_<tag>.npz      directional derivatives of a weighted sum of the next state
    tests/golden/mpm_rollout.npz         12 in-place MPMForwardSim steps with an analytic stress, span enabling, extra set
    tests/golden/mpm_init.npz            statics / state initializer outputs
    tests/golden/svd_rule.npz            batch_svd outputs (both raw-factor conventions)
    tests/golden/cov_deform.npz          deform_cov_by_F outputs
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
OUT = Path(os.environ.get("NEUMA_GOLDEN_OUT", HERE))      # (tests/test_golden_regen.py regenerates into a temporary directory)
sys.path.insert(0, str(HERE))

import warp_scalar as wps  # noqa: E402


def install():
    import types
    wps.install(sys.modules)
    oc = types.ModuleType("omegaconf")

    class DictConfig(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    oc.DictConfig = DictConfig
    oc.OmegaConf = object
    sys.modules["omegaconf"] = oc
    sys.path.insert(0, str(REF))
    return DictConfig


DictConfig = install()
import modules.nclaw.sim as rsim            # noqa: E402  the reference package (mpm.py + interface.py)
import modules.nclaw.warp.svd as rsvd       # noqa: E402
import importlib.util                        # noqa: E402

_spec = importlib.util.spec_from_file_location("ref_simulation_utils", REF / "modules/d3gs/utils/simulation_utils.py")
rcov = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(rcov)


# ----------------------------------------------------------------------------------------------------------------
def make_case(N, G, seed, dt=1e-3):
    """Inputs in fp64: interior cloud + particles within one cell of the low wall (truncation of int(), BC masks),
    near the high wall, floor contact; a disabled span; large stress / C so every term matters."""
    r = np.random.Generator(np.random.PCG64(seed))
    dx = 1.0 / G
    x = 0.3 + 0.4 * r.random((N, 3))
    k = max(N // 16, 2)
    x[:k] = 0.1 * dx + 0.9 * dx * r.random((k, 3))
    x[k:2 * k] = 1.0 - 1.6 * dx - 0.9 * dx * r.random((k, 3))
    x[2 * k:3 * k, 1] = 0.1 * dx + 0.5 * dx * r.random(k)
    v = r.standard_normal((N, 3))
    C = 2.0 * r.standard_normal((N, 3, 3))
    F = np.eye(3)[None] + 0.1 * r.standard_normal((N, 3, 3))
    S = 50.0 * r.standard_normal((N, 3, 3))
    vol = np.full(N, (dx / 2) ** 3) * (0.5 + r.random(N))
    rho = np.full(N, 1000.0) * (0.5 + r.random(N))
    clip = np.full(N, 0.1)
    clip[: N // 4] = 1.5                      # a wider clamp band so that the position clamp fires
    en = np.ones(N, np.int32)
    en[N // 2: N // 2 + max(N // 10, 1)] = 0
    return dict(x=x, v=v, C=C, F=F, stress=S, vol=vol, rho=rho, clip_bound=clip, enabled=en, G=G, dt=dt)


def build_model(G, dt, bc, gravity=(0.0, -9.8, 0.0), bound=1, eps=6e-7):
    cfg = DictConfig(num_grids=G, dt=dt, bound=bound, gravity=list(gravity), bc=bc, eps=eps)
    return rsim.MPMModelBuilder().parse_cfg(cfg).finalize("cpu", requires_grad=False)


def run_step(case, bc, ftype, _sentinel=True, **over):
    """One reference substep: MPMModel.forward (mpm.py:279-297) on fresh states. Returns dict of numpy arrays.
    _sentinel=False: the next state is left exactly as model.state() created it (what MPMDiffSim hands to the kernels,
    interface.py:101-105) - the rows of disabled particles then show what the reference's out-of-place sims return for them."""
    wps.set_float(ftype)
    wps.oob_atomics = 0
    c = dict(case)
    c.update(over)
    N = c["x"].shape[0]
    model = build_model(c["G"], c["dt"], bc)
    statics = model.statics(N)
    statics.vol.assign(c["vol"]); statics.rho.assign(c["rho"]); statics.clip_bound.assign(c["clip_bound"])
    statics.enabled.assign(c["enabled"])
    cur = model.state(N)
    nxt = model.state(N)
    for name in ["x", "v", "C", "F", "stress"]:
        getattr(cur.particle, name).assign(c[name])
    # next state pre-filled with a sentinel: disabled particles must be left untouched by g2p (mpm.py:443-444)
    for name, val in [("x", -7.0), ("v", -7.0), ("C", -7.0), ("F", -7.0)]:
        if _sentinel:
            getattr(nxt.particle, name).data[...] = val
    model.forward(statics, cur, nxt, None)
    assert wps.oob_atomics == 0, "fixture touches out-of-range nodes (reference UB)"
    p = nxt.particle
    return dict(mv=model.grid.mv.data.copy(), m=model.grid.m.data.copy(), gv=model.grid.v.data.copy(),
                x=p.x.data.copy(), v=p.v.data.copy(), C=p.C.data.copy(), F=p.F.data.copy())


def gen_steps():
    for N, G, seed in [(64, 16, 11), (2048, 32, 12)]:
        case = make_case(N, G, seed)
        for bc in ["noslip", "freeslip"]:
            tag = f"n{N}_g{G}_{bc}"
            out = {f"in_{k}": v for k, v in case.items()}
            r64 = run_step(case, bc, np.float64)
            r32 = run_step(case, bc, np.float32)
            for k, v in r64.items():
                out[f"f64_{k}"] = v
            for k, v in r32.items():
                out[f"f32_{k}"] = v
            rf = run_step(case, bc, np.float32, _sentinel=False)        # next state = a fresh model.state(): disabled rows
            for k in ["x", "v", "C", "F"]:
                out[f"fresh_{k}"] = rf[k]
            nz = int((r64["m"] > 0).sum())
            print(f"step {tag}: touched nodes {nz}/{G**3}, max|f32-f64| x {abs(r32['x'] - r64['x']).max():.2e} "
                  f"v {abs(r32['v'] - r64['v']).max():.2e} C {abs(r32['C'] - r64['C']).max():.2e} F {abs(r32['F'] - r64['F']).max():.2e}")
            np.savez_compressed(OUT / f"mpm_step_{tag}.npz", **out)


def gen_grads():
    """Directional central differences of L = <Wx,x'> + <Wv,v'> + <WC,C'> + <WF,F'> through the reference forward."""
    for N, G, seed in [(64, 16, 21), (384, 16, 22)]:
        case = make_case(N, G, seed)
        r = np.random.Generator(np.random.PCG64(seed + 100))
        W = {k: r.standard_normal(case[k].shape) for k in ["x", "v", "C", "F"]}
        en = case["enabled"] != 0

        def loss(bc, **over):
            o = run_step(case, bc, np.float64, **over)
            # disabled rows hold the sentinel - exclude them (their next state is not a function of the inputs)
            return sum(float((W[k][en] * o[k][en]).sum()) for k in W)

        for bc in ["noslip", "freeslip"]:
            out = {f"in_{k}": v for k, v in case.items()}
            out.update({f"W_{k}": v for k, v in W.items()})
            for name in ["x", "v", "C", "F", "stress"]:
                dirs, vals, vals2 = [], [], []
                for t in range(3):
                    d = r.standard_normal(case[name].shape)
                    if name == "x":
                        d *= 1e-2          # keep x +- h d far inside the same cell
                    h = 1e-6
                    fd = (loss(bc, **{name: case[name] + h * d}) - loss(bc, **{name: case[name] - h * d})) / (2 * h)
                    h2 = 4e-6
                    fd2 = (loss(bc, **{name: case[name] + h2 * d}) - loss(bc, **{name: case[name] - h2 * d})) / (2 * h2)
                    dirs.append(d); vals.append(fd); vals2.append(fd2)
                out[f"dir_{name}"] = np.stack(dirs)
                out[f"fd_{name}"] = np.array(vals)
                out[f"fd4_{name}"] = np.array(vals2)
                print(f"grad n{N} {bc} d/d{name}: {vals}  (h vs 4h rel diff "
                      f"{max(abs(a - b) / max(abs(a), 1e-30) for a, b in zip(vals, vals2)):.1e})")
            np.savez_compressed(OUT / f"mpm_grad_n{N}_g{G}_{bc}.npz", **out)


def analytic_stress(F, mu=400.0, lam=600.0):
    """The stand-in constitutive law of the roll-out fixture (stated here and in the test; NOT from the reference):
    Kirchhoff stress of a compressible neo-Hookean solid, tau = mu (F F^T - I) + lam log(J) I."""
    J = np.linalg.det(F)
    I = np.eye(3)[None]
    return mu * (F @ np.swapaxes(F, 1, 2) - I) + lam * np.log(J)[:, None, None] * I


def gen_rollout():
    """render.py:304-310 order: stress -> in-place MPMForwardSim -> statics_initializer.update(statics, step)."""
    for ftype, key in [(np.float64, "f64"), (np.float32, "f32")]:
        wps.set_float(ftype)
        G, dt = 16, 2e-3
        model = build_model(G, dt, "noslip")
        r = np.random.Generator(np.random.PCG64(31))
        # two bodies: a block resting just above the floor (contact within a few steps), and one that joins at step 4
        pa = np.stack(np.meshgrid(*[np.linspace(0.40, 0.60, 6)] * 3, indexing="ij"), -1).reshape(-1, 3)
        pa[:, 1] -= 0.31
        pb = 0.45 + 0.1 * r.random((40, 3))
        pb[:, 1] += 0.2
        ga = rsim.MPMInitData(rho=1000.0, clip_bound=0.1, span=(0, 1000), num_particles=pa.shape[0], vol=(0.04 ** 3), pos=pa,
                              lin_vel=np.array([0.3, -2.0, 0.1]), ang_vel=np.array([0.0, 0.0, 3.0]))
        gb = rsim.MPMInitData(rho=700.0, clip_bound=0.2, span=(4, 1000), num_particles=pb.shape[0], vol=2e-5, pos=pb,
                              lin_vel=np.array([0.0, -1.0, 0.0]))
        si = rsim.MPMStateInitializer(model); si.add_group(ga); si.add_group(gb)
        state, sections = si.finalize()
        sti = rsim.MPMStaticsInitializer(model); sti.add_group(ga); sti.add_group(gb)
        statics = sti.finalize()
        sim = rsim.MPMForwardSim(model)
        # extra passive set advected through the grid of the main set (MPMExtraSim, interface.py:138-147)
        xe = 0.42 + 0.16 * r.random((36, 3)); xe[:, 1] -= 0.28
        xe[24:32] = 0.75 + 0.1 * r.random((8, 3))           # in empty space: reads v = g dt of untouched nodes (mpm.py:413-414)
        xe[32:36] = 0.7 + 0.2 * r.random((4, 3)); xe[32:36, 1] = 0.3 / G + 0.5 / G * r.random(4)   # empty space at the floor: BC-masked
        st_e = model.statics(36)
        st_e.enabled.assign(np.ones(36, np.int32)); st_e.clip_bound.assign(np.full(36, 0.1))
        state_e = model.state(36)
        state_e.particle.x.assign(xe)
        extra = rsim.MPMExtraSim(model)
        out = dict(G=G, dt=dt, sections=np.array(sections), x0=state.particle.x.data.copy(), v0=state.particle.v.data.copy(),
                   vol=statics.vol.data.copy(), rho=statics.rho.data.copy(), clip_bound=statics.clip_bound.data.copy(),
                   enabled0=statics.enabled.data.copy(), xe0=xe, mu=400.0, lam=600.0,
                   spans=np.array([[0, 1000], [4, 1000]]))
        for step in range(1, 13):
            F = state.particle.F.data.astype(np.float64)
            stress = analytic_stress(F).astype(ftype)
            state.from_torch(stress=torch.from_numpy(stress))
            if step in (3, 9):
                x_e = extra(statics, state, st_e, state_e)      # before the step: same grid as the step would build
                out[f"{key}_xe_{step}"] = x_e.numpy().copy()
            x, v, C, Fn = sim(statics, state)
            sti.update(statics, step)
            if step in (1, 5, 12):
                out[f"{key}_x_{step}"] = x.numpy().copy(); out[f"{key}_v_{step}"] = v.numpy().copy()
                out[f"{key}_C_{step}"] = C.numpy().copy(); out[f"{key}_F_{step}"] = Fn.numpy().copy()
                out[f"{key}_enabled_{step}"] = statics.enabled.data.copy()
        if key == "f64":
            res = out
        else:
            res.update({k: v for k, v in out.items() if k.startswith("f32_")})
        print(f"rollout {key}: min y after 12 steps {x.numpy()[:, 1].min():.4f}, |F-I| max {abs(Fn.numpy() - np.eye(3)).max():.3f}")
    np.savez_compressed(OUT / "mpm_rollout.npz", **res)


def gen_init():
    """MPMStaticsInitializer / MPMStateInitializer (mpm.py:698-776), MPMInitData.alignment (576-594)."""
    wps.set_float(np.float32)
    model = build_model(8, 1e-3, "noslip")
    r = np.random.Generator(np.random.PCG64(41))
    groups, out = [], {}
    specs = [dict(n=5, rho=1000.0, clip=0.1, span=(0, 5), vol=1e-5, lin=[1.0, 2.0, 3.0], ang=[0.5, -1.0, 2.0]),
             dict(n=7, rho=250.0, clip=0.3, span=(3, 1000), vol=3e-5, lin=[0.0, 0.0, 0.0], ang=[0.0, 0.0, 0.0]),
             dict(n=4, rho=10.0, clip=0.2, span=(0, 0), vol=7e-5, lin=[0.0, 0.0, 0.0], ang=[0.0, 0.0, 0.0])]
    for i, s in enumerate(specs):
        pos = 0.2 + 0.6 * r.random((s["n"], 3))
        g = rsim.MPMInitData(rho=s["rho"], clip_bound=s["clip"], span=s["span"], num_particles=s["n"], vol=s["vol"], pos=pos,
                             lin_vel=np.array(s["lin"]), ang_vel=np.array(s["ang"]))
        if i == 1:
            g.set_ind_vel(r.standard_normal((s["n"], 3)))
            out["ind_vel_1"] = g.ind_vel
        groups.append(g)
        out[f"pos_{i}"] = pos
        out[f"spec_{i}"] = np.array([s["rho"], s["clip"], s["span"][0], s["span"][1], s["vol"], *s["lin"], *s["ang"]])
    si = rsim.MPMStateInitializer(model)
    sti = rsim.MPMStaticsInitializer(model)
    for g in groups:
        si.add_group(g); sti.add_group(g)
    state, sections = si.finalize()
    statics = sti.finalize()
    out.update(sections=np.array(sections), x=state.particle.x.data.copy(), v=state.particle.v.data.copy(),
               C=state.particle.C.data.copy(), F=state.particle.F.data.copy(), vol=statics.vol.data.copy(),
               rho=statics.rho.data.copy(), clip_bound=statics.clip_bound.data.copy())
    for step in [0, 2, 3, 4, 5, 999, 1000]:
        sti.update(statics, step)
        out[f"enabled_{step}"] = statics.enabled.data.copy()
    s, t = rsim.MPMInitData.alignment(np.array([-1.0, -2.0, 0.0]), np.array([1.0, 2.0, 4.0]),
                                      np.array([0.3, 0.3, 0.3]), np.array([0.7, 0.6, 0.5]))
    out["align_scale"], out["align_trans"] = s, t
    c = model.constant
    out["constant"] = np.array([c.num_grids, c.dt, c.bound, c.dx, c.inv_dx, c.eps, *c.gravity.a])
    np.savez_compressed(OUT / "mpm_init.npz", **out)
    print("init: sections", sections)


def gen_svd():
    """batch_svd (svd.py:61-96).  wp.svd3 itself is absent; numpy's SVD provides the raw factors in two conventions."""
    r = np.random.Generator(np.random.PCG64(51))
    n = 96
    A = np.eye(3)[None] + 0.3 * r.standard_normal((n, 3, 3))
    A[-16:, :, 1] *= -1.0                         # reflected: det < 0
    A[0] = np.eye(3)                              # sigma = (1,1,1)
    A[1] = np.diag([2.0, 1.0, 0.5])
    A[2] = np.diag([1.0, 1.0, -1.0])
    out = dict(A=A)
    for mode in ["numpy", "rot"]:
        for ftype, key in [(np.float64, "f64"), (np.float32, "f32")]:
            wps.set_float(ftype)
            wps.SVD3_MODE = mode
            a = wps.array(A.astype(ftype), dtype=wps.mat33)
            U = wps.zeros(n, dtype=wps.mat33); s = wps.zeros(n, dtype=wps.vec3); Vh = wps.zeros(n, dtype=wps.mat33)
            wps.launch(rsvd.SVDFunction.batch_svd, dim=n, inputs=[a, U, s, Vh])
            out[f"{mode}_{key}_U"], out[f"{mode}_{key}_sigma"], out[f"{mode}_{key}_Vh"] = U.data.copy(), s.data.copy(), Vh.data.copy()
    wps.SVD3_MODE = "numpy"
    rec = np.einsum("nij,nj,njk->nik", out["numpy_f64_U"], out["numpy_f64_sigma"], out["numpy_f64_Vh"])
    print("svd rule: recon err", abs(rec - A).max(), " det U min", np.linalg.det(out["numpy_f64_U"]).min(),
          " det Vh min", np.linalg.det(out["numpy_f64_Vh"]).min(), " negative sigma2:", int((out["numpy_f64_sigma"][:, 2] < 0).sum()),
          " same in both conventions:", abs(out["numpy_f64_sigma"] - out["rot_f64_sigma"]).max())
    np.savez_compressed(OUT / "svd_rule.npz", **out)


def gen_cov():
    r = np.random.Generator(np.random.PCG64(61))
    K = 256
    L = 0.05 * r.standard_normal((K, 3, 3))
    cov = L @ np.swapaxes(L, 1, 2)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2], cov[:, 2, 2]], -1)
    F = np.eye(3)[None] + 0.4 * r.standard_normal((K, 3, 3))
    out = dict(cov6=cov6, F=F)
    for ftype, key in [(np.float64, "f64"), (np.float32, "f32")]:
        wps.set_float(ftype)
        a = wps.array(cov6.astype(ftype).reshape(-1).copy(), dtype=float)
        f = wps.array(F.astype(ftype), dtype=wps.mat33)
        o = wps.array(np.zeros(K * 6, ftype), dtype=float)
        wps.launch(rcov.deform_cov_by_F, dim=K, inputs=[a, f, o])
        out[f"{key}_out"] = o.data.reshape(K, 6).copy()
    print("cov deform: max |f32 - f64|", abs(out["f32_out"] - out["f64_out"]).max())
    np.savez_compressed(OUT / "cov_deform.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["steps", "grads", "rollout", "init", "svd", "cov"]
    for w in which:
        globals()["gen_" + w]()
