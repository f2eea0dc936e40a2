"""Invariant checks that pin the oracle's MPM restatement (no runnable reference exists: SURVEY §8c)."""
import numpy as np
import pytest
import torch

from oracle import mpm as om


def _setup(N=512, G=16, seed=0, bc="noslip", near_wall=True, dtype=torch.float64):
    g = torch.Generator().manual_seed(seed)
    const = om.MPMConstant(num_grids=G, dt=1e-3, bound=1, gravity=(0.0, -9.8, 0.0), eps=6e-7, bc=bc)
    x = 0.3 + 0.4 * torch.rand(N, 3, generator=g, dtype=dtype)
    if near_wall:
        x[:32] = 0.02 * torch.rand(32, 3, generator=g, dtype=dtype) + 0.1 * const.dx   # within 1/2 cell of the low wall
        x[32:64] = 1.0 - 1.6 * const.dx - 0.02 * torch.rand(32, 3, generator=g, dtype=dtype)
    v = torch.randn(N, 3, generator=g, dtype=dtype)
    C = torch.randn(N, 3, 3, generator=g, dtype=dtype)
    F = torch.eye(3, dtype=dtype)[None] + 0.1 * torch.randn(N, 3, 3, generator=g, dtype=dtype)
    S = torch.randn(N, 3, 3, generator=g, dtype=dtype)
    vol = torch.full((N,), (const.dx / 2) ** 3, dtype=dtype)
    rho = torch.full((N,), 1000.0, dtype=dtype)
    clip = torch.full((N,), 0.1, dtype=dtype)
    en = torch.ones(N, dtype=torch.int32)
    return const, vol, rho, clip, en, x, v, C, F, S


def test_truncation_not_floor():
    const, vol, rho, clip, en, x, *_ = _setup()
    base, f, w = om._stencil(const, x)
    assert (base >= 0).all()
    # particle at 0.1 dx: px - 0.5 = -0.4 -> int() = 0 (floor would give -1)
    assert (base[:32] == 0).all() and (f[:32] < 0.5).any()
    assert torch.allclose(w.sum(-1), torch.ones_like(w.sum(-1)))


def test_mass_momentum_conservation():
    const, vol, rho, clip, en, x, v, C, F, S = _setup(near_wall=False)
    mv, m = om.p2g(const, vol, rho, en, x, v, torch.zeros_like(C), torch.zeros_like(S))
    pm = vol * rho
    assert abs(m.sum() - pm.sum()) < 1e-12 * pm.sum()
    assert torch.allclose(mv.sum((0, 1, 2)), (pm[:, None] * v).sum(0), rtol=1e-12, atol=1e-12)


def test_affine_field_reproduced():
    """grid v(x_i) = a + A x_i  =>  g2p gives v' = a + A x_p and C' = A (pins w, dpos, 4 inv_dx^2, outer order)."""
    const, vol, rho, clip, en, x, v, C, F, S = _setup(near_wall=False)
    G = const.num_grids
    A = torch.tensor([[0.1, -0.3, 0.2], [0.5, 0.4, -0.1], [0.0, 0.7, -0.6]], dtype=torch.float64)
    a = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64)
    idx = torch.arange(G, dtype=torch.float64) * const.dx
    X = torch.stack(torch.meshgrid(idx, idx, idx, indexing="ij"), -1)
    gv = a + X @ A.T
    const0 = om.MPMConstant(G, 1e-3, 0, (0.0, 0.0, 0.0), 0.0)
    nx, nv, nC, nF = om.g2p(const0, clip, en, x, F, gv)
    assert torch.allclose(nv, a + x @ A.T, atol=1e-12)
    assert torch.allclose(nC, A.expand_as(nC), atol=1e-9)
    assert torch.allclose(nF, (torch.eye(3, dtype=torch.float64) + 1e-3 * nC) @ F, atol=1e-14)


@pytest.mark.parametrize("bc", ["noslip", "freeslip"])
def test_boundary_conditions(bc):
    const, *_ = _setup(bc=bc)
    G = const.num_grids
    mv = torch.randn(G, G, G, 3, dtype=torch.float64)
    m = torch.rand(G, G, G, dtype=torch.float64)
    m[::3] = 0.0
    v = om.grid_op(const, mv, m)
    gdt = torch.tensor(const.gravity, dtype=torch.float64) * const.dt
    raw = torch.where((m > 0)[..., None], mv / (m + const.eps)[..., None] + gdt, gdt.expand_as(mv))
    # interior untouched
    assert torch.equal(v[1:-1, 1:-1, 1:-1], raw[1:-1, 1:-1, 1:-1])
    # wall x=0: outward (negative) x velocity
    out = raw[0, 1:-1, 1:-1, 0] < 0
    if bc == "noslip":
        assert (v[0, 1:-1, 1:-1][out] == 0).all()
    else:
        assert (v[0, 1:-1, 1:-1, 0][out] == 0).all()
        assert torch.equal(v[0, 1:-1, 1:-1, 1][out], raw[0, 1:-1, 1:-1, 1][out])
    with pytest.raises(ValueError):
        om.grid_op(om.MPMConstant(G, 1e-3, 1, (0, 0, 0), 0.0, bc="sticky"), mv, m)


def test_gradients_finite_difference():
    const, vol, rho, clip, en, x, v, C, F, S = _setup(N=96, G=8, near_wall=False)
    ins = [t.clone().requires_grad_(True) for t in (x, v, C, F, S)]
    torch.manual_seed(5)
    outs = om.step(const, vol, rho, clip, en, *ins)
    gws = [torch.randn_like(o) for o in outs]
    loss = sum((o * g).sum() for o, g in zip(outs, gws))
    grads = torch.autograd.grad(loss, ins)

    def L(args):
        o = om.step(const, vol, rho, clip, en, *args)
        return sum((oo * g).sum() for oo, g in zip(o, gws)).item()

    h = 1e-6
    for ti in range(5):
        flat = ins[ti].detach().reshape(-1)
        for j in [0, 7, 31]:
            if ti == 0:
                continue  # x: piecewise (base cast) — covered by the directional test below
            d = torch.zeros_like(flat)
            d[j] = h
            args_p = [t.detach().clone() for t in ins]
            args_m = [t.detach().clone() for t in ins]
            args_p[ti] = (flat + d).reshape(ins[ti].shape)
            args_m[ti] = (flat - d).reshape(ins[ti].shape)
            fd = (L(args_p) - L(args_m)) / (2 * h)
            assert abs(fd - grads[ti].reshape(-1)[j].item()) <= 1e-5 * max(1.0, abs(fd))
    # x: move particles by an amount that keeps every base cell fixed
    d = 1e-7 * torch.randn_like(x)
    args_p = [t.detach().clone() for t in ins]
    args_m = [t.detach().clone() for t in ins]
    args_p[0] = x + d
    args_m[0] = x - d
    assert torch.equal(om._stencil(const, args_p[0])[0], om._stencil(const, args_m[0])[0])
    fd = (L(args_p) - L(args_m)) / 2
    an = (grads[0] * d).sum().item()
    assert abs(fd - an) <= 1e-5 * max(abs(fd), 1e-12)


def test_disabled_particles_skipped():
    const, vol, rho, clip, en, x, v, C, F, S = _setup(N=128, near_wall=False)
    en = en.clone()
    en[:40] = 0
    mv, m = om.p2g(const, vol, rho, en, x, v, C, S)
    mv2, m2 = om.p2g(const, vol[40:], rho[40:], en[40:], x[40:], v[40:], C[40:], S[40:])
    assert torch.allclose(m, m2) and torch.allclose(mv, mv2)
    nx, nv, nC, nF = om.step(const, vol, rho, clip, en, x, v, C, F, S)
    assert torch.equal(nx[:40], x[:40]) and torch.equal(nF[:40], F[:40])


def test_touched_nodes_count():
    const, vol, rho, clip, en, x, *_ = _setup(N=1, near_wall=False)
    assert om.touched_nodes(const, x) == 27
