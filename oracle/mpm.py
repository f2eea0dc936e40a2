"""Oracle (TEST INFRASTRUCTURE ONLY) — MLS-MPM substep of the reference, restated in PyTorch.

Follows /root/reference/modules/nclaw/sim/mpm.py:
  p2g               321-371
  grid_op_freeslip  373-400
  grid_op_noslip    402-429
  g2p               432-498
  MPMModel.forward  279-297   (clear grid, p2g, grid_op, g2p)
  MPMModel.backward 299-319   (recompute p2g + grid_op, reverse g2p, grid_op, p2g)  -> here: torch
                               autograd over the same forward, which is what the Warp tape computes.
  MPMModelBuilder.parse_cfg 507-528 (dx = 1/G, inv_dx = G)

Parity PINNED (since round 2): the reference executes these only through warp-lang (absent here) and ships no vectors, so the
fixtures were made by EXECUTING the reference's own kernel bodies under a scalar stand-in for the ~20 wp.* names they use
(tests/golden/gen_mpm_golden.py + warp_scalar.py -> tests/golden/mpm_step_*.npz, mpm_grad_*.npz, mpm_rollout.npz, mpm_init.npz);
tests/test_oracle_pinned.py holds this file to them (fp64 run: 1e-11), tests/test_oracle_mpm.py adds invariants.

Semantic details kept from the reference:
  * base = int(x*inv_dx - 0.5) is a C cast (truncation toward zero), not floor  (mpm.py:336-339)
  * wp.mat33(a, b, c) builds COLUMNS, so w[axis, i] = i-th 1-D weight on `axis`   (mpm.py:350-355)
  * `m > 0` test before dividing by (m + eps)                                     (mpm.py:382-385)
  * noslip zeroes the whole velocity, tests run sequentially on the updated v     (mpm.py:416-427)
  * disabled particles are skipped in p2g and g2p                                 (mpm.py:331,443)
    (reference leaves their next-state stale; the oracle passes the current state through,
     detached, which is what the in-place MPMForwardSim yields.)
  * stencil nodes with any index >= G (reference UB when x > 1-1.5dx) are dropped in p2g and read
    as zero velocity in g2p (documented guard; the build does the same).
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import Tensor


@dataclass
class MPMConstant:
    """mpm.py:158-167"""
    num_grids: int
    dt: float
    bound: int
    gravity: Tuple[float, float, float]
    eps: float
    bc: str = "noslip"

    @property
    def dx(self) -> float:
        return 1.0 / self.num_grids

    @property
    def inv_dx(self) -> float:
        return float(self.num_grids)


def _stencil(const: MPMConstant, x: Tensor):
    """Shared head of p2g / g2p (mpm.py:335-355 / 446-466).

    Returns base (N,3) long, f (N,3), w (N,3,3) with w[:, axis, i]."""
    dt_ = x.dtype
    inv_dx = torch.tensor(const.inv_dx, dtype=dt_)
    p_x = x * inv_dx
    base = torch.trunc((p_x - 0.5).detach()).to(torch.long)  # C cast: toward zero; zero derivative
    f = p_x - base.to(dt_)
    wa = 1.5 - f
    wb = f - 1.0
    wc = f - 0.5
    w = torch.stack([wa * wa * 0.5, 0.75 - wb * wb, wc * wc * 0.5], dim=-1)  # (N, axis, i)
    return base, f, w


def _offsets(dtype):
    ii, jj, kk = torch.meshgrid(torch.arange(3), torch.arange(3), torch.arange(3), indexing="ij")
    off = torch.stack([ii.reshape(-1), jj.reshape(-1), kk.reshape(-1)], dim=-1)  # (27,3) i-major
    return off, off.to(dtype)


def p2g(const: MPMConstant, vol: Tensor, rho: Tensor, enabled: Tensor,
        x: Tensor, v: Tensor, C: Tensor, stress: Tensor) -> Tuple[Tensor, Tensor]:
    """mpm.py:321-371. Returns grid mv (G,G,G,3), m (G,G,G)."""
    G = const.num_grids
    dt_ = x.dtype
    base, f, w = _stencil(const, x)
    off_i, off_f = _offsets(dt_)
    p_mass = vol * rho                                                   # :334
    kappa = -const.dt * 4.0 * const.inv_dx * const.inv_dx
    affine = (kappa * vol)[:, None, None] * stress + p_mass[:, None, None] * C   # :357-358
    dpos = (off_f[None] - f[:, None, :]) * const.dx                      # (N,27,3) :365
    weight = (w[:, 0, off_i[:, 0]] * w[:, 1, off_i[:, 1]] * w[:, 2, off_i[:, 2]])  # (N,27) :366
    mom = p_mass[:, None, None] * v[:, None, :] + torch.einsum("nab,nkb->nka", affine, dpos)
    mv_c = weight[..., None] * mom                                       # :367
    m_c = weight * p_mass[:, None]                                       # :368
    node = base[:, None, :] + off_i[None]                                # (N,27,3)
    ok = (enabled != 0)[:, None] & (node >= 0).all(-1) & (node < G).all(-1)
    lin = (node[..., 0] * G + node[..., 1]) * G + node[..., 2]
    lin = torch.where(ok, lin, torch.zeros_like(lin))
    okf = ok.to(dt_)
    grid_mv = torch.zeros(G * G * G, 3, dtype=dt_).index_add(0, lin.reshape(-1), (mv_c * okf[..., None]).reshape(-1, 3))
    grid_m = torch.zeros(G * G * G, dtype=dt_).index_add(0, lin.reshape(-1), (m_c * okf).reshape(-1))
    return grid_mv.view(G, G, G, 3), grid_m.view(G, G, G)


def grid_op(const: MPMConstant, grid_mv: Tensor, grid_m: Tensor) -> Tensor:
    """mpm.py:373-400 (freeslip) / 402-429 (noslip). Returns grid v (G,G,G,3)."""
    G = const.num_grids
    dt_ = grid_mv.dtype
    g = torch.tensor(const.gravity, dtype=dt_) * const.dt
    has = grid_m > 0
    safe_m = torch.where(has, grid_m, torch.ones_like(grid_m))
    v = torch.where(has[..., None], grid_mv / (safe_m + const.eps)[..., None] + g, g.expand_as(grid_mv))
    idx = torch.arange(G)
    lo = idx < const.bound
    hi = idx >= G - const.bound
    shape = [(G, 1, 1), (1, G, 1), (1, 1, G)]
    if const.bc == "freeslip":
        comps = []
        for a in range(3):
            va = v[..., a]
            kill = (lo.view(shape[a]) & (va < 0)) | (hi.view(shape[a]) & (va > 0))
            comps.append(torch.where(kill, torch.zeros_like(va), va))
        return torch.stack(comps, dim=-1)
    if const.bc == "noslip":
        # sequential tests on the progressively updated v: once zeroed, later tests see 0 and do not fire
        alive = torch.ones_like(grid_m, dtype=torch.bool)
        for a in range(3):
            va = v[..., a]
            alive = alive & ~(lo.view(shape[a]) & (va < 0))
        for a in range(3):
            va = v[..., a]
            alive = alive & ~(hi.view(shape[a]) & (va > 0))
        return torch.where(alive[..., None], v, torch.zeros_like(v))
    raise ValueError("invalid boundary condition: {}".format(const.bc))  # mpm.py:550


def g2p(const: MPMConstant, clip_bound: Tensor, enabled: Tensor,
        x: Tensor, F: Tensor, grid_v: Tensor,
        v_old: Optional[Tensor] = None, C_old: Optional[Tensor] = None, fresh: bool = False):
    """mpm.py:432-498. Returns x', v', C', F'.  A disabled particle's row (mpm.py:443-444: the kernel returns at once) keeps
    what the next-state buffer held: fresh=False - the step is in place (MPMForwardSim, interface.py:126-135), the row keeps the
    particle's state; fresh=True - the buffer is a new model.state() (MPMDiffSim, interface.py:101-105; mpm.py:84-93): zeros,
    F = identity."""
    G = const.num_grids
    dt_ = x.dtype
    base, f, w = _stencil(const, x)
    off_i, off_f = _offsets(dt_)
    dpos = (off_f[None] - f[:, None, :]) * const.dx
    weight = (w[:, 0, off_i[:, 0]] * w[:, 1, off_i[:, 1]] * w[:, 2, off_i[:, 2]])
    node = base[:, None, :] + off_i[None]
    ok = (node >= 0).all(-1) & (node < G).all(-1)
    lin = (node[..., 0] * G + node[..., 1]) * G + node[..., 2]
    lin = torch.where(ok, lin, torch.zeros_like(lin))
    gv = grid_v.reshape(-1, 3)[lin] * ok.to(dt_)[..., None]               # (N,27,3)
    new_v = (weight[..., None] * gv).sum(1)                               # :478
    kap = 4.0 * const.inv_dx * const.inv_dx
    new_C = kap * torch.einsum("nk,nka,nkb->nab", weight, gv, dpos)       # :479  outer(v, dpos)
    I = torch.eye(3, dtype=dt_)
    new_F = (I + const.dt * new_C) @ F                                    # :489
    bnd = clip_bound[:, None] * const.dx
    new_x = x + const.dt * new_v
    # :491-497 wp.clamp — like torch.clamp, passes the gradient iff lo <= x <= hi
    new_x = torch.clamp(new_x, min=(0.0 + bnd).expand_as(new_x), max=(1.0 - bnd).expand_as(new_x))
    en = (enabled != 0)
    if not bool(en.all()) and fresh:
        new_x = torch.where(en[:, None], new_x, torch.zeros_like(new_x))
        new_v = torch.where(en[:, None], new_v, torch.zeros_like(new_v))
        new_C = torch.where(en[:, None, None], new_C, torch.zeros_like(new_C))
        new_F = torch.where(en[:, None, None], new_F, I.expand_as(new_F))
    elif not bool(en.all()):
        v_old = torch.zeros_like(new_v) if v_old is None else v_old
        C_old = torch.zeros_like(new_C) if C_old is None else C_old
        new_x = torch.where(en[:, None], new_x, x.detach())
        new_v = torch.where(en[:, None], new_v, v_old.detach())
        new_C = torch.where(en[:, None, None], new_C, C_old.detach())
        new_F = torch.where(en[:, None, None], new_F, F.detach())
    return new_x, new_v, new_C, new_F


def step(const: MPMConstant, vol, rho, clip_bound, enabled, x, v, C, F, stress, return_grid=False, fresh: bool = False):
    """MPMModel.forward mpm.py:279-297 (fresh: see g2p)."""
    grid_mv, grid_m = p2g(const, vol, rho, enabled, x, v, C, stress)
    grid_v = grid_op(const, grid_mv, grid_m)
    out = g2p(const, clip_bound, enabled, x, F, grid_v, v, C, fresh=fresh)
    if return_grid:
        return out, (grid_mv, grid_m, grid_v)
    return out


def touched_nodes(const: MPMConstant, x: Tensor, enabled: Optional[Tensor] = None) -> int:
    """Number T of distinct grid nodes inside any enabled particle's 3x3x3 stencil."""
    G = const.num_grids
    base, _, _ = _stencil(const, x)
    off_i, _ = _offsets(x.dtype)
    node = base[:, None, :] + off_i[None]
    ok = (node >= 0).all(-1) & (node < G).all(-1)
    if enabled is not None:
        ok = ok & (enabled != 0)[:, None]
    lin = (node[..., 0] * G + node[..., 1]) * G + node[..., 2]
    return int(torch.unique(lin[ok]).numel())


def sim_backward_nan_to_num(grads):
    """interface.py:65-74 — every gradient returned by the sim gets nan_to_num(0,0,0)."""
    return tuple(None if g is None else torch.nan_to_num(g, 0.0, 0.0, 0.0) for g in grads)
