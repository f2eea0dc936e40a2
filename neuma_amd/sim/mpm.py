"""MLS-MPM model / state / statics / builders on top of libneuma_hip.

Mirrors the public surface of /root/reference/modules/nclaw/sim/mpm.py:
  MPMStatics 14-72, MPMParticleData 75-128, MPMConstant 158-167, MPMState 170-242, MPMModel 245-319,
  MPMModelBuilder 501-551, MPMInitData 554-692, MPMStateInitializer 695-735, MPMStaticsInitializer 738-776.
The Warp structs become plain holders of torch tensors; the kernels (p2g / grid_op / g2p and their
adjoints) live in csrc/nm_mpm.hip and are reached through nm_mpm_forward / nm_mpm_backward.
"""
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Optional, Sequence, Union
import ctypes as C

import numpy as np
import torch
from torch import Tensor

from .abstract import State, Model, ModelBuilder, StateInitializer, StaticsInitializer
from .. import _lib as L


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    if hasattr(cfg, "get"):
        try:
            return cfg.get(key, default)
        except TypeError:
            pass
    return getattr(cfg, key, default)


def _cfg_req(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class MPMStatics(object):
    """mpm.py:14-72 — per-particle constants (float32 / int32 tensors)."""

    vol: Tensor
    rho: Tensor
    clip_bound: Tensor
    enabled: Tensor

    def init(self, shape: Union[Sequence[int], int], device=None) -> None:
        self.vol = torch.zeros(shape, dtype=torch.float32, device=device)
        self.rho = torch.zeros(shape, dtype=torch.float32, device=device)
        self.clip_bound = torch.zeros(shape, dtype=torch.float32, device=device)
        self.enabled = torch.zeros(shape, dtype=torch.int32, device=device)

    @staticmethod
    def _fill(t: Tensor, sections, values):
        offset = 0
        for section, value in zip(sections, values):
            if isinstance(value, (np.ndarray, Tensor)) and np.ndim(value) > 0:
                t[offset:offset + section] = torch.as_tensor(value, dtype=t.dtype, device=t.device)[offset:offset + section]
            else:
                t[offset:offset + section] = value
            offset += section

    def update_vol(self, sections, vols) -> None:
        self._fill(self.vol, sections, vols)

    def update_rho(self, sections, rhos) -> None:
        self._fill(self.rho, sections, rhos)

    def update_clip_bound(self, sections, clip_bounds) -> None:
        self._fill(self.clip_bound, sections, clip_bounds)

    def update_enabled(self, sections, spans, step: int = 0) -> None:
        # mpm.py:67-72: enabled = span[0] <= step < span[1]
        self._fill(self.enabled, sections, [1 if (span[0] <= step < span[1]) else 0 for span in spans])

    def c_struct(self) -> L.nm_statics:
        L.same_device(self.vol, self.rho, self.clip_bound, self.enabled)
        return L.nm_statics(L.ptr(self.vol, torch.float32), L.ptr(self.rho, torch.float32), L.ptr(self.clip_bound, torch.float32),
                            L.ptr(self.enabled, torch.int32))


class MPMParticleData(object):
    """mpm.py:75-128 — x,v (N,3); C,F,stress (N,3,3); `.grad` twins filled during backward."""

    def init(self, shape, device=None, requires_grad: bool = False) -> None:
        n = int(shape)
        self.requires_grad = requires_grad
        self.x = torch.zeros(n, 3, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, 3, dtype=torch.float32, device=device)
        self.C = torch.zeros(n, 3, 3, dtype=torch.float32, device=device)
        self.F = torch.eye(3, dtype=torch.float32, device=device).repeat(n, 1, 1)   # init_F, mpm.py:104-116
        self.stress = torch.zeros(n, 3, 3, dtype=torch.float32, device=device)
        self.x_grad = self.v_grad = self.C_grad = self.F_grad = self.stress_grad = None

    def clear(self) -> None:
        self.x.zero_(); self.v.zero_(); self.C.zero_(); self.stress.zero_()
        self.F.copy_(torch.eye(3, dtype=torch.float32, device=self.F.device).expand_as(self.F))

    def zero_grad(self) -> None:
        for g in (self.x_grad, self.v_grad, self.C_grad, self.F_grad, self.stress_grad):
            if g is not None:
                g.zero_()

    def c_struct(self) -> L.nm_particles:
        L.same_device(self.x, self.v, self.C, self.F, self.stress)
        f32 = torch.float32
        return L.nm_particles(L.ptr(self.x, f32), L.ptr(self.v, f32), L.ptr(self.C, f32), L.ptr(self.F, f32), L.ptr(self.stress, f32))

    def c_struct_grad(self) -> L.nm_particles:
        return L.nm_particles(L.ptr(self.x_grad), L.ptr(self.v_grad), L.ptr(self.C_grad), L.ptr(self.F_grad),
                              L.ptr(self.stress_grad))


@dataclass
class MPMConstant(object):
    """mpm.py:158-167"""
    num_grids: int = None
    dt: float = None
    bound: int = None
    gravity: Any = None
    dx: float = None
    inv_dx: float = None
    eps: float = None


class MPMState(State):
    """mpm.py:170-242"""

    def __init__(self, shape: int, device=None, requires_grad: bool = False) -> None:
        super().__init__(shape, device, requires_grad)
        particle = MPMParticleData()
        particle.init(shape, self.device, requires_grad)
        self.particle = particle

    def zero_grad(self) -> None:
        self.particle.zero_grad()

    def clear(self) -> None:
        self.particle.clear()

    def to_torch(self):
        p = self.particle
        return p.x, p.v, p.C, p.F, p.stress

    def to_torch_grad(self):
        p = self.particle
        return p.x_grad, p.v_grad, p.C_grad, p.F_grad, p.stress_grad

    @staticmethod
    def _prep(t: Tensor) -> Tensor:
        t = t.detach()
        if t.dtype != torch.float32:
            t = t.float()
        return t.contiguous()     # zero-copy alias when already contiguous fp32 (mpm.py:206-223)

    def from_torch(self, x=None, v=None, C=None, F=None, stress=None) -> None:
        p = self.particle
        if x is not None:
            p.x = self._prep(x)
        if v is not None:
            p.v = self._prep(v)
        if C is not None:
            p.C = self._prep(C)
        if F is not None:
            p.F = self._prep(F)
        if stress is not None:
            p.stress = self._prep(stress)

    def from_torch_grad(self, grad_x=None, grad_v=None, grad_C=None, grad_F=None, grad_stress=None) -> None:
        p = self.particle
        if grad_x is not None:
            p.x_grad = self._prep(grad_x)
        if grad_v is not None:
            p.v_grad = self._prep(grad_v)
        if grad_C is not None:
            p.C_grad = self._prep(grad_C)
        if grad_F is not None:
            p.F_grad = self._prep(grad_F)
        if grad_stress is not None:
            p.stress_grad = self._prep(grad_stress)


class GridTape(object):
    """What the `tape` positional of MPMModel.forward / backward carries here: one grid cache record (the substep's
    touched grid blocks, nm_mpm_forward_ex) instead of Warp's operation tape, so that the backward pass restores the
    grid instead of re-running p2g (mpm.py:312-315)."""

    def __init__(self, cap_blocks: int, device) -> None:
        self.cap = int(cap_blocks)
        self.buf = torch.empty(int(L.lib().nm_mpm_gridcache_bytes(self.cap)), dtype=torch.uint8, device=device)


class MPMModel(Model):
    """mpm.py:245-319.  Owns the (sparse-blocked) grid through an nm_mpm handle."""

    ConstantType = MPMConstant
    StaticsType = MPMStatics
    StateType = MPMState

    def __init__(self, constant: MPMConstant, device=None, requires_grad: bool = False, bc: str = "noslip") -> None:
        super().__init__(constant, device)
        self.requires_grad = requires_grad
        self.bc = bc
        self._handle = None
        # grid cache of the per-operator differentiable path: "auto" sizes the per-substep record from the first substep
        # (x1.5 + 64 blocks, one host sync); an int fixes the capacity; 0 / None = recompute like the reference
        self.grid_cache = "auto"
        self._cache_blocks = None
        self.exchange = None        # set by shard(): this model steps one rank's share of the particles

    def shard(self, group=None, cap=None, cap_shared=None, cap_dil=None, cap_frame=None):
        """Make this model one rank of a particle-sharded simulation (sim/shard.py): every forward / backward sums the
        grid blocks it shares with other ranks over `group`.  Capacities default to 1.5x what the first substep needs
        (cap, cap_shared: per-substep lists and grid cache records; cap_dil, cap_frame: the fused roll-out's frame-level
        neighbourhood and exchange lists)."""
        from .shard import GridExchange
        self.exchange = GridExchange(self, group, cap, cap_shared, cap_dil, cap_frame)
        return self.exchange

    def new_tape(self):
        """A fresh GridTape for one differentiable substep (None while the capacity is not known yet / cache disabled)."""
        if self.exchange is not None:
            from .shard import ShardTape
            return ShardTape(self.exchange.generation)
        if self.grid_cache in (0, None, False):
            return None
        if self.grid_cache != "auto":
            return GridTape(int(self.grid_cache), self.device)
        if self._cache_blocks is None:
            return None
        return GridTape(self._cache_blocks, self.device)

    def _size_cache(self) -> None:
        if self.exchange is None and self.grid_cache == "auto" and self._cache_blocks is None:
            blocks, _ = self.grid_stats()
            self._cache_blocks = int(1.5 * blocks) + 64

    # -- handle management
    def handle(self):
        if self._handle is None:
            if self.device.type != "cuda":
                raise L.NeumaHipError("MPMModel needs a GPU device: neuma_amd has no CPU path")
            c = self.constant
            g = [float(v) for v in np.asarray(c.gravity, dtype=np.float32).reshape(3)]
            cfg = L.nm_mpm_cfg(int(c.num_grids), float(c.dt), int(c.bound), (C.c_float * 3)(*g), float(c.eps),
                               {"noslip": 0, "freeslip": 1}[self.bc])
            out = C.c_void_p()
            with torch.cuda.device(self.device):
                L.check(L.lib().nm_mpm_create(C.byref(cfg), C.byref(out)), "nm_mpm_create")
            self._handle = out
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().nm_mpm_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _stream(self):
        return L.stream_ptr(self.device)

    # -- operators
    def forward(self, statics: MPMStatics, state_curr: MPMState, state_next: MPMState, tape=None) -> None:
        """mpm.py:279-297.  `tape`: None (nothing is recorded; the backward pass recomputes the grid exactly like
        mpm.py:312-315) or a GridTape from new_tape() (the substep's touched grid blocks are saved into it)."""
        if self.exchange is not None:
            return self.exchange.forward(statics, state_curr, state_next, tape)
        n = state_curr.particle.x.shape[0]
        st = statics.c_struct()
        cur = state_curr.particle.c_struct()
        nxt = state_next.particle.c_struct()
        if isinstance(tape, GridTape):
            L.check(L.lib().nm_mpm_forward_ex(self.handle(), n, C.byref(st), C.byref(cur), C.byref(nxt), L.ptr(tape.buf), tape.cap,
                                              self._stream()), "nm_mpm_forward_ex")
        else:
            L.check(L.lib().nm_mpm_forward(self.handle(), n, C.byref(st), C.byref(cur), C.byref(nxt), self._stream()),
                    "nm_mpm_forward")

    def backward(self, statics: MPMStatics, state_curr: MPMState, state_next: MPMState, tape=None) -> None:
        """mpm.py:299-319.  Reads state_next.particle.*_grad, writes state_curr.particle.*_grad."""
        pc, pn = state_curr.particle, state_next.particle
        n = pc.x.shape[0]
        dev = pc.x.device
        pc.x_grad = torch.empty(n, 3, dtype=torch.float32, device=dev)
        pc.v_grad = torch.empty(n, 3, dtype=torch.float32, device=dev)
        pc.C_grad = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
        pc.F_grad = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
        pc.stress_grad = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
        for name, shape in (("x_grad", (n, 3)), ("v_grad", (n, 3)), ("C_grad", (n, 3, 3)), ("F_grad", (n, 3, 3))):
            if getattr(pn, name) is None:
                setattr(pn, name, torch.zeros(shape, dtype=torch.float32, device=dev))
        st = statics.c_struct()
        cur, nxt = pc.c_struct(), pn.c_struct()
        gn = L.nm_particles(L.ptr(pn.x_grad), L.ptr(pn.v_grad), L.ptr(pn.C_grad), L.ptr(pn.F_grad), None)
        gc = pc.c_struct_grad()
        if self.exchange is not None:
            return self.exchange.backward(statics, state_curr, state_next, gn, gc, tape)
        if isinstance(tape, GridTape):
            L.check(L.lib().nm_mpm_backward_ex(self.handle(), n, C.byref(st), C.byref(cur), C.byref(nxt), C.byref(gn), C.byref(gc),
                                               L.ptr(tape.buf), tape.cap, self._stream()), "nm_mpm_backward_ex")
        else:
            L.check(L.lib().nm_mpm_backward(self.handle(), n, C.byref(st), C.byref(cur), C.byref(nxt), C.byref(gn),
                                            C.byref(gc), self._stream()), "nm_mpm_backward")

    def forward_extra(self, statics, state, statics_extra, state_extra) -> None:
        """mpm.py:260-277."""
        if self.exchange is not None:
            raise L.NeumaHipError("forward_extra is not available on a sharded model (gather the particles to one rank)")
        n, ne = state.particle.x.shape[0], state_extra.particle.x.shape[0]
        st, ste = statics.c_struct(), statics_extra.c_struct()
        cur, ext = state.particle.c_struct(), state_extra.particle.c_struct()
        L.check(L.lib().nm_mpm_forward_extra(self.handle(), n, C.byref(st), C.byref(cur), ne, C.byref(ste), C.byref(ext),
                                             self._stream()), "nm_mpm_forward_extra")

    # -- introspection (tests / roofline accounting)
    def grid_stats(self):
        a, b = C.c_int32(0), C.c_int32(0)
        L.check(L.lib().nm_mpm_grid_stats(self.handle(), C.byref(a), C.byref(b), self._stream()), "nm_mpm_grid_stats")
        return int(a.value), int(b.value)

    def grid_export(self):
        G = int(self.constant.num_grids)
        mv = torch.empty(G, G, G, 3, dtype=torch.float32, device=self.device)
        m = torch.empty(G, G, G, dtype=torch.float32, device=self.device)
        v = torch.empty(G, G, G, 3, dtype=torch.float32, device=self.device)
        L.check(L.lib().nm_mpm_grid_export(self.handle(), L.ptr(mv), L.ptr(m), L.ptr(v), self._stream()), "nm_mpm_grid_export")
        return mv, m, v


class MPMModelBuilder(ModelBuilder):
    """mpm.py:501-551"""

    StateType = MPMState
    ConstantType = MPMConstant
    ModelType = MPMModel

    def __init__(self) -> None:
        super().__init__()
        self.reserve("bc")

    def parse_cfg(self, cfg) -> 'MPMModelBuilder':
        num_grids = int(_cfg_req(cfg, "num_grids"))
        self.config['num_grids'] = num_grids
        self.config['dt'] = float(_cfg_req(cfg, "dt"))
        self.config['bound'] = int(_cfg_req(cfg, "bound"))
        self.config['gravity'] = np.array(list(_cfg_req(cfg, "gravity")), dtype=np.float32)
        self.config['dx'] = 1 / num_grids
        self.config['inv_dx'] = float(num_grids)
        self.config['bc'] = str(_cfg_req(cfg, "bc"))
        self.config['eps'] = float(_cfg_req(cfg, "eps"))
        return self

    def build_constant(self) -> MPMConstant:
        c = MPMConstant()
        for k in ("num_grids", "dt", "bound", "gravity", "dx", "inv_dx", "eps"):
            setattr(c, k, self.config[k])
        return c

    def finalize(self, device=None, requires_grad: bool = False) -> MPMModel:
        if not self.ready:
            raise RuntimeError(f'config uninitialized: {self.config}')
        bc = self.config['bc']
        if bc not in ('freeslip', 'noslip'):
            raise ValueError('invalid boundary condition: {}'.format(bc))   # mpm.py:550
        return MPMModel(self.build_constant(), device, requires_grad, bc=bc)


def _per_particle(values, counts, dtype, device) -> Tensor:
    """Expand one value per group to one value per particle (group g owns counts[g] consecutive particles)."""
    vals = torch.as_tensor(np.asarray(values, dtype=np.float64), dtype=dtype)
    return torch.repeat_interleave(vals, torch.as_tensor(list(counts), dtype=torch.long)).to(device)


@dataclass
class MPMInitData(object):
    """One body of particles as the initializers consume it - field names of mpm.py:554-574 (rho, clip_bound, span,
    num_particles, vol, pos, lin_vel, ang_vel, center, ind_vel, bounds, size).  Particle positions come from an ndarray or
    from the `<name>.npz` cache (p_x, vol) the reference itself writes (mpm.py:653); sampling a mesh / PLY with trimesh is
    asset preprocessing and out of scope."""

    rho: float
    clip_bound: float
    span: tuple
    num_particles: int
    vol: float
    pos: np.ndarray
    lin_vel: np.ndarray = field(default_factory=lambda: np.zeros(3))
    ang_vel: np.ndarray = field(default_factory=lambda: np.zeros(3))
    center: Optional[np.ndarray] = None
    ind_vel: Optional[np.ndarray] = None
    bounds: Optional[np.ndarray] = None
    size: Optional[np.ndarray] = None

    def __post_init__(self) -> None:
        if self.center is None:
            self.center = self.pos.mean(0)          # rotation centre of ang_vel defaults to the centroid

    @staticmethod
    def alignment(min_bound_1, max_bound_1, min_bound_2, max_bound_2):
        """Per-axis affine map p -> p * scale + shift taking box 1 onto box 2 (mpm.py:576-594). Returns (scale, shift)."""
        lo1, hi1, lo2, hi2 = (np.asarray(b, dtype=np.float64) for b in (min_bound_1, max_bound_1, min_bound_2, max_bound_2))
        scale = (hi2 - lo2) / (hi1 - lo1)
        shift = 0.5 * (lo2 + hi2) - 0.5 * (lo1 + hi1) * scale
        return scale, shift

    @classmethod
    def get(cls, cfg) -> 'MPMInitData':
        """From a `particle_data` config node (rho, clip_bound, span, shape.{name, asset_root, sort, ori_bounds, sim_bounds})."""
        shape = _cfg_req(cfg, "shape")
        body = cls.get_pcd(_cfg_req(shape, "name"), _cfg_get(shape, "asset_root"), _cfg_get(shape, "sort"),
                           _cfg_get(shape, "ori_bounds"), _cfg_get(shape, "sim_bounds"))
        return cls(rho=_cfg_req(cfg, "rho"), clip_bound=_cfg_req(cfg, "clip_bound"), span=tuple(_cfg_req(cfg, "span")), **body)

    @classmethod
    def get_pcd(cls, name, asset_root, sort=None, ori_bounds=None, sim_bounds=None) -> dict:
        """Counterpart of mpm.py:607-677: the `<name>.npz` cache (p_x, vol) if present, else `<name>.ply` (a point cloud; sorted
        descending along `sort`), whose per-particle volume is the volume of the single `mesh.*` file next to it divided by
        the particle count, or of the points' convex hull when there is no such mesh; the cache is then written, as the
        reference does.  (trimesh is replaced by the PLY / OBJ readers of neuma_amd.io and scipy's ConvexHull.)"""
        assert ori_bounds is not None, "ori_bounds must be provided for pcd shape."
        assert sim_bounds is not None, "sim_bounds must be provided for pcd shape."
        root = Path(asset_root) if asset_root is not None else Path("experiments") / "assets"
        cache = root / f"{name}.npz"
        if cache.is_file():
            with np.load(cache) as file:
                return cls.from_points(file['p_x'], float(file['vol']), ori_bounds, sim_bounds, None)
        pcd_path = root / f"{name}.ply"
        if not pcd_path.is_file():
            raise FileNotFoundError(f"neither {cache} nor {pcd_path} exists")
        from .. import io as nio
        from ..extras import mesh_sampling as mesh      # (outside the section-8 scope: see its header)
        p_x = nio.load_particles_ply(pcd_path)
        if sort is not None:
            p_x = p_x[np.argsort(-p_x[:, sort], kind="stable")]
        meshes = sorted(pcd_path.parent.glob("mesh.*"))
        if len(meshes) == 1:
            reader = mesh.read_obj_mesh if meshes[0].suffix.lower() == ".obj" else mesh.read_ply_mesh
            vol = abs(mesh.mesh_volume(*reader(meshes[0]))) / p_x.shape[0]
        else:
            from scipy.spatial import ConvexHull
            vol = float(ConvexHull(p_x).volume) / p_x.shape[0]
            print('  WARNING: mesh file not found, using convex hull volume.')
        np.savez(cache, p_x=p_x, vol=vol)
        return cls.from_points(p_x, float(vol), ori_bounds, sim_bounds, None)

    @classmethod
    def from_points(cls, p_x: np.ndarray, vol: float, ori_bounds, sim_bounds, sort=None) -> dict:
        """Map raw points and their per-particle volume from `ori_bounds` into the unit-cube box `sim_bounds`."""
        pts = np.array(p_x, dtype=np.float64).reshape(-1, 3)
        if sort is not None:
            pts = pts[np.argsort(-pts[:, sort], kind="stable")]            # descending along axis `sort`
        ori, sim = np.asarray(ori_bounds, dtype=np.float64), np.asarray(sim_bounds, dtype=np.float64)
        scale, shift = cls.alignment(ori[0], ori[1], sim[0], sim[1])
        pts = np.ascontiguousarray(pts * scale + shift)
        if pts.min() < 0.0 or pts.max() > 1.0:                              # same condition as the asserts of mpm.py:673-675
            raise AssertionError("particles leave the unit cube after mapping ori_bounds -> sim_bounds")
        return dict(num_particles=pts.shape[0], vol=vol * float(np.prod(scale)), pos=pts, center=shift, size=scale)

    def set_lin_vel(self, value) -> None:
        self.lin_vel = np.array(value)

    def zero_lin_vel(self) -> None:
        self.lin_vel = np.zeros_like(self.lin_vel)

    def set_ang_vel(self, value) -> None:
        self.ang_vel = np.array(value)

    def zero_ang_vel(self) -> None:
        self.ang_vel = np.zeros_like(self.ang_vel)

    def set_ind_vel(self, ind_vel) -> None:
        self.ind_vel = np.array(ind_vel)

    def velocities(self) -> np.ndarray:
        """Initial particle velocities: the individual ones if given, else rigid motion lin_vel + ang_vel x (pos - center)."""
        if self.ind_vel is not None:
            return np.asarray(self.ind_vel, dtype=np.float64)
        arm = np.asarray(self.pos, dtype=np.float64) - np.asarray(self.center, dtype=np.float64)
        return np.asarray(self.lin_vel, dtype=np.float64) + np.cross(np.asarray(self.ang_vel, dtype=np.float64), arm)


class MPMStateInitializer(StateInitializer):
    """Builds the initial MPMState of all added bodies, concatenated in the order added (interface of mpm.py:695-735):
    finalize() -> (state, sections)."""

    StateType = MPMState
    ModelType = MPMModel

    def __init__(self, model) -> None:
        super().__init__(model)
        self.groups = []

    def add_group(self, group: MPMInitData) -> None:
        self.groups.append(group)

    def finalize(self):
        sections = [int(g.num_particles) for g in self.groups]
        state = super().finalize(shape=sum(sections), requires_grad=False)
        x = np.concatenate([np.asarray(g.pos, dtype=np.float64) for g in self.groups], axis=0)
        v = np.concatenate([g.velocities() for g in self.groups], axis=0)
        state.particle.x.copy_(torch.from_numpy(x).to(torch.float32))
        state.particle.v.copy_(torch.from_numpy(v).to(torch.float32))
        return state, sections


class MPMStaticsInitializer(StaticsInitializer):
    """Per-particle constants of all added bodies (interface of mpm.py:738-776): finalize() -> statics;
    update(statics, step) re-evaluates `enabled` from each body's span."""

    StaticsType = MPMStatics
    ModelType = MPMModel

    def __init__(self, model) -> None:
        super().__init__(model)
        self.groups = []
        self.sections, self.vols, self.rhos, self.clip_bounds, self.spans = [], [], [], [], []

    def add_group(self, group: MPMInitData) -> None:
        self.groups.append(group)

    def update(self, statics, step: int = 0) -> None:
        statics.update_enabled(self.sections, self.spans, step=step)

    def finalize(self):
        new = self.groups[len(self.sections):]          # bodies not tabulated yet (finalize may follow further add_group calls)
        self.sections += [int(g.num_particles) for g in new]
        self.vols += [g.vol for g in new]
        self.rhos += [g.rho for g in new]
        self.clip_bounds += [g.clip_bound for g in new]
        self.spans += [tuple(g.span) for g in new]
        statics = super().finalize(shape=sum(self.sections))
        dev = statics.vol.device
        statics.vol.copy_(_per_particle(self.vols, self.sections, torch.float32, dev))
        statics.rho.copy_(_per_particle(self.rhos, self.sections, torch.float32, dev))
        statics.clip_bound.copy_(_per_particle(self.clip_bounds, self.sections, torch.float32, dev))
        self.update(statics, step=0)
        return statics
