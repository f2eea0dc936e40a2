"""Deterministic synthetic workloads: particle ball, bound Gaussians, ring cameras.
Datasets and checkpoints cannot be downloaded here, and the reference ships no assets, so every benchmark /
parity configuration of BASELINE.json is generated from this file.  numpy/scipy on the host, one-off setup."""
import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch

CONFIGS = {
    # name: particles, grid, gaussians, (W,H), dt, substeps/frame, views, sh_degree, material
    "bb":     dict(N=8_000,     G=64,  K=16_000,  W=256,  H=256,  dt=1e-3, S=1,  V=1, sh=3, mat="jelly"),
    "jd":     dict(N=50_000,    G=128, K=100_000, W=800,  H=800,  dt=1e-3, S=1,  V=1, sh=3, mat="jelly"),
    "sf":     dict(N=100_000,   G=128, K=100_000, W=800,  H=800,  dt=5e-4, S=1,  V=1, sh=3, mat="sand"),
    "burger": dict(N=80_000,    G=128, K=200_000, W=1920, H=1080, dt=5e-4, S=20, V=3, sh=0, mat="jelly", bg="black"),
    "metric": dict(N=100_000,   G=128, K=200_000, W=1920, H=1080, dt=5e-4, S=20, V=3, sh=3, mat="jelly"),
    "stress": dict(N=1_000_000, G=256, K=500_000, W=1920, H=1080, dt=5e-4, S=1,  V=1, sh=3, mat="jelly"),
    "tiny":   dict(N=2_000,     G=32,  K=3_000,   W=128,  H=96,   dt=1e-3, S=2,  V=2, sh=3, mat="jelly"),
}


def hilbert_index(cells: np.ndarray, bits: int) -> np.ndarray:
    """Index of integer 3-D cells along a Hilbert curve of 2^bits cells per axis (Skilling's transpose algorithm):
    consecutive indices are face-adjacent cells."""
    X = np.array(cells, dtype=np.int64).copy()
    n = X.shape[1]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(n):
            mask = (X[:, i] & Q) != 0
            X[mask, 0] ^= P
            nm = ~mask
            t = (X[nm, 0] ^ X[nm, i]) & P
            X[nm, 0] ^= t
            X[nm, i] ^= t
        Q >>= 1
    for i in range(1, n):
        X[:, i] ^= X[:, i - 1]
    t = np.zeros(len(X), dtype=np.int64)
    Q = M
    while Q > 1:
        mask = (X[:, n - 1] & Q) != 0
        t[mask] ^= (Q - 1)
        Q >>= 1
    for i in range(n):
        X[:, i] ^= t
    idx = np.zeros(len(X), dtype=np.int64)
    for b in range(bits - 1, -1, -1):
        for i in range(n):
            idx = (idx << 1) | ((X[:, i] >> b) & 1)
    return idx


def stencil_order(P: np.ndarray, G: int, curve: str = "hilbert") -> np.ndarray:
    """Permutation that sorts particles by the origin of their 3x3x3 stencil, base = int(x*G - 0.5) (mpm.py:336-339),
    along a space-filling curve over the base cells.  Particles sharing a base touch the same 27 nodes and stay
    contiguous; any run of consecutive particles is spatially compact, so the scatter kernels' per-workgroup node
    tiles stay small.  A Hilbert curve never jumps (a Morton / Z-order curve does at every power-of-two boundary,
    which leaves a few workgroups with two or three disjoint clusters and makes them the kernel's critical path).
    Any order gives the same results - this one is the fast one (a load-time sort like the reference's `sort`
    option, mpm.py:640-642)."""
    base = np.trunc(P.astype(np.float64) * G - 0.5).astype(np.int64)
    base = np.clip(base, 0, None)
    if curve == "hilbert":
        bits = max(1, int(np.ceil(np.log2(G + 2))))
        return np.argsort(hilbert_index(base, bits), kind="stable")
    blk = base // 4

    def spread(v):  # Morton bit spread of a 10-bit integer
        v = v & 0x3FF
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v

    morton = (spread(blk[:, 0]) << 2) | (spread(blk[:, 1]) << 1) | spread(blk[:, 2])
    key = morton * 64 + ((base[:, 0] % 4) * 16 + (base[:, 1] % 4) * 4 + base[:, 2] % 4)
    return np.argsort(key, kind="stable")


def ball_particles(N: int, G: int, centers=((0.5, 0.5, 0.5),), seed: int = 0):
    """Jittered lattice at spacing dx/2 (8 particles per cell), the N/len(centers) lattice points nearest each
    centre, jitter U(-dx/8, dx/8).  Returned in stencil order (see stencil_order)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    dx = 1.0 / G
    h = dx / 2
    per = N // len(centers)
    out = []
    for c in centers:
        r = (3 * per / (4 * math.pi)) ** (1 / 3) * h * 1.15 + 2 * h
        m = int(math.ceil(r / h))
        ax = (np.arange(-m, m + 1) + 0.25) * h
        X = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
        d = (X ** 2).sum(1)
        idx = np.argpartition(d, per)[:per]
        P = X[idx] + np.asarray(c)[None]
        P = P + rng.uniform(-dx / 8, dx / 8, size=P.shape)
        out.append(P)
    P = np.concatenate(out, 0)
    P = P[stencil_order(P, G)]
    assert P.min() > 0.05 and P.max() < 0.95
    return P.astype(np.float32)


def look_at(eye, target, up, fovx, fovy, znear=0.01, zfar=100.0):
    """(world_view_transform, full_proj_transform, camera_center) in the reference convention
    (cameras.py:54-57, graphics_utils.py:51-71): row-vector, +x right, +y down, +z forward."""
    eye, target, up = (np.asarray(a, dtype=np.float64) for a in (eye, target, up))
    fwd = target - eye; fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rw2c = np.stack([right, down, fwd], 0)
    Rt = np.eye(4); Rt[:3, :3] = Rw2c; Rt[:3, 3] = -Rw2c @ eye
    wv = Rt.T
    P = np.zeros((4, 4))
    P[0, 0] = 1 / math.tan(fovx / 2); P[1, 1] = 1 / math.tan(fovy / 2)
    P[3, 2] = 1.0; P[2, 2] = zfar / (zfar - znear); P[2, 3] = -(zfar * znear) / (zfar - znear)
    full = wv @ P.T
    center = np.linalg.inv(wv)[3, :3]
    return wv.astype(np.float32), full.astype(np.float32), center.astype(np.float32)


class SynthCamera(object):
    """Duck-type of cameras.py Camera/MiniCam for get_rasterizer."""

    def __init__(self, W, H, fovx, eye, target=(0.5, 0.5, 0.5), up=(0.0, -1.0, 0.0), device="cpu"):
        self.image_width, self.image_height = int(W), int(H)
        self.FoVx = float(fovx)
        self.FoVy = 2 * math.atan(math.tan(fovx / 2) * H / W)
        wv, full, c = look_at(eye, target, up, self.FoVx, self.FoVy)
        self.world_view_transform = torch.tensor(wv, device=device)
        self.full_proj_transform = torch.tensor(full, device=device)
        self.camera_center = torch.tensor(c, device=device)
        self.original_image = None


def ring_cameras(V: int, W: int, H: int, device="cpu", dist=1.5, fov_deg=40.0) -> List[SynthCamera]:
    cams = []
    for i in range(V):
        a = 2 * math.pi * i / max(V, 1) + 0.3
        eye = (0.5 + dist * math.sin(a), 0.5 + 0.25, 0.5 + dist * math.cos(a))
        cams.append(SynthCamera(W, H, math.radians(fov_deg), eye, device=device))
    return cams


@dataclass
class Scene:
    name: str
    cfg: dict
    x0: np.ndarray            # (N,3) float32 in [0,1]^3
    v0: np.ndarray
    vol: float
    g_xyz: np.ndarray         # (K,3)
    g_logscale: np.ndarray    # (K,3)
    g_rot: np.ndarray         # (K,4)
    g_opacity_logit: np.ndarray  # (K,1)
    g_sh: np.ndarray          # (K,(deg+1)^2,3)
    bind_idx: np.ndarray      # (K,nb) particle ids
    bind_w: np.ndarray        # (K,nb)


def make_scene(name: str, seed: int = 0, nbind: int = 8, override: Optional[dict] = None) -> Scene:
    from scipy.spatial import cKDTree
    cfg = dict(CONFIGS[name])
    if override:
        cfg.update(override)
    N, G, K = cfg["N"], cfg["G"], cfg["K"]
    centers = ((0.5, 0.5, 0.5),) if name != "stress" else ((0.3, 0.5, 0.3), (0.7, 0.5, 0.3), (0.3, 0.5, 0.7), (0.7, 0.5, 0.7))
    x0 = ball_particles(N, G, centers, seed)
    N = x0.shape[0]
    cfg["N"] = N
    dx = 1.0 / G
    v0 = np.tile(np.array([[0.0, -0.5, 0.0]], dtype=np.float32), (N, 1))
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    pick = rng.integers(0, N, size=K)
    g_xyz = (x0[pick] + rng.normal(0, dx, size=(K, 3))).astype(np.float32)
    g_logscale = rng.uniform(math.log(0.5 * dx), math.log(2 * dx), size=(K, 3)).astype(np.float32)
    q = rng.normal(size=(K, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    g_op = rng.normal(2.0, 1.0, size=(K, 1)).astype(np.float32)
    M = (cfg["sh"] + 1) ** 2
    g_sh = rng.normal(0, 0.3, size=(K, M, 3)).astype(np.float32)
    tree = cKDTree(x0)
    _, idx = tree.query(g_xyz, k=nbind, workers=-1)
    idx = idx.reshape(K, nbind)
    w = np.full((K, nbind), 1.0 / nbind, dtype=np.float32)       # binding_utils.py:265-275: equal weights 1/n
    return Scene(name, cfg, x0, v0, float((dx / 2) ** 3), g_xyz, g_logscale, q.astype(np.float32), g_op, g_sh,
                 idx.astype(np.int64), w)


def load_base_weights(material: str, root=None):
    """Shipped NeuMA checkpoints as plain arrays: package data neuma_amd/data/base_models.npz (the three
    experiments/base_models/*_0300.pt files converted by tests/golden/gen_material_golden.py)."""
    from pathlib import Path
    p = Path(root) if root else Path(__file__).resolve().parent / "data" / "base_models.npz"
    z = np.load(p)
    return {t: [z[f"{material}_{t}_w{i}"] for i in range(3)] for t in "ep"}
