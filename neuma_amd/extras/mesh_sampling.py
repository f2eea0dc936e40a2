"""OUTSIDE the hot-path scope (data preparation is OUT OF SCOPE; section 8 names no mesh code).

Kept apart from the path's own readers (neuma_amd/io.py) on purpose: a convenience for running the finetune / render entry points
on a machine that has neither trimesh nor the reference's prebuilt `VolumeSampling` binary - triangle-mesh readers (PLY / OBJ),
volume sampling by a ray-parity inside test, and the mesh volume.  Nothing under csrc/, nothing on the frame path and no parity
claim depends on it; prepare.py and MPMInitData.get_pcd use it only when the `.npz` particle cache the reference writes is absent."""
from typing import Tuple

import numpy as np

from ..io import _ply_header, _ply_vertex_block


def read_ply_mesh(path):
    """(vertices (n,3) float64, triangles (m,3) int64) of a PLY mesh: `vertex` element followed by a `face` element with one
    list property; polygons are fan-triangulated."""
    with open(path, "rb") as f:
        fmt, elements = _ply_header(f, path)
        v = _ply_vertex_block(f, path, fmt, elements[0])
        verts = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float64)
        face = next((e for e in elements[1:2] if e["name"] == "face"), None)
        if face is None or len(face["props"]) < 1 or face["props"][0][0] != "list":
            raise ValueError(f"{path}: no `face` element right after `vertex`")
        tris = []
        if fmt == "ascii":
            for _ in range(face["count"]):
                tok = f.readline().split()
                k = int(tok[0])
                idx = [int(t) for t in tok[1:1 + k]]
                tris += [(idx[0], idx[i], idx[i + 1]) for i in range(1, k - 1)]
        else:
            end = "<" if fmt == "binary_little_endian" else ">"
            ct, it = (np.dtype(end + _PLY_TYPES[t]) for t in face["props"][0][2])
            extra = sum(np.dtype(p[0]).itemsize for p in face["props"][1:])       # fixed-size properties after the list
            for _ in range(face["count"]):
                k = int(np.frombuffer(f.read(ct.itemsize), dtype=ct)[0])
                idx = np.frombuffer(f.read(it.itemsize * k), dtype=it).astype(np.int64)
                if extra:
                    f.read(extra)
                tris += [(idx[0], idx[i], idx[i + 1]) for i in range(1, k - 1)]
        return verts, np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def read_obj_mesh(path):
    """(vertices, triangles) of a Wavefront OBJ (v / f records; polygons fan-triangulated, negative indices supported)."""
    verts, tris = [], []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "v":
                verts.append([float(t) for t in tok[1:4]])
            elif tok[0] == "f":
                idx = [int(t.split("/")[0]) for t in tok[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                tris += [(idx[0], idx[i], idx[i + 1]) for i in range(1, len(idx) - 1)]
    return np.asarray(verts, dtype=np.float64), np.asarray(tris, dtype=np.int64).reshape(-1, 3)


def points_in_mesh(points: np.ndarray, verts: np.ndarray, tris: np.ndarray) -> np.ndarray:
    """Inside test of a closed triangle mesh by ray parity along +z (counterpart of trimesh's `mesh.contains`, which the
    reference's samplers rely on, tune/utils.py:49-200).  points (n,3) -> bool (n,)."""
    p = np.asarray(points, dtype=np.float64)
    a, b, c = (np.asarray(verts, dtype=np.float64)[np.asarray(tris)[:, k]] for k in range(3))
    inside = np.zeros(len(p), dtype=bool)
    # (the ray is moved off the point by a tiny irrational offset: a grid point exactly above a triangle edge would otherwise
    # count both triangles)
    span = float(np.abs(np.asarray(verts)).max()) or 1.0
    p = p + np.array([1.2345678e-7, 2.7182818e-7, 0.0]) * span
    d = (b[:, 1] - c[:, 1]) * (a[:, 0] - c[:, 0]) + (c[:, 0] - b[:, 0]) * (a[:, 1] - c[:, 1])       # 2 x signed area in xy
    ok = np.abs(d) > 1e-300
    a, b, c, d = a[ok], b[ok], c[ok], d[ok]
    for i0 in range(0, len(p), 2048):
        q = p[i0:i0 + 2048]
        px, py = q[:, 0:1], q[:, 1:2]
        l0 = ((b[:, 1] - c[:, 1])[None] * (px - c[:, 0][None]) + (c[:, 0] - b[:, 0])[None] * (py - c[:, 1][None])) / d[None]
        l1 = ((c[:, 1] - a[:, 1])[None] * (px - c[:, 0][None]) + (a[:, 0] - c[:, 0])[None] * (py - c[:, 1][None])) / d[None]
        l2 = 1.0 - l0 - l1
        hit = (l0 >= 0) & (l1 >= 0) & (l2 >= 0)
        z = l0 * a[:, 2][None] + l1 * b[:, 2][None] + l2 * c[:, 2][None]
        inside[i0:i0 + 2048] = ((hit & (z > q[:, 2:3])).sum(1) % 2) == 1
    return inside


def sample_mesh_points(verts: np.ndarray, tris: np.ndarray, mode: str = "volumetric", resolution: int = 30, seed: int = 0) -> np.ndarray:
    """Particles inside a closed mesh (tune/utils.py:49-200 without trimesh / the prebuilt VolumeSampling binary):
    'volumetric' = the points of a regular grid with `resolution` cells along the longest side of the bounding box that lie
    inside; 'uniform' = resolution^3 uniformly random points of the bounding box, those inside kept."""
    lo, hi = verts.min(0), verts.max(0)
    if mode == "volumetric":
        h = float((hi - lo).max()) / int(resolution)
        axes = [np.arange(lo[k] + 0.5 * h, hi[k], h) for k in range(3)]
        pts = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)
    elif mode == "uniform":
        pts = lo + (hi - lo) * np.random.default_rng(seed).random((int(resolution) ** 3, 3))
    else:
        raise ValueError(f"mesh_sample_mode '{mode}' is not available here (volumetric / uniform)")
    return pts[points_in_mesh(pts, verts, tris)]


def mesh_volume(verts: np.ndarray, tris: np.ndarray) -> float:
    """Signed volume of a closed triangle mesh (sum of tetrahedra against the origin), what trimesh's `mesh.volume` returns."""
    a, b, c = verts[tris[:, 0]], verts[tris[:, 1]], verts[tris[:, 2]]
    return float(np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0)
