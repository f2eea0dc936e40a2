"""ctypes front-end of oracle/c/libneuma_ref.so - the C++/OpenMP restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
All arrays are numpy, C-contiguous, float32 / int32; functions return new arrays."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent / "c"
LIB = HERE / "libneuma_ref.so"
_lib = None

_F, _I, _P = C.c_float, C.c_int32, C.c_void_p


class SimCfg(C.Structure):
    _fields_ = [("G", _I), ("dt", _F), ("bound", _I), ("gravity", _F * 3), ("eps", _F), ("bc", _I)]


class Cam(C.Structure):
    _fields_ = [("H", _I), ("W", _I), ("tanfovx", _F), ("tanfovy", _F), ("bg", _F * 3), ("view", _F * 16), ("proj", _F * 16),
                ("sh_degree", _I), ("campos", _F * 3)]


def build():
    subprocess.run(["make", "-C", str(HERE)], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        # idle OpenMP threads sleep instead of spinning: under a cgroup CPU quota (the GPU boxes: 16 of 256 visible CPUs) spinning
        # waiters spend the quota the working threads need (a 256-thread frame took 95x a 32-thread one)
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        os.environ.setdefault("GOMP_SPINCOUNT", "0")
        h = C.CDLL(str(LIB))
        h.ref_num_threads.restype = C.c_int
        h.ref_raster_forward.restype = _P
        h.ref_raster_pairs.restype = C.c_int64
        h.ref_raster_pairs.argtypes = [_P]
        h.ref_raster_free.argtypes = [_P]
        h.ref_pixel_loss.restype = C.c_double
        _lib = h
    return _lib


def threads() -> int:
    return int(lib().ref_num_threads())


def set_threads(n: int):
    lib().ref_set_num_threads(int(n))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def sim_cfg(G, dt, bound=1, gravity=(0.0, -9.8, 0.0), eps=6e-7, bc="noslip"):
    return SimCfg(int(G), float(dt), int(bound), (_F * 3)(*[float(g) for g in gravity]), float(eps), {"noslip": 0, "freeslip": 1}[bc])


class Grid:
    """The dense grid arrays of one MPMModel (mpm.py:131-155, 251-258) plus adjoint scratch."""

    def __init__(self, G):
        n = G ** 3
        self.mv = np.zeros((n, 3), np.float32); self.m = np.zeros(n, np.float32); self.v = np.zeros((n, 3), np.float32)
        self.gv = np.zeros((n, 3), np.float32); self.gm = np.zeros((n, 4), np.float32)


def mpm_forward(cfg, grid, vol, rho, clip, en, x, v, Cm, F, S, out=None):
    N = x.shape[0]
    x, v, Cm, F, S = _f(x), _f(v), _f(Cm), _f(F), _f(S)
    if out is None:
        out = (x.copy(), v.copy(), Cm.copy(), F.copy())     # disabled particles: next state left as it was (pass-through)
    xn, vn, Cn, Fn = out
    lib().ref_mpm_forward(C.byref(cfg), N, _p(_f(vol)), _p(_f(rho)), _p(_f(clip)), _p(np.ascontiguousarray(en, np.int32)),
                          _p(x), _p(v), _p(Cm), _p(F), _p(S), _p(xn), _p(vn), _p(Cn), _p(Fn), _p(grid.mv), _p(grid.m), _p(grid.v))
    return xn, vn, Cn, Fn


def mpm_backward(cfg, grid, vol, rho, clip, en, x, v, Cm, F, S, vn, Cn, gxn, gvn, gCn, gFn):
    N = x.shape[0]
    gx = np.zeros((N, 3), np.float32); gv = np.zeros((N, 3), np.float32)
    gC = np.zeros((N, 3, 3), np.float32); gF = np.zeros((N, 3, 3), np.float32); gS = np.zeros((N, 3, 3), np.float32)
    args = [_f(a) for a in (vol, rho, clip)] + [np.ascontiguousarray(en, np.int32)] + [_f(a) for a in (x, v, Cm, F, S, vn, Cn, gxn, gvn, gCn, gFn)]
    lib().ref_mpm_backward(C.byref(cfg), N, *[_p(a) for a in args], _p(gx), _p(gv), _p(gC), _p(gF), _p(gS), _p(grid.mv), _p(grid.m),
                           _p(grid.v), _p(grid.gv), _p(grid.gm))
    return gx, gv, gC, gF, gS


def material_forward(kind, alpha, F, W):
    F = _f(F)
    out = np.empty_like(F)
    W = [_f(w) for w in W]
    lib().ref_material_forward(int(kind), _F(alpha), F.shape[0], _p(F), _p(W[0]), _p(W[1]), _p(W[2]), _p(out))
    return out


def material_backward(kind, alpha, F, W, gout, svd_clamp=1e-6):
    F, gout = _f(F), _f(gout)
    W = [_f(w) for w in W]
    gF = np.empty_like(F)
    gW = [np.empty_like(w) for w in W]
    lib().ref_material_backward(int(kind), _F(alpha), F.shape[0], _p(F), _p(W[0]), _p(W[1]), _p(W[2]), _p(gout), _p(gF), _p(gW[0]),
                                _p(gW[1]), _p(gW[2]), _F(svd_clamp))
    return gF, gW


def svd3(F):
    F = _f(F)
    U, s, Vh = np.empty_like(F), np.empty((F.shape[0], 3), np.float32), np.empty_like(F)
    lib().ref_svd3(F.shape[0], _p(F), _p(U), _p(s), _p(Vh))
    return U, s, Vh


def spmm(rowptr, col, val, x, base=None):
    K, D = rowptr.shape[0] - 1, int(np.prod(x.shape[1:]))
    x = _f(x).reshape(x.shape[0], D)
    out = np.empty((K, D), np.float32)
    lib().ref_spmm_csr(K, D, _p(rowptr), _p(col), _p(val), _p(x), _p(None if base is None else _f(base)), _p(out))
    return out


def spmm_t(rowptr, col, val, g, N):
    K, D = g.shape[0], int(np.prod(g.shape[1:]))
    g = _f(g).reshape(K, D)
    out = np.empty((N, D), np.float32)
    lib().ref_spmm_csr_t(K, N, D, _p(rowptr), _p(col), _p(val), _p(g), _p(out))
    return out


def cov_deform(cov6, F):
    cov6, F = _f(cov6), _f(F)
    out = np.empty_like(cov6)
    lib().ref_cov_deform(cov6.shape[0], _p(cov6), _p(F), _p(out))
    return out


def pixel_loss(kind, img, gt, weight=1.0):
    img, gt = _f(img), _f(gt)
    g = np.empty_like(img)
    val = lib().ref_pixel_loss({"l1": 0, "l2": 1}[kind], C.c_int64(img.size), _p(img), _p(gt), _F(weight), _p(g))
    return float(val), g


def camera(H, W, tanfovx, tanfovy, bg, view, proj, sh_degree, campos):
    return Cam(int(H), int(W), float(tanfovx), float(tanfovy), (_F * 3)(*[float(b) for b in bg]),
               (_F * 16)(*[float(t) for t in np.asarray(view, np.float64).reshape(-1)]),
               (_F * 16)(*[float(t) for t in np.asarray(proj, np.float64).reshape(-1)]), int(sh_degree),
               (_F * 3)(*[float(t) for t in campos]))


class Raster:
    """One forward render; keeps the C-side state for backward()."""

    def __init__(self, cam, means3D, opac, cov6, shs=None, colors=None):
        self.cam = cam
        self.keep = [_f(means3D), None if shs is None else _f(shs), None if colors is None else _f(colors), _f(opac).reshape(-1), _f(cov6)]
        K = self.keep[0].shape[0]
        M = 0 if shs is None else self.keep[1].shape[1]
        self.image = np.empty((3, cam.H, cam.W), np.float32)
        self.radii = np.empty(K, np.int32)
        self.K, self.M = K, M
        self.h = lib().ref_raster_forward(C.byref(cam), K, M, _p(self.keep[0]), _p(self.keep[1]), _p(self.keep[2]), _p(self.keep[3]),
                                          _p(self.keep[4]), _p(self.image), _p(self.radii))
        self.pairs = int(lib().ref_raster_pairs(self.h))

    def backward(self, gimg, want=("means3D",)):
        gimg = _f(gimg)
        out = {"means3D": np.empty((self.K, 3), np.float32)}
        dcov = np.empty((self.K, 6), np.float32) if "cov" in want else None
        dop = np.empty(self.K, np.float32) if "opacity" in want else None
        dsh = np.empty((self.K, self.M, 3), np.float32) if ("shs" in want and self.M) else None
        dcol = np.empty((self.K, 3), np.float32) if "colors" in want else None
        lib().ref_raster_backward(C.c_void_p(self.h), _p(gimg), _p(out["means3D"]), _p(dcov), _p(dop), _p(dsh), _p(dcol))
        out.update(cov=dcov, opacity=dop, shs=dsh, colors=dcol)
        return out

    def close(self):
        if self.h:
            lib().ref_raster_free(C.c_void_p(self.h))
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
