"""A scalar, numpy-backed stand-in for the subset of `warp` (warp-lang 0.6.1 semantics) that the reference's
MPM / SVD / covariance kernels use.  GENERATOR-SIDE TEST INFRASTRUCTURE: it exists so that
`tests/golden/gen_mpm_golden.py` can import the reference's own modules from /root/reference and EXECUTE the
Python bodies of its `@wp.kernel` functions (one Python call per thread id) to produce golden vectors.
Nothing in the product or in the tests imports this file; it never travels to the GPU box in a role other
than documentation of how the fixtures were made.

Semantics reproduced (from the Warp 0.6 documentation and the reference's own comments):
  * `float` is fp32 (set FLOAT = np.float64 to execute the same code in fp64), `int` is int32;
  * `wp.vec3(s)` broadcasts, `wp.vec3(a, b, c)`; `wp.mat33(v0, v1, v2)` takes COLUMN vectors (mpm.py:350
    "wp.mat33(col_vec, col_vec, col_vec)"), `wp.mat33(9 scalars)` is row-major;
  * `m * v`, `m * m` are matrix products, `s * m`, `m * s`, `s * v`, `v * s`, `v / s` scale; `m[i, j]`, `v[i]` index;
  * `wp.cw_mul`, `wp.outer(a, b) = a b^T`, `wp.transpose`, `wp.determinant`, `wp.clamp`;
  * `wp.atomic_add(array, i, j, k, value)`: executed sequentially in thread order (one legal atomics order);
  * `wp.svd3(A, U, sigma, V)` writes U, sigma, V with A = U diag(sigma) V^T - backed by numpy's SVD.  Warp's own
    routine is absent, so a caller can choose whether the raw factors come back as numpy gives them (sigma >= 0,
    det(U), det(V) = +-1: exercises every branch of the reference's det/sign rule, svd.py:76-92) - see `SVD3_MODE`;
  * `wp.launch(kernel, dim, inputs, device)`: Python loop over thread ids, `wp.tid()` returns the id (a tuple for
    multi-dimensional launches);
  * `wp.struct` returns an object with `.cls` whose call builds an instance deriving from the decorated class
    (so the class's own methods, e.g. `MPMStatics.update_enabled`, run), `wp.zeros/empty/zeros_like`, array
    `.zero_() .assign() .numpy() .shape .grad .requires_grad .device`, `wp.to_torch / wp.from_torch` (zero-copy).
"""
import itertools
import types

import numpy as np

FLOAT = np.float32          # switchable: gen_mpm_golden runs the reference in fp32 (its own precision) and fp64
SVD3_MODE = "numpy"         # "numpy": raw numpy factors; "rot": pre-rotated like McAdams' routine (U,V in SO(3))
_tid = None
oob_atomics = 0             # atomic_add calls that fell outside the array (reference UB) - must stay 0 for a fixture


def set_float(t):
    global FLOAT
    FLOAT = t


class vec3:
    __array_ufunc__ = None
    __slots__ = ("a",)

    def __init__(self, *args):
        if len(args) == 0:
            self.a = np.zeros(3, FLOAT)
        elif len(args) == 1:
            s = args[0]
            self.a = s.a.copy() if isinstance(s, vec3) else np.full(3, s, FLOAT)
        else:
            assert len(args) == 3
            self.a = np.array([FLOAT(x) for x in args], FLOAT)

    @staticmethod
    def wrap(a):
        v = vec3.__new__(vec3)
        v.a = np.asarray(a, FLOAT)
        return v

    def __getitem__(self, i):
        return self.a[i]

    def __add__(self, o):
        return vec3.wrap(self.a + o.a)

    def __sub__(self, o):
        return vec3.wrap(self.a - o.a)

    def __neg__(self):
        return vec3.wrap(-self.a)

    def __mul__(self, s):
        assert not isinstance(s, (vec3, mat33)), "vec * vec is not used by the reference kernels"
        return vec3.wrap(self.a * FLOAT(s))

    __rmul__ = __mul__

    def __truediv__(self, s):
        return vec3.wrap(self.a / FLOAT(s))


class mat33:
    __array_ufunc__ = None
    __slots__ = ("a",)

    def __init__(self, *args):
        if len(args) == 0:
            self.a = np.zeros((3, 3), FLOAT)
        elif len(args) == 3:
            self.a = np.stack([vec3(c).a for c in args], axis=1)      # columns
        elif len(args) == 9:
            self.a = np.array([FLOAT(x) for x in args], FLOAT).reshape(3, 3)   # row-major
        else:
            raise TypeError("mat33 takes 3 column vectors or 9 scalars")

    @staticmethod
    def wrap(a):
        m = mat33.__new__(mat33)
        m.a = np.asarray(a, FLOAT)
        return m

    def __getitem__(self, ij):
        if isinstance(ij, tuple):
            return self.a[ij]
        return vec3.wrap(self.a[ij].copy())

    def __add__(self, o):
        return mat33.wrap(self.a + o.a)

    def __sub__(self, o):
        return mat33.wrap(self.a - o.a)

    def __neg__(self):
        return mat33.wrap(-self.a)

    def __mul__(self, o):
        if isinstance(o, mat33):
            # products accumulated over k in order, one rounding per multiply-add step is NOT modelled (no fma)
            r = self.a[:, 0:1] * o.a[0:1, :]
            r = r + self.a[:, 1:2] * o.a[1:2, :]
            r = r + self.a[:, 2:3] * o.a[2:3, :]
            return mat33.wrap(r)
        if isinstance(o, vec3):
            r = self.a[:, 0] * o.a[0]
            r = r + self.a[:, 1] * o.a[1]
            r = r + self.a[:, 2] * o.a[2]
            return vec3.wrap(r)
        return mat33.wrap(self.a * FLOAT(o))

    def __rmul__(self, s):
        return mat33.wrap(self.a * FLOAT(s))


def cw_mul(a, b):
    return vec3.wrap(a.a * b.a)


def outer(a, b):
    return mat33.wrap(a.a[:, None] * b.a[None, :])


def transpose(m):
    return mat33.wrap(m.a.T.copy())


def determinant(m):
    a = m.a
    return FLOAT(a[0, 0] * (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1])
                 - a[0, 1] * (a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
                 + a[0, 2] * (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]))


def clamp(x, lo, hi):
    return min(max(x, FLOAT(lo)), FLOAT(hi))


def svd3(A, U, sigma, V):
    u, s, vt = np.linalg.svd(A.a.astype(np.float64))
    v = vt.T
    if SVD3_MODE == "rot":
        if np.linalg.det(u) < 0:
            u[:, 2] *= -1.0
            s[2] *= -1.0
        if np.linalg.det(v) < 0:
            v[:, 2] *= -1.0
            s[2] *= -1.0
    U.a = u.astype(FLOAT)
    sigma.a = s.astype(FLOAT)
    V.a = v.astype(FLOAT)


_ELEM = {float: (), int: (), vec3: (3,), mat33: (3, 3)}


class array:
    """wp.array: `data` has shape `shape + element shape`."""

    def __init__(self, data=None, dtype=float, shape=None, device=None, requires_grad=False, ndim=None, **kw):
        self.dtype = dtype
        self.device = device
        self.requires_grad = requires_grad
        self.grad = None
        self.data = data
        if data is None and shape is not None:
            shape = (shape,) if isinstance(shape, int) else tuple(shape)
            self.data = np.zeros(shape + _ELEM[dtype], np.int32 if dtype is int else FLOAT)
        if self.data is not None and requires_grad:
            self.grad = array(np.zeros_like(self.data), dtype=dtype, device=device)

    @property
    def shape(self):
        n = self.data.ndim - len(_ELEM[self.dtype])
        return tuple(self.data.shape[:n])

    def __bool__(self):
        return self.data is not None

    def __getitem__(self, idx):
        x = self.data[idx]
        if self.dtype is vec3:
            return vec3.wrap(x.copy())
        if self.dtype is mat33:
            return mat33.wrap(x.copy())
        return x

    def __setitem__(self, idx, val):
        self.data[idx] = val.a if isinstance(val, (vec3, mat33)) else val

    def zero_(self):
        self.data[...] = 0

    def assign(self, src):
        self.data[...] = np.asarray(src).reshape(self.data.shape)

    def numpy(self):
        return self.data


def zeros(shape=None, dtype=float, device=None, requires_grad=False, ndim=None, **kw):
    return array(dtype=dtype, shape=shape, device=device, requires_grad=requires_grad)


empty = zeros


def zeros_like(a):
    return array(np.zeros_like(a.data), dtype=a.dtype, device=a.device)


def atomic_add(arr, *args):
    global oob_atomics
    *idx, val = args
    idx = tuple(int(i) for i in idx)
    if any(i < 0 or i >= n for i, n in zip(idx, arr.shape)):
        oob_atomics += 1
        return
    arr.data[idx] += val.a if isinstance(val, (vec3, mat33)) else val


def tid():
    return _tid


def launch(kernel, dim, inputs, device=None, **kw):
    global _tid
    if isinstance(kernel, staticmethod):
        kernel = kernel.__func__
    dims = (dim,) if isinstance(dim, int) else tuple(int(d) for d in dim)
    try:
        if len(dims) == 1:
            for i in range(dims[0]):
                _tid = i
                kernel(*inputs)
        else:
            for t in itertools.product(*[range(d) for d in dims]):
                _tid = t
                kernel(*inputs)
    finally:
        _tid = None


class StructInstance:
    pass


class Struct:
    def __init__(self, cls):
        self.cls = cls

    def __call__(self):
        cls = self.cls
        inst = type(cls.__name__, (cls, StructInstance), {})()
        inst._struct_ = self
        return inst

    def __getattr__(self, name):        # static kernels reached through the class (MPMStatics.set_int)
        return getattr(self.cls, name)


def struct(cls):
    return Struct(cls)


def kernel(f):
    return f


func = kernel


class Tape:
    def __init__(self):
        self.gradients = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def backward(self):
        raise NotImplementedError("the stand-in executes forward kernels only (Warp's generated adjoints are not "
                                  "part of the reference tree)")


def get_device(device=None):
    return "cpu"


def device_from_torch(device):
    return "cpu"


def synchronize():
    pass


def from_torch(t, dtype=float, **kw):
    return array(t.detach().numpy(), dtype=dtype, device="cpu", requires_grad=False)


def to_torch(a):
    import torch
    return torch.from_numpy(a.data)


def install(sys_modules):
    """Register this module as `warp` (plus the submodule attributes the reference touches)."""
    import sys
    me = sys.modules[__name__]
    me.context = types.SimpleNamespace(Devicelike=object)
    me.codegen = types.SimpleNamespace(StructInstance=StructInstance)
    me.types = types.SimpleNamespace(array=array)
    me.float32 = np.float32
    me.int8 = np.int8
    sys_modules["warp"] = me
    return me
