"""Cell-size distribution of one view's (bin, depth slab) cells - what k_cell_sort has to sort.
    python tools/exp_cells.py [workload] [view]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from neuma_amd.render import _raster_inputs, deform_cov_by_F
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
lib = _lib.lib()
with torch.no_grad():
    x, v, C_, F = rt.rollout(*rt.start)
    m3 = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
    dg = compute_bindings_F(F, rt.bindings)
    rt.render_view(m3, dg, view)
rast = rt.cameras[view]._nm_raster_cache[0][1]
cfg = rast._cam.cfg
cov = deform_cov_by_F(rt._cov, dg)
m3c, sh, cp, op, cv = _raster_inputs(m3, rt._shs, None, rt._opacity, cov)
K = m3c.size(0)
cap = 16 * K
lib.nm_raster_state_bytes.restype = C.c_size_t
sb = int(lib.nm_raster_state_bytes(C.byref(cfg), K, cap))
state = torch.empty(sb, dtype=torch.uint8, device=dev)
radii = torch.empty(K, dtype=torch.int32, device=dev)
W, H = cfg.image_width, cfg.image_height
color = torch.empty(3, H, W, device=dev)
lib.nm_raster_forward(C.byref(cfg), K, sh.size(1), _lib.ptr(m3c), _lib.ptr(sh), None, _lib.ptr(op), _lib.ptr(cv), _lib.ptr(radii),
                      _lib.ptr(state), sb, cap, _lib.ptr(color), None, _lib.stream_ptr(dev))
nc = np.zeros(W * H, np.uint32)
off = np.zeros((1 << 20) + 1, np.uint32)
ncell = C.c_int32(0)
f = lib.nm_debug_raster_tiles
f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
f(C.byref(cfg), K, _lib.ptr(state), cap, nc.ctypes.data, off.ctypes.data, 1 << 20, C.byref(ncell), _lib.stream_ptr(dev))
n = np.diff(off[:ncell.value + 1].astype(np.int64))
nz = n[n > 0]
print(f"{name} view {view}: cells {ncell.value}, non-empty {nz.size}, pairs {int(n.sum())}, mean {nz.mean():.1f}, p50/p90/p99/max "
      f"{np.percentile(nz, [50, 90, 99, 100]).tolist()}")
edges = [1, 2, 17, 65, 129, 257, 513, 1025, 2049, 4097, 1 << 30]
h, _ = np.histogram(n, bins=edges)
w = [int(n[(n >= a) & (n < b)].sum()) for a, b in zip(edges[:-1], edges[1:])]
w2 = [float((n[(n >= a) & (n < b)].astype(np.float64) ** 2).sum()) for a, b in zip(edges[:-1], edges[1:])]
for a, b, c, p, q in zip(edges[:-1], edges[1:], h, w, w2):
    print(f"  size [{a}, {b}): {int(c)} cells, {p} pairs, sum n^2 = {q:.3g}")
# groups of 4 consecutive cells (one workgroup of k_cell_sort each): the quadratic cost a workgroup gets
g = n[:(n.size // 4) * 4].reshape(-1, 4).astype(np.float64)
cost = (g ** 2).sum(1)
print("per-workgroup sum n^2 over its 4 cells: total %.3g, max %.3g, mean of the non-empty %.3g" % (cost.sum(), cost.max(), cost[cost > 0].mean()))
