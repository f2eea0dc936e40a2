"""Per-kernel times of ONE view's render forward + backward, alone on the GPU (no stream overlap).
    python tools/exp_raster.py [workload] [reps]"""
import ctypes as C
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
lib = _lib.lib()
with torch.no_grad():
    x, v, C_, F = rt.rollout(*rt.start)
    m3 = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
    dg = compute_bindings_F(F, rt.bindings)
gw = torch.randn(3, rt.scene.cfg["H"], rt.scene.cfg["W"], device=dev)


def once():
    m = m3.clone().requires_grad_(True)
    img = rt.render_view(m, dg, 0)
    (img * gw).sum().backward()


for _ in range(3):
    once()
torch.cuda.synchronize()
lib.nm_prof_reset()
lib.nm_prof_enable(1, None)
for _ in range(reps):
    once()
torch.cuda.synchronize()
lib.nm_prof_enable(0, None)
buf = C.create_string_buffer(1 << 16)
lib.nm_prof_report(buf, len(buf))
tot = 0.0
for line in buf.value.decode().splitlines():
    nm, calls, ms = line.rsplit(" ", 2)
    print(f"{nm:40s} {int(calls):5d} calls  {1e3 * float(ms) / int(calls):9.1f} us/call")
    tot += float(ms)
print(f"kernel total per render fwd+bwd: {1e3 * tot / reps:.1f} us")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    once()
b.record(); torch.cuda.synchronize()
print(f"wall per render fwd+bwd (incl. torch glue): {1e3 * a.elapsed_time(b) / reps:.1f} us")
os.environ["NM_RASTER_DEBUG"] = "1"
cam = rt.cameras[0]._nm_raster_cache[0][1]._cam
from neuma_amd import render as _r
_r.flush_pending()
once(); torch.cuda.synchronize()
st = _r._PENDING.entries[-1][1]
print("pairs binned", int(st[0]), " largest cell", int(st[1]) >> 32, " work items wanted", int(st[2]), " capacity", cam.bins.items)
_r._PENDING.entries.clear()

# ---- cell size distribution
import numpy as np
from neuma_amd.render import _raster_inputs, deform_cov_by_F
rast = rt.cameras[0]._nm_raster_cache[0][1]
cfg = rast._cam.cfg
cov = deform_cov_by_F(rt._cov, dg)
m3c, sh, cp, op, cv = _raster_inputs(m3, rt._shs, None, rt._opacity, cov)
K = m3c.size(0)
cap = 16 * K
cfg = cam.cfg
lib.nm_raster_state_bytes.restype = C.c_size_t
sb = int(lib.nm_raster_state_bytes(C.byref(cfg), K, cap))
state = torch.empty(sb, dtype=torch.uint8, device=dev)
radii = torch.empty(K, dtype=torch.int32, device=dev)
color = torch.empty(3, cfg.image_height, cfg.image_width, device=dev)
lib.nm_raster_forward(C.byref(cfg), K, sh.size(1), _lib.ptr(m3c), _lib.ptr(sh), None, _lib.ptr(op), _lib.ptr(cv), _lib.ptr(radii),
                      _lib.ptr(state), sb, cap, _lib.ptr(color), None, _lib.stream_ptr(dev))
cnt = np.zeros(1 << 20, np.uint32)
nc = C.c_int32(0)
f = lib.nm_debug_raster_cells
f.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
f(C.byref(cfg), K, _lib.ptr(state), cap, cnt.ctypes.data, 1 << 20, C.byref(nc), _lib.stream_ptr(dev))
c = cnt[:nc.value].astype(np.int64)
nz = c[c > 0]
print("cells", nc.value, "non-empty", len(nz), "pairs", int(c.sum()), "mean", nz.mean(), "p50/p90/p99/max", np.percentile(nz, [50, 90, 99]).tolist(), nz.max())
print("cells >512:", int((c > 512).sum()), " >256:", int((c > 256).sum()), " sum n^2 = %.3g" % float((c * c).sum()))
bins = c.reshape(-1, 256)
print("non-empty bins", int((bins.sum(1) > 0).sum()), "largest bin", int(bins.sum(1).max()))
