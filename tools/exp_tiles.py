"""Per-tile work distribution of one view (forward walk length = last contributor, list length of the bin) and the
critical path a tile-per-workgroup schedule has.    python tools/exp_tiles.py [workload] [view]"""
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
if len(sys.argv) > 3:
    rt.set_start_state(sys.argv[3])
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
gx, gy = (W + 15) // 16, (H + 15) // 16
nbx, nby = (gx + 3) // 4, (gy + 3) // 4
img = np.zeros((gy * 16, gx * 16), np.int64)
img[:H, :W] = nc.reshape(H, W)
last = img.reshape(gy, 16, gx, 16).max(axis=(1, 3))               # per tile: position of its last contributor in the bin's list
binlen = (off[256 * np.arange(1, nbx * nby + 1)].astype(np.int64) - off[256 * np.arange(nbx * nby)].astype(np.int64)).reshape(nby, nbx)
tl = np.repeat(np.repeat(binlen, 4, 0), 4, 1)[:gy, :gx]
busy = last > 0
print("last contributor beyond the end of the tile's list:", int((last > tl).sum()), "tiles; max last", int(last.max()), "max list", int(tl.max()),
      "pairs binned vs capacity", int(off[ncell.value]), cap)
print(f"tiles {gx * gy}, busy (something composited) {int(busy.sum())}, with a non-empty list {int((tl > 0).sum())}")
q = [50, 90, 99, 100]
print("walk length (list positions up to the last contributor) of busy tiles: mean %.0f  p50/p90/p99/max %s" % (last[busy].mean(), np.percentile(last[busy], q).tolist()))
print("bin list length over busy tiles: mean %.0f  p50/p90/p99/max %s" % (tl[busy].mean(), np.percentile(tl[busy], q).tolist()))
print("sum of walk lengths %.3g (candidates tested), if spread evenly over 1536 resident workgroups: %.0f each; longest %d" %
      (last.sum(), last.sum() / 1536, last.max()))
h, e = np.histogram(last[busy], bins=[1, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 1 << 20])
print("histogram of walk lengths:", dict(zip([int(x) for x in e[:-1]], h.tolist())))
# per-pixel spread inside a tile: how much of the walk is done with most pixels already stopped
pl = img.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy, gx, 256)
med = np.median(pl, axis=2)
print("median pixel's last / tile's last over busy tiles: mean %.2f" % float((med[busy] / last[busy]).mean()))
# ---- walk record of the camera (nm_raster_forward_ex): how far the forward walk really went
walk = rt.cameras[view]._nm_raster_cache[0][1]._cam.tile_walk(dev)
if walk is not None:
    w = walk.cpu().numpy().astype(np.int64).reshape(gy, gx)
    nz = w > 0
    print("walk record: tiles with a walk %d; reached the end of their list %d; mean %.0f p50/p90/p99/max %s; sum %.3g" %
          (int(nz.sum()), int((nz & (w >= tl)).sum()), w[nz].mean(), np.percentile(w[nz], q).tolist(), w.sum()))
    r = w[busy] / np.maximum(last[busy], 1)
    print("walk / last contributor over busy tiles: p50/p90/p99 %s" % np.percentile(r, [50, 90, 99]).tolist())
    from neuma_amd.render import split_plan
    print("hinted plan (work items, segment):", split_plan(rast, m3, rt._opacity, shs=rt._shs, cov3D_precomp=cov, hinted=True))
