"""Per-workgroup cycles of k_render stage 0 (library built with -DNM_FIXDBG).  python tools/exp_fixdbg.py [workload]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
lib = _lib.lib()
import os
lib.nm_raster_set_hinted(int(os.environ.get("FWD_LEN", "0")), 256)
NW = 8160 + 8192
with torch.no_grad():
    x, v, C_, F = rt.rollout(*rt.start)
    m3 = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
    dg = compute_bindings_F(F, rt.bindings)
    for _ in range(3):
        rt.render_view(m3, dg, 0)
    torch.cuda.synchronize()
    buf = torch.zeros(NW * 8, dtype=torch.int64, device=dev)
    lib.nm_debug_fix_buffer.argtypes = [C.c_void_p]
    lib.nm_debug_fix_buffer(buf.data_ptr())
    rt.render_view(m3, dg, 0)
    torch.cuda.synchronize()
    lib.nm_debug_fix_buffer(None)
d = buf.cpu().numpy().reshape(NW, 8)
t0 = d[:, 0][d[:, 0] > 0].min()
dur = d[:, 1] - d[:, 0]
beg = d[:, 0] - t0
end = d[:, 1] - t0
print("kernel span", int(end.max()), "cycles; workgroups", NW)
for nm, sl in (("whole tiles", slice(0, 8160)), ("segments", slice(8160, NW))):
    a = dur[sl]; busy = a > 20000
    print(f"{nm}: > 20k cycles: {int(busy.sum())}; of those mean {a[busy].mean():.0f} p50 {np.percentile(a[busy], 50):.0f} p99 {np.percentile(a[busy], 99):.0f} max {a.max()}; sum {a.sum():.3g}")
print("start time p50/p90/p99/max", np.percentile(beg, [50, 90, 99, 100]).tolist())
print("end time p50/p90/p99/max", np.percentile(end, [50, 90, 99, 100]).tolist())
o = np.argsort(-dur)[:12]
print("slowest:", [(int(i), int(dur[i]), int(beg[i])) for i in o])
h, e = np.histogram(end, bins=10)
print("end-time histogram", h.tolist(), [int(x) for x in e])
seg = d[8160:]
b = (seg[:, 1] - seg[:, 0]) > 20000
for nm, sl in (("scan", 2), ("gather", 3), ("loop (thread 0's wave)", 4), ("top-of-round wait", 5)):
    print(f"segments, {nm}: mean {seg[b, sl].mean():.0f} p50 {np.percentile(seg[b, sl], 50):.0f} p99 {np.percentile(seg[b, sl], 99):.0f}")
print("hits per segment: mean %.0f p50 %.0f p99 %.0f max %d" % (seg[b, 6].mean(), np.percentile(seg[b, 6], 50), np.percentile(seg[b, 6], 99), seg[b, 6].max()))
print("---- slowest workgroups of stage 0: (index, kind, cycles, scan, gather, loop, wait, hits)")
for i in np.argsort(-dur)[:16]:
    print((int(i), "whole" if i < 8160 else "item", int(dur[i]), int(d[i, 2]), int(d[i, 3]), int(d[i, 4]), int(d[i, 5]), int(d[i, 6])))
tot = dur[dur > 0].sum()
print("sum of workgroup cycles %.3g; whole %.3g, items %.3g; sum of hits %d" % (tot, dur[:8160].sum(), dur[8160:].sum(), int(d[:, 6].sum())))
