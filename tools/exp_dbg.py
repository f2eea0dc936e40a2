import sys, os
sys.path.insert(0, ".")
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from bench import prof_table
lib = _lib.lib(); dev = torch.device("cuda", 0)
scene = synth.make_scene("metric", override=dict(K=1000))
rt = SceneRuntime(scene, dev, fused=True)
for it in range(3):
    if it == 2:
        lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
    with torch.no_grad():
        rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
    torch.cuda.synchronize()
lib.nm_prof_enable(0, None)
t = prof_table(lib)
print("NM_DBG", os.environ.get("NM_DBG"), {k: round(1e3*v[1]/v[0],1) for k, v in t.items() if k in ("k_p2g","k_g2p","k_grid_op","k_clear")})
import ctypes as C, numpy as np
if os.environ.get("NM_DBG") and int(os.environ["NM_DBG"]) & 8:
    fn = lib.nm_mpm_debug_fetch
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    buf = np.zeros(4096 * 4, dtype=np.int64)
    print("fetch rc", fn(rt.model.handle(), buf.ctypes.data, 4096 * 4))
    cyc = buf[:4 * 196:4]
    print("total cycles: mean %.0f median %.0f p90 %.0f max %.0f" % (cyc.mean(), np.median(cyc), np.percentile(cyc, 90), cyc.max()))
    ph = buf[4096 * 2: 4096 * 2 + 196 * 8].reshape(196, 8)
    names = ["A stage", "B sort", "C accumulate", "C rmw", "flush", "-", "-", "single"]
    for i, nm in enumerate(names):
        print("  %-14s mean %9.0f median %9.0f max %9.0f" % (nm, ph[:, i].mean(), np.median(ph[:, i]), ph[:, i].max()))
