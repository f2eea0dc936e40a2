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
