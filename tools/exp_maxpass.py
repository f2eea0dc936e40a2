import sys, os
sys.path.insert(0, ".")
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from bench import prof_table
lib = _lib.lib(); dev = torch.device("cuda", 0)
scene = synth.make_scene("metric", override=dict(K=1000))
rt = SceneRuntime(scene, dev, fused=True)
rt.sim_fused._cache_blocks = 0
for it in range(3):
    if it == 2:
        lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
    xx, v, Cc, F = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, rt.F0)]
    out = rt.rollout(xx, v, Cc, F)
    (out[0].sum() + out[3].sum()).backward()
    torch.cuda.synchronize()
lib.nm_prof_enable(0, None)
t = prof_table(lib)
print("NM_MAXPASS", os.environ.get("NM_MAXPASS"), {k: round(1e3*v[1]/v[0],1) for k, v in t.items() if k in ("k_p2g","k_g2p_bwd")})
