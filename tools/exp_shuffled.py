"""Frame time of the metric workload with the particles in random order (as NeuMA's data preparation leaves them),
with and without the roll-out's automatic re-ordering."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
dev = torch.device("cuda", 0)
scene = synth.make_scene("metric")
rng = np.random.default_rng(0)
perm = rng.permutation(scene.x0.shape[0])
inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
scene.x0, scene.v0 = scene.x0[perm], scene.v0[perm]
scene.bind_idx = inv[scene.bind_idx]
for reorder in ("auto", False):
    rt = SceneRuntime(scene, dev)
    rt.sim_fused.reorder = reorder
    rt.make_ground_truth()
    for _ in range(3):
        for p in rt.parameters(): p.grad = None
        rt.frame()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        for p in rt.parameters(): p.grad = None
        rt.frame()
    torch.cuda.synchronize()
    print("shuffled particles, reorder=%s: %.2f ms/frame (perm %s)" % (reorder, 1e3 * (time.perf_counter() - t0) / 5,
          "on" if torch.is_tensor(rt.sim_fused._perm) else "off"))
