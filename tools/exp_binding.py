"""Time the GPU binding construction at the metric sizes (K=200k Gaussians, N=100k particles)."""
import sys, time
sys.path.insert(0, ".")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
from neuma_amd.binding import build_bindings
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene("metric"), dev)
g = rt.gaussians
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    counts, inside, cols = build_bindings(g.get_xyz, g.get_covariance(), rt.x0, 0.95, 10)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print("K", g.get_xyz.shape[0], "N", rt.N, "time ms", 1e3 * (t1 - t0), "mean kept", float(counts.float().mean()), "mean inside", float(inside.float().mean()),
      "empty", int((counts == 0).sum()))
