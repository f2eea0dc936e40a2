"""Host-side cost of one frame: cProfile of SceneRuntime.frame() on a scene whose GPU work is negligible (few Gaussians, few
particles), so that what is measured is Python / ctypes / launch overhead per frame.   python tools/exp_host_profile.py"""
import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else "metric", override=dict(N=4000, K=2000) if len(sys.argv) < 2 else None), dev)
rt.make_ground_truth()
for _ in range(5):
    for p in rt.parameters():
        p.grad = None
    rt.frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    for p in rt.parameters():
        p.grad = None
    rt.frame()
torch.cuda.synchronize()
print("host-bound frame time: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 20))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    for p in rt.parameters():
        p.grad = None
    rt.frame()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
