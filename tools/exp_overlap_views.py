import sys, time
sys.path.insert(0, "/root/repo")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene("metric"), dev)
rt.make_ground_truth()
for ov, ns in ((False, 2), (True, 2), (True, 3), (False, 2), (True, 2), (True, 3)):
    rt.overlap_views = ov; rt.num_view_streams = ns
    if hasattr(rt, '_view_streams'): del rt._view_streams
    for _ in range(3):
        for p in rt.parameters(): p.grad = None
        r = rt.frame()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        for p in rt.parameters(): p.grad = None
        r = rt.frame()
    torch.cuda.synchronize()
    g = torch.cat([p.grad.reshape(-1) for p in rt.parameters()])
    print("streams", ns, "overlap_views %s: %.3f ms/frame loss %.8e |g| %.8e" % (ov, 1e2 * (time.perf_counter() - t0), float(r.loss), float(g.norm())))
