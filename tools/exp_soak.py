"""300 frames of the metric workload back to back: allocator footprint and loss must stay put (leak / drift check)."""
import sys, time
sys.path.insert(0, ".")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else "metric"), dev)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 301
rt.make_ground_truth()
t0 = time.perf_counter()
for i in range(N):
    for p in rt.parameters():
        p.grad = None
    r = rt.frame()
    if i % max(1, (N - 1) // 3) == 0:
        torch.cuda.synchronize()
        print(i, "alloc MB %.1f reserved MB %.1f loss %.6e t %.1fs" % (torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, float(r.loss), time.perf_counter() - t0), flush=True)
