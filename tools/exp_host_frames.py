"""Host time of consecutive SceneRuntime.frame() calls without any synchronisation in between (metric workload): does the host run
ahead of the device, or does something inside a frame wait for it?    python tools/exp_host_frames.py [frames]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene("metric"), dev, fused=True)
rt.make_ground_truth()
for _ in range(10):
    for p in rt.parameters():
        p.grad = None
    rt.frame()
torch.cuda.synchronize(dev)
t = [time.perf_counter()]
for _ in range(n):
    for p in rt.parameters():
        p.grad = None
    rt.frame()
    t.append(time.perf_counter())
torch.cuda.synchronize(dev)
t_end = time.perf_counter()
print("host ms per frame() call, no sync between calls:", " ".join(f"{1e3 * (b - a):.2f}" for a, b in zip(t, t[1:])))
print(f"all {n} frames incl. final sync: {1e3 * (t_end - t[0]):.2f} ms = {1e3 * (t_end - t[0]) / n:.3f} ms per frame")
