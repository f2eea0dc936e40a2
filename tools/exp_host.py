"""Host-side enqueue time of one frame (forward + backward) against its GPU time.
    python tools/exp_host.py [workload]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
name = sys.argv[1] if len(sys.argv) > 1 else "bb"
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
rt.make_ground_truth()
for _ in range(10):
    rt.frame()
torch.cuda.synchronize()
host, gpu = [], []
for _ in range(50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); e0.record()
    rt.frame()
    t1 = time.perf_counter(); e1.record()
    torch.cuda.synchronize()
    host.append(1e3 * (t1 - t0)); gpu.append(e0.elapsed_time(e1))
host.sort(); gpu.sort()
print(f"{name}: host enqueue median {host[25]:.3f} ms, GPU span median {gpu[25]:.3f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    rt.frame()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
