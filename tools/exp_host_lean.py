"""Where the host time of a one-node frame goes: wall time inside the library calls (kernel launches) against the Python around
them, forward and backward (the backward runs on the autograd engine's thread: cProfile does not see it).
    python tools/exp_host_lean.py [workload] [frames]"""
import sys, time
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth, _lib, harness, render
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "bb"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
rt.make_ground_truth()
acc = defaultdict(lambda: [0, 0.0])
lib = _lib.lib()


class TimedLib(object):
    def __init__(self, inner):
        self._inner = inner

    def __getattr__(self, k):
        f = getattr(self._inner, k)

        def g(*a):
            t0 = time.perf_counter()
            r = f(*a)
            e = acc["C " + k]
            e[0] += 1; e[1] += time.perf_counter() - t0
            return r
        setattr(self, k, g)
        return g


tl = TimedLib(lib)
_lib.lib = lambda: tl


def timed(mod, owner, attr, label):
    f = getattr(owner, attr)
    raw = f.__func__ if hasattr(f, "__func__") else f

    def g(*a, **k):
        t0 = time.perf_counter()
        r = raw(*a, **k)
        e = acc[label]
        e[0] += 1; e[1] += time.perf_counter() - t0
        return r
    setattr(owner, attr, staticmethod(g) if isinstance(owner, type) else g)


timed(None, harness._Frame, "forward", "py _Frame.forward (total)")
timed(None, harness._Frame, "backward", "py _Frame.backward (total)")
timed(None, harness, "_tail_forward", "py   _tail_forward")
timed(None, harness, "_tail_backward", "py   _tail_backward")
timed(None, render, "raster_forward_raw", "py     raster_forward_raw")
timed(None, render, "raster_backward_raw", "py     raster_backward_raw")
params = rt.parameters()
for it in range(frames + 20):
    if it == 20:
        torch.cuda.synchronize()
        acc.clear()
        t0 = time.perf_counter()
    for p in params:
        p.grad = None
    rt.frame()
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"{name}: {1e6 * total / frames:.1f} us per frame over {frames} frames (timers on)")
csum = 0.0
for k, (c, t) in sorted(acc.items(), key=lambda kv: kv[0]):
    print(f"  {k:44s} {c / frames:5.1f} per frame {1e6 * t / frames:8.1f} us per frame")
    if k.startswith("C "):
        csum += t
print(f"  inside library calls {1e6 * csum / frames:.1f} us per frame")
