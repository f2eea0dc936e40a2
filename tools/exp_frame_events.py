"""Where a one-node frame's GPU time goes WITHOUT a tracer attached: HIP events recorded on the streams the work runs on,
next to the host's clock at the same points (is the host ahead of the device, or is the device waiting for launches?).
    python tools/exp_frame_events.py [workload] [frames]
Prints, per probe point, the mean device time since the frame's first launch and the mean host time since the call began."""
import sys, time
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth, harness, render
from neuma_amd import _lib as L
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
rt.make_ground_truth()
marks = []          # (label, event, host time) of the current frame
t_host0 = [0.0]


def mark(label):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream(dev))
    marks.append((label, ev, time.perf_counter() - t_host0[0]))


def wrap(owner, attr, label, count=[0]):
    f = getattr(owner, attr)

    def g(*a, **k):
        n = wrap.n[label] = wrap.n.get(label, 0) + 1
        mark(f"{label}#{n} begin")
        r = f(*a, **k)
        mark(f"{label}#{n} end")
        return r
    setattr(owner, attr, g)


wrap.n = {}
wrap(render, "raster_forward_raw", "view fwd")
wrap(render, "raster_backward_raw", "view bwd")
tf, tb = harness._tail_forward, harness._tail_backward


def tail_f(*a, **k):
    mark("roll-out fwd enqueued; tail begin")
    r = tf(*a, **k)
    mark("tail fwd end (main stream joined)")
    return r


def tail_b(*a, **k):
    mark("tail bwd begin")
    r = tb(*a, **k)
    mark("tail bwd end (B^T enqueued)")
    return r


harness._tail_forward, harness._tail_backward = tail_f, tail_b
acc_d, acc_h, cnt = defaultdict(float), defaultdict(float), 0
params = rt.parameters()
done = []
for it in range(frames + 10):
    for p in params:
        p.grad = None
    marks.clear(); wrap.n.clear()
    t_host0[0] = time.perf_counter()
    mark("frame begin")
    rt.frame()
    mark("frame end (reverse roll-out enqueued)")
    done.append(list(marks))
torch.cuda.synchronize()
for ms in done[10:]:
    e0 = ms[0][1]
    for label, ev, th in ms:
        acc_d[label] += e0.elapsed_time(ev) * 1e3
        acc_h[label] += th * 1e6
    cnt += 1
print(f"{name}: {cnt} frames; device us since the frame's first event | host us since the call began | host lead")
for label, _, _ in done[-1]:
    d, h = acc_d[label] / cnt, acc_h[label] / cnt
    print(f"  {label:44s} {d:9.1f} {h:9.1f} {d - h:9.1f}")
