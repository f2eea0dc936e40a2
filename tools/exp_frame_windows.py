"""GPU time of the phases of an (untraced) one-node frame from HIP events on the main stream: roll-out forward, render forward
(all views, streams joined), render reverse, roll-out reverse.    python tools/exp_frame_windows.py [workload] [frames]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth, harness
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
import os
_S = os.environ.get("EXP_SUBSTEPS")
rt = SceneRuntime(synth.make_scene(name, override=dict(S=int(_S)) if _S else None), dev)
rt.make_ground_truth()
marks = []


def ev():
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


tf, tb = harness._tail_forward, harness._tail_backward


def f(*a, **k):
    e0 = ev(); r = tf(*a, **k); e1 = ev()
    marks.append(("render fwd", e0, e1))
    return r


def b(*a, **k):
    e0 = ev(); r = tb(*a, **k); e1 = ev()
    marks.append(("render bwd", e0, e1))
    return r


harness._tail_forward, harness._tail_backward = f, b
params = rt.parameters()
fm = []
for it in range(frames + 10):
    for p in params:
        p.grad = None
    e0 = ev()
    rt.frame()
    e1 = ev()
    fm.append((e0, e1))
torch.cuda.synchronize()
fm, marks = fm[10:], marks[20:]
tot = sum(a.elapsed_time(b) for a, b in fm) / len(fm)
rf = [a.elapsed_time(b) for n, a, b in marks if n == "render fwd"]
rb = [a.elapsed_time(b) for n, a, b in marks if n == "render bwd"]
# roll-out forward = frame start -> render fwd start, roll-out reverse = render bwd end -> frame end
fwd = [fm[i][0].elapsed_time(marks[2 * i][1]) for i in range(len(fm))]
bwd = [marks[2 * i + 1][2].elapsed_time(fm[i][1]) for i in range(len(fm))]
mid = [marks[2 * i][2].elapsed_time(marks[2 * i + 1][1]) for i in range(len(fm))]
m = lambda v: sum(v) / len(v)
print(f"{name}: frame {tot:.3f} ms = roll-out fwd {m(fwd):.3f} + render fwd {m(rf):.3f} + between {m(mid):.3f} + render bwd {m(rb):.3f} + roll-out bwd {m(bwd):.3f}")
