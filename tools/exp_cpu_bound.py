"""Where does the host spend a frame?  CPU wall time of each phase of SceneRuntime.frame() (no extra syncs), against the
GPU-side frame time."""
import sys, time
sys.path.insert(0, ".")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else "metric"), dev)
rt.make_ground_truth()
import os
if os.environ.get("NOCACHE"):
    rt.sim_fused._cache_blocks = 0
EV = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); EV.append((name, e))
def frame(T):
    mark("start")
    t = time.perf_counter()
    x, v, C, F = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
    T["rollout_fwd"] += time.perf_counter() - t; t = time.perf_counter()
    mark("rollout_fwd")
    means3D = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
    dg = compute_bindings_F(F, rt.bindings)
    T["bind"] += time.perf_counter() - t; t = time.perf_counter()
    mark("bind")
    loss = torch.zeros((), device=dev)
    for vi in range(rt.V):
        r = rt.render_view(means3D, dg, vi)
        T["render_fwd"] += time.perf_counter() - t; t = time.perf_counter()
        loss = loss + rt.pixel_loss(r, rt.gt[vi])
        T["loss"] += time.perf_counter() - t; t = time.perf_counter()
        mark("view%d" % vi)
    loss.backward()
    T["backward"] += time.perf_counter() - t
    mark("backward")
for _ in range(3):
    for p in rt.parameters(): p.grad = None
    frame({k: 0.0 for k in ("rollout_fwd", "bind", "render_fwd", "loss", "backward")})
torch.cuda.synchronize()
T = {k: 0.0 for k in ("rollout_fwd", "bind", "render_fwd", "loss", "backward")}
EV.clear()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    for p in rt.parameters(): p.grad = None
    frame(T)
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("per frame: cpu enqueue %.3f ms, wall %.3f ms" % (1e3 * t_cpu / N, 1e3 * t_all / N))
for k, v in T.items():
    print("  %-12s %.3f ms" % (k, 1e3 * v / N))

import collections
acc = collections.OrderedDict()
for (n0, e0), (n1, e1) in zip(EV[:-1], EV[1:]):
    key = n1 if n1 != "start" else "(next frame start)"
    acc[key] = acc.get(key, 0.0) + e0.elapsed_time(e1)
print("GPU timeline per phase (event to event), ms per frame:")
for k, v in acc.items():
    print("  %-20s %.3f" % (k, v / N))
