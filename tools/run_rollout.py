"""A few fused roll-outs (S substeps, forward + backward) of a workload and nothing else: the program rocprofv3 traces for the
per-substep timeline (tools/timeline.py).  python tools/run_rollout.py [workload] [reps] [particles]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda", 0)
npart = int(sys.argv[3]) if len(sys.argv) > 3 else None
rt = SceneRuntime(synth.make_scene(name, override=dict(N=npart, K=1000) if npart else None), dev)
g = torch.Generator().manual_seed(0)
wx = torch.randn(rt.N, 3, generator=g).to(dev)
wF = torch.randn(rt.N, 3, 3, generator=g).to(dev)
for it in range(reps):
    for p in rt.parameters():
        p.grad = None
    o = rt.rollout(*rt.start)
    ((o[0] * wx).sum() + (o[3] * wF).sum()).backward()
    torch.cuda.synchronize()
print("done")
