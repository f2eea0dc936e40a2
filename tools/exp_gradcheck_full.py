"""Fused vs per-operator gradients at full size, rest vs deformed start.  python tools/exp_gradcheck_full.py [workload] [S]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
rt.S = rt.sim_fused.substeps = S


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))


g = torch.Generator().manual_seed(4)
for label, F0, wts in (("rest, sum-loss", rt.F0, None),
                       ("deformed, random weights", (torch.eye(3) + 0.05 * torch.randn(rt.N, 3, 3, generator=g)).to(dev), True)):
    if wts:
        wts = [torch.randn(s, generator=g).to(dev) for s in ((rt.N, 3), (rt.N, 3), (rt.N, 3, 3), (rt.N, 3, 3))]
    res = {}
    for rep in range(2):
        for fused in (True, False):
            rt.fused = fused
            for p in rt.parameters():
                p.grad = None
            ins = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, F0)]
            out = rt.rollout(*ins)
            loss = (out[0].sum() + (out[3] ** 2).sum()) if not wts else sum((o * w).sum() for o, w in zip(out, wts))
            loss.backward()
            res[(fused, rep)] = [t.grad.clone() for t in ins + rt.parameters()]
    names = ["x", "v", "C", "F"] + [f"p{i}" for i in range(len(rt.parameters()))]
    print(f"== {label}: fused vs per-op | fused vs fused (run-to-run) | per-op vs per-op")
    for i, nm in enumerate(names):
        print(f"   {nm:4s} {rel(res[(True, 0)][i], res[(False, 0)][i]):.2e}   {rel(res[(True, 0)][i], res[(True, 1)][i]):.2e}   "
              f"{rel(res[(False, 0)][i], res[(False, 1)][i]):.2e}   |g| {float(res[(False, 0)][i].abs().max()):.3e}")
