"""Run-to-run spread of ONE path: the same frame (same inputs, same kernels) executed R times - loss, final state and every
gradient compared with the first run.  The differences come from the order of fp32 atomics (p2g, the rasterizer's gradient
scatter) and from what they trigger downstream: a last-bit change of a Gaussian's position can move a pixel across one of the
rasterizer's cut-offs (alpha < 1/255, T < 1e-4), which is a jump of the loss, not a rounding error.
    python tools/exp_grad_noise.py [scene] [runs] [deformed]
deformed: start from F0 = diag(1.25, 0.8, 1) with the nets' output layers scaled by -4 (what tests/shard_worker.py does): at
F = I the invariants sigma - 1 and F^T F - I are differences of nearly equal numbers and the gradients inherit their rounding."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
rt = SceneRuntime(synth.make_scene(name), dev, fused=True)
rt.make_ground_truth()
if len(sys.argv) > 3 and sys.argv[3] == "deformed":
    rt.F0 = torch.diag(torch.tensor([1.25, 0.8, 1.0])).to(dev).repeat(rt.N, 1, 1).contiguous()
    with torch.no_grad():
        for p in rt.parameters():
            if p.shape[0] in (64, 9):
                p.mul_(-4.0)
rt.v0.requires_grad_(True)


def once():
    for p in list(rt.parameters()) + [rt.v0]:
        p.grad = None
    r = rt.frame()
    torch.cuda.synchronize()
    return float(r.loss), r.x.clone(), r.F.clone(), [p.grad.clone() for p in rt.parameters()], rt.v0.grad.clone() if rt.v0.grad is not None else None


def rel_max(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def rel_l2(a, b):
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


once()
ref = once()
worst = {"loss": 0.0, "x": 0.0, "F": 0.0, "grad max-norm": 0.0, "grad 2-norm": 0.0, "v0 grad max-norm": 0.0}
for i in range(R):
    cur = once()
    worst["loss"] = max(worst["loss"], abs(cur[0] - ref[0]) / abs(ref[0]))
    worst["x"] = max(worst["x"], rel_max(cur[1], ref[1]))
    worst["F"] = max(worst["F"], rel_max(cur[2], ref[2]))
    for a, b in zip(cur[3], ref[3]):
        worst["grad max-norm"] = max(worst["grad max-norm"], rel_max(a, b))
        worst["grad 2-norm"] = max(worst["grad 2-norm"], rel_l2(a, b))
    if cur[4] is not None:
        worst["v0 grad max-norm"] = max(worst["v0 grad max-norm"], rel_max(cur[4], ref[4]))
print(f"scene {name}: {R} repeats of the same frame against the first; worst relative difference")
for k, v in worst.items():
    print(f"  {k:20s} {v:.3e}")
