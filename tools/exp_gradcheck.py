"""Where does the end-to-end finite-difference / analytic gradient ratio deviate from 1?  (a) roll-out only: L = <w, x_S>
w.r.t. the LoRA parameters; (b) render only: pixel loss w.r.t. means3D along the analytic gradient direction."""
import sys
sys.path.insert(0, ".")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F
dev = torch.device("cuda", 0)
torch.manual_seed(0)
scene = synth.make_scene("tiny", override=dict(S=10, V=2))
rt = SceneRuntime(scene, dev, fused=True)
rt.F0 = torch.diag(torch.tensor([1.12, 0.92, 1.0])).to(dev).repeat(rt.N, 1, 1).contiguous()
params = rt.parameters()
w = torch.randn(rt.N, 3, device=dev)
def Lsim():
    x, v, C, F = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
    x, v, C, F = rt.rollout(x, v, C, F)
    return (x * w).sum() * 1e3
L0 = Lsim()
g = torch.autograd.grad(L0, params)
g2 = sum(float((a.double() ** 2).sum()) for a in g)
for frac in (1e-1, 1e-2, 1e-3):
    eps = frac * abs(float(L0)) / g2
    vals = []
    for sgn in (1, -1):
        with torch.no_grad():
            for p, a in zip(params, g): p.add_(sgn * eps * a)
            vals.append(float(Lsim()))
            for p, a in zip(params, g): p.sub_(sgn * eps * a)
    print("sim-only: frac %g ratio %.4f (L0 %.4e g2 %.3e)" % (frac, (vals[0] - vals[1]) / (2 * eps * g2), float(L0), g2))
# render only
rt.make_ground_truth()
with torch.no_grad():
    x, v, C, F = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
    m0 = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
    dg = compute_bindings_F(F, rt.bindings)
for use_dg in (True, False):
    def Lr(m):
        return sum(rt.pixel_loss(rt.render_view(m, dg if use_dg else None, vi), rt.gt[vi]) for vi in range(rt.V))
    m = m0.clone().requires_grad_(True)
    L0 = Lr(m)
    (gm,) = torch.autograd.grad(L0, m)
    g2 = float((gm.double() ** 2).sum())
    for frac in (1e-1, 1e-2, 1e-3, 1e-4):
        eps = frac * float(L0) / g2
        with torch.no_grad():
            lp, lm = float(Lr(m0 + eps * gm)), float(Lr(m0 - eps * gm))
        print("render-only (deform cov %s): frac %g ratio %.4f  max|step| %.2e (L0 %.3e)" % (use_dg, frac, (lp - lm) / (2 * eps * g2), float((eps * gm).abs().max()), float(L0)))
