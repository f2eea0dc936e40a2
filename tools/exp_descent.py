import sys; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
from neuma_amd.train import DEFAULT_CFG, simulate_video, video_loss
dev = torch.device("cuda", 0)
scene = synth.make_scene("tiny", override=dict(S=10, V=2))
F0 = torch.diag(torch.tensor([1.12, 0.92, 1.0])).to(dev)
true = SceneRuntime(scene, dev, fused=True); true.F0 = F0.repeat(true.N, 1, 1).contiguous()
for net in (true.elasticity, true.plasticity):
    for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
        lin.lora_B.data.mul_(10.0)
for frames in (1, 3):
    gt = simulate_video(true, frames, deform_cov=False)
    for fused in (True, False):
        rt = SceneRuntime(scene, dev, fused=fused); rt.F0 = true.F0.clone()
        c = dict(DEFAULT_CFG, num_frames=frames, decay_steps=2)
        params = rt.parameters()
        L0 = video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=False)
        grads = torch.autograd.grad(L0, params)
        g2 = sum(float((g.double() ** 2).sum()) for g in grads)
        base = [p.detach().clone() for p in params]
        for frac in (0.2, 0.02, 0.002):
            eps = frac * float(L0) / g2
            with torch.no_grad():
                for p, b, g in zip(params, base, grads):
                    p.copy_(b - eps * g)
                L1 = video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=False)
            print(f"frames {frames} fused {fused} frac {frac}: L0 {float(L0):.4e} dec {float(L0)-float(L1):.4e} pred {eps*g2:.4e} ratio {(float(L0)-float(L1))/(eps*g2):.3f}")
