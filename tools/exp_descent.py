"""End-to-end first-order check of the BPTT gradient in regimes with different signal levels (central differences)."""
import sys
sys.path.insert(0, ".")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
from neuma_amd.train import DEFAULT_CFG, simulate_video, video_loss
dev = torch.device("cuda", 0)
scene = synth.make_scene("tiny", override=dict(S=10, V=2))
for stretch, mult in (((1.12, 0.92, 1.0), 10.0), ((1.3, 0.75, 1.0), 10.0), ((1.3, 0.75, 1.0), 40.0), ((1.5, 0.6, 1.1), 40.0)):
    torch.manual_seed(0)
    F0 = torch.diag(torch.tensor(stretch)).to(dev)
    true = SceneRuntime(scene, dev, fused=True)
    true.F0 = F0.repeat(true.N, 1, 1).contiguous()
    for net in (true.elasticity, true.plasticity):
        for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
            lin.lora_B.data.mul_(mult)
    gt = simulate_video(true, 3, deform_cov=False)
    torch.manual_seed(0)
    rt = SceneRuntime(scene, dev, fused=True)
    rt.F0 = true.F0.clone()
    c = dict(DEFAULT_CFG, num_frames=3, decay_steps=2)
    params = rt.parameters()
    L0 = video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=False)
    grads = torch.autograd.grad(L0, params)
    g2 = sum(float((g.double() ** 2).sum()) for g in grads)
    def loss_at(step):
        with torch.no_grad():
            for p, g in zip(params, grads): p.add_(step * g)
            L = float(video_loss(rt, gt, c, 1.0, [0, 1], deform_cov=False))
            for p, g in zip(params, grads): p.sub_(step * g)
        return L
    for frac in (0.05, 0.01, 0.002):
        eps = frac * float(L0) / g2
        print("stretch %s mult %g: L0 %.3e frac %g central ratio %.3f" % (stretch, mult, float(L0), frac, (loss_at(eps) - loss_at(-eps)) / (2 * eps * g2)))
