"""Multi-frame BPTT epoch (finetune.py schedule) with and without render/sim stream overlap: time and result equality."""
import sys, time
sys.path.insert(0, ".")
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
from neuma_amd.train import DEFAULT_CFG, simulate_video, video_loss
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "metric"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
scene = synth.make_scene(name)
true = SceneRuntime(scene, dev)
for net in (true.elasticity, true.plasticity):
    for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
        lin.lora_B.data.mul_(30.0)
true.F0 = torch.diag(torch.tensor([1.2, 0.85, 1.0])).to(dev).repeat(true.N, 1, 1).contiguous()
gt = simulate_video(true, frames)
rt = SceneRuntime(scene, dev)
rt.F0 = true.F0.clone()
c = dict(DEFAULT_CFG, num_frames=frames, decay_steps=80)
views = list(range(rt.V))
res = {}
for overlap in (False, True, False, True):
    for rep in range(3):
        for p in rt.parameters(): p.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        L = video_loss(rt, gt, c, 1.0, views, overlap_render=overlap)
        L.backward()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    g = torch.cat([p.grad.reshape(-1) for p in rt.parameters()])
    print("overlap %s: %.2f ms per epoch of %d frames (%.2f ms/frame), loss %.6e |g| %.6e" % (overlap, 1e3 * dt, frames, 1e3 * dt / frames, float(L), float(g.norm())))
    res[overlap] = (float(L), g.clone())
print("loss diff", abs(res[True][0] - res[False][0]), "grad rel diff", float((res[True][1] - res[False][1]).norm() / res[False][1].norm()))
