import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from test_gpu_bind_raster import _scene, _gpu_raster
from gpu_util import dev
for (W, H) in ((128, 100), (120, 96), (125, 90)):
    s, means, cov, op, shs, _, _ = _scene(K=3000, W=W, H=H, deg=0, spread=0.6, scale=(0.03, 0.1))
    rast = _gpu_raster(s)
    ins = [t.to(dev()).requires_grad_(True) for t in (means, shs, op, cov)]
    for rep in range(3):
        img, radii = rast(means3D=ins[0], means2D=None, opacities=ins[2], shs=ins[1], cov3D_precomp=ins[3])
        torch.cuda.synchronize(); print(W, H, rep, "fwd ok", flush=True)
        img.sum().backward()
        torch.cuda.synchronize(); print(W, H, rep, "bwd ok", flush=True)
