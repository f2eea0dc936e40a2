"""Gradient error against the fp64 oracle of the three compositing plans (whole tiles / parallel forward segments / front-to-back
forward with checkpoints) on the raster test scene.  python tools/exp_ckpt_precision.py"""
import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from test_gpu_bind_raster import _scene, _gpu_raster
from gpu_util import dev, rel_max, abs_max
from oracle import raster as orr
from neuma_amd import _lib
lib = _lib.lib()
for opaque in (False, True):
    s, means, cov, op, shs, _, _ = _scene(deg=0, K=1500, scale=(0.04, 0.12))
    if opaque:
        op = torch.full_like(op, 0.97)
    gw = torch.randn(3, s.image_height, s.image_width, generator=torch.Generator().manual_seed(6)).to(dev())
    oins = [t.double().requires_grad_(True) for t in (means, shs, op, cov)]
    sd = orr.Settings(*[(f.double() if torch.is_tensor(f) else f) for f in s])
    oimg, _ = orr.render(sd, oins[0], oins[3], oins[2], shs=oins[1])
    og = torch.autograd.grad((oimg * gw.cpu().double()).sum(), oins)

    def render(rast):
        ins = [t.to(dev()).requires_grad_(True) for t in (means, shs, op, cov)]
        img, _ = rast(means3D=ins[0], means2D=None, opacities=ins[2], shs=ins[1], cov3D_precomp=ins[3])
        return img.detach(), torch.autograd.grad((img * gw).sum(), ins)
    lib.nm_raster_set_split(0, 32, 1 << 21)
    for label, fl in (("whole", None), ("parallel fwd segments", 0), ("front-to-back + checkpoints", 1 << 20)):
        rast = _gpu_raster(s)
        if fl is not None:
            lib.nm_raster_set_hinted(fl, 32)
            render(rast)
        img, g = render(rast)
        print(f"opaque={opaque} {label:28s} image {abs_max(img, oimg):.2e}  grads vs oracle " + " ".join(f"{rel_max(a, b):.2e}" for a, b in zip(g, og)))
lib.nm_raster_set_split(512, 512, 1 << 21); lib.nm_raster_set_hinted(0, 256)
