import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
from test_gpu_bind_raster import _scene, _gpu_raster
from gpu_util import dev, rel_max, abs_max
from oracle import raster as orr
from neuma_amd import _lib
lib = _lib.lib()
for seed in (7, 0, 5):
    g = torch.Generator().manual_seed(100 + seed)
    K = int(torch.randint(300, 4000, (1,), generator=g))
    W = int(torch.randint(40, 200, (1,), generator=g)); H = int(torch.randint(40, 160, (1,), generator=g))
    deg = int(torch.randint(0, 4, (1,), generator=g))
    lo = float(0.01 + 0.05 * torch.rand(1, generator=g)); hi = lo + float(0.02 + 0.15 * torch.rand(1, generator=g))
    s, means, cov, op, shs, _, _ = _scene(K=K, W=W, H=H, deg=deg, seed=seed, spread=float(0.2 + 0.5 * torch.rand(1, generator=g)), scale=(lo, hi))
    if seed % 2:
        op = torch.clamp(op * 3.0, max=0.995)
    if seed % 3 == 0:
        means = means.clone(); means[:5] = s.campos + 0.01
    gw = torch.randn(3, H, W, generator=g).to(dev())
    oins = [t.double().requires_grad_(True) for t in (means, shs, op, cov)]
    sd = orr.Settings(*[(f.double() if torch.is_tensor(f) else f) for f in s])
    oimg, _ = orr.render(sd, oins[0], oins[3], oins[2], shs=oins[1])
    og = torch.autograd.grad((oimg * gw.cpu().double()).sum(), oins)

    def render(rast):
        ins = [t.to(dev()).requires_grad_(True) for t in (means, shs, op, cov)]
        img, _ = rast(means3D=ins[0], means2D=None, opacities=ins[2], shs=ins[1], cov3D_precomp=ins[3])
        return img.detach(), torch.autograd.grad((img * gw).sum(), ins)
    lib.nm_raster_set_split(0, 32, 1 << 21)
    for ms in (16, 64, 256, 1024):
        for fl in (0, 1 << 20):
            rast = _gpu_raster(s)
            lib.nm_raster_set_hinted(fl, ms)
            errs = []
            for rep in range(6):
                img, gr = render(rast)
                errs.append(max(rel_max(a, b) for a, b in zip(gr, og)))
            print(f"seed {seed} minseg {ms:5d} fwd_len {fl:8d}: " + " ".join(f"{e:.1e}" for e in errs), flush=True)
lib.nm_raster_set_split(512, 512, 1 << 21); lib.nm_raster_set_hinted(0, 256)
