"""Split compositing on / off for one view of a workload: per-kernel times and the plan the device chose.
    python tools/exp_split.py [workload] [busy_tiles] [min_segment]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from neuma_amd.render import split_plan, deform_cov_by_F
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F

name = sys.argv[1] if len(sys.argv) > 1 else "bb"
busy = int(sys.argv[2]) if len(sys.argv) > 2 else 512
minseg = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
lib = _lib.lib()
with torch.no_grad():
    x, v, C_, F = rt.rollout(*rt.start)
    m3 = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
    dg = compute_bindings_F(F, rt.bindings)
gw = torch.randn(3, rt.scene.cfg["H"], rt.scene.cfg["W"], device=dev)


def once():
    m = m3.clone().requires_grad_(True)
    img = rt.render_view(m, dg, 0)
    (img * gw).sum().backward()
    return img.detach()


for label, (b, s) in (("whole tiles", (0, 1024)), (f"split busy<{busy} minseg {minseg}", (busy, minseg))):
    lib.nm_raster_set_split(b, s, int(sys.argv[4]) if len(sys.argv) > 4 else (1 << 21))
    for _ in range(3):
        img = once()
    torch.cuda.synchronize()
    rast = rt.cameras[0]._nm_raster_cache[0][1]
    print(f"== {label}: plan (work items, segment) = {split_plan(rast, m3, rt._opacity, shs=rt._shs, cov3D_precomp=deform_cov_by_F(rt._cov, dg))}"
          f"  final_T<1e-3 on {float((img.mean(0) < 2).float().mean()):.2f}")
    lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    lib.nm_prof_enable(0, None)
    buf = C.create_string_buffer(1 << 16)
    lib.nm_prof_report(buf, len(buf))
    tot = 0.0
    for line in buf.value.decode().splitlines():
        nm, calls, ms = line.rsplit(" ", 2)
        if "render" in nm or "split" in nm:
            print(f"   {nm:36s} {1e3 * float(ms) / int(calls):9.1f} us/call")
        tot += float(ms)
    print(f"   kernel total per render fwd+bwd: {1e3 * tot / 5:.1f} us")
