"""Hinted split compositing: per-kernel times of one view for several minimum segment lengths.
    python tools/exp_hint_sweep.py [workload] [minseg ...]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
segs = [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or [(512, 3072)]
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


for ms, fl in segs:
    lib.nm_raster_set_hinted(fl, ms)
    for _ in range(4):
        once()
    torch.cuda.synchronize()
    lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    lib.nm_prof_enable(0, None)
    buf = C.create_string_buffer(1 << 16)
    lib.nm_prof_report(buf, len(buf))
    t = {}
    for line in buf.value.decode().splitlines():
        nm, calls, ms_ = line.rsplit(" ", 2)
        t[nm] = 1e3 * float(ms_) / 5
    fwd = sum(v for k, v in t.items() if k in ("k_render", "k_render_fix", "k_render_sum", "k_split_plan", "k_hint_fill"))
    bwd = sum(v for k, v in t.items() if k.startswith("k_render_bwd"))
    print(f"min_seg {ms:5d} fwd_len {fl:6d}: forward composite {fwd:7.1f} us (render {t.get('k_render', 0):.0f} fix {t.get('k_render_fix', 0):.0f} sum "
          f"{t.get('k_render_sum', 0):.0f} plan {t.get('k_split_plan', 0):.0f} fill {t.get('k_hint_fill', 0):.0f})  reverse {bwd:7.1f} us")
