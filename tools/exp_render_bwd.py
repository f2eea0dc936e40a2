"""k_render_bwd alone (one view of the metric scene, nothing else in flight), optionally with an experiment variant of the
kernel compiled in (-DNM_RB_VARIANT=n).   python tools/exp_render_bwd.py [variant]"""
import os, subprocess, sys, ctypes as C
sys.path.insert(0, ".")
variant = sys.argv[1] if len(sys.argv) > 1 else None
extra = sys.argv[2] if len(sys.argv) > 2 else ""       # e.g. -DNM_RB_BATCH=64
if variant:
    src = "neuma_amd/csrc"
    out = f"/tmp/libneuma_rb{variant}.so"
    files = " ".join(f"{src}/{f}" for f in ("nm_api.hip", "nm_mpm.hip", "nm_shard.hip", "nm_material.hip", "nm_bind.hip", "nm_bindbuild.hip",
                                             "nm_raster.hip", "nm_rollout.hip", "nm_rccl.hip"))
    subprocess.run(f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DNM_RB_VARIANT={variant} {extra} -Iinclude "
                   f"-shared {files} -o {out}", shell=True, check=True)
    os.environ["NEUMA_HIP_LIB"] = out
import torch
from neuma_amd import _lib, synth
from neuma_amd.harness import SceneRuntime
from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F
lib = _lib.lib()
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene("metric"), dev)
rt.make_ground_truth()
with torch.no_grad():
    x, v, Cc, F = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
    m3 = compute_bindings_xyz(x, rt.x0, rt.gaussians.get_xyz, rt.bindings)
    dg = compute_bindings_F(F, rt.bindings)
def once():
    mm = m3.clone().requires_grad_(True)
    rt.pixel_loss(rt.render_view(mm, dg, 0), rt.gt[0]).backward()
for _ in range(3):
    once()
torch.cuda.synchronize()
lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
for _ in range(10):
    once()
torch.cuda.synchronize()
lib.nm_prof_enable(0, None)
buf = C.create_string_buffer(1 << 16)
lib.nm_prof_report(buf, len(buf))
for line in buf.value.decode().splitlines():
    name, calls, ms = line.rsplit(" ", 2)
    if "render" in name or "preprocess" in name or "emit" in name:
        print(f"variant {variant} {extra}: {name:36s} {1e3 * float(ms) / int(calls):8.1f} us")
