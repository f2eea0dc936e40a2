"""Build a -DNM_PHASES copy of the library and print per-phase cycles of k_material_fwd (debug experiment)."""
import os, subprocess, sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
src = "neuma_amd/csrc"
out = "/tmp/libneuma_phases.so"
subprocess.run(f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DNM_PHASES -Iinclude -shared "
               f"{src}/nm_api.hip {src}/nm_mpm.hip {src}/nm_shard.hip {src}/nm_material.hip {src}/nm_bind.hip {src}/nm_bindbuild.hip {src}/nm_raster.hip {src}/nm_rollout.hip {src}/nm_rccl.hip -o {out}",
               shell=True, check=True)
os.environ["NEUMA_HIP_LIB"] = out
import torch
from neuma_amd import _lib, synth
from neuma_amd.harness import SceneRuntime
lib = _lib.lib()
dev = torch.device("cuda", 0)
scene = synth.make_scene("metric", override=dict(K=1000))
rt = SceneRuntime(scene, dev)
F = (torch.eye(3, device=dev) + 0.02 * torch.randn(rt.N, 3, 3, device=dev)).contiguous()
import sys as _s
BWD = len(_s.argv) > 1 and _s.argv[1] == "bwd"
if BWD:
    Fg = F.clone().requires_grad_(True)
    for _ in range(3):
        out = rt.elasticity(Fg)
        g = torch.autograd.grad(out.sum(), [Fg] + rt.parameters()[:6])
    torch.cuda.synchronize()
else:
    with torch.no_grad():
        for _ in range(3):
            rt.elasticity(F)
    torch.cuda.synchronize()
fn = lib.nm_debug_phases; fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(8 * 2048, dtype=np.int64)
print("rc", fn(buf.ctypes.data, 8 * 2048))
b = buf.reshape(2048, 8)[:1024 if BWD else 1563]
names = ["stage weights", "svd+feat+ybar", "fwd recompute", "(a) W2 grad", "(b) h2bar", "(c) W1 grad", "(d) h1bar", "(e,f)+epilogue"] if BWD else ["stage weights", "svd+features", "mlp 4 tiles", "epilogue"]
for i, nm in enumerate(names):
    print(f"{nm:16s} mean {b[:, i].mean():9.0f} median {np.median(b[:, i]):9.0f} max {b[:, i].max():9.0f}")
print("total mean", b.sum(1).mean(), "max", b.sum(1).max())
