"""-DNM_PHASES build: per-phase cycles of the LAST constitutive adjoint kernel of a fused roll-out's reverse sweep (the
elasticity adjoint of substep 0), with the activation cache on or off (NEUMA_ACT_CACHE).
    python tools/exp_phases_rollout.py"""
import os, subprocess, sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
src = "neuma_amd/csrc"
out = "/tmp/libneuma_phases.so"
if os.environ.get("NEUMA_PHASES_LIB"):
    out = os.path.abspath(os.environ["NEUMA_PHASES_LIB"])
elif os.path.exists("tools/libneuma_phases.so"):
    out = os.path.abspath("tools/libneuma_phases.so")
else:
  subprocess.run(f"/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DNM_PHASES -Iinclude -shared "
               f"{src}/nm_api.hip {src}/nm_mpm.hip {src}/nm_shard.hip {src}/nm_material.hip {src}/nm_bind.hip {src}/nm_bindbuild.hip {src}/nm_raster.hip {src}/nm_rollout.hip {src}/nm_rccl.hip -o {out}",
               shell=True, check=True)
os.environ["NEUMA_HIP_LIB"] = out
import torch
from neuma_amd import _lib, synth
from neuma_amd.harness import SceneRuntime
lib = _lib.lib()
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene("metric", override=dict(K=1000, S=int(os.environ.get("NM_EXP_S", "3")))), dev)
for _ in range(3):
    x = rt.x0.clone().requires_grad_(True)
    o = rt.rollout(x, rt.v0, rt.C0, rt.F0)
    (o[0].sum() + o[3].sum()).backward()
torch.cuda.synchronize()
fn = lib.nm_debug_phases; fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(3 * 8 * 2048, dtype=np.int64)
print("rc", fn(buf.ctypes.data, 3 * 8 * 2048), "act cache:", os.environ.get("NEUMA_ACT_CACHE", "auto"))
# phase marks: (2) closes behind the first-layer recompute of a tile and therefore also holds (f) + (e) of the tile before it
names = ["stage weights", "svd+feat+ybar", "(f,e) of prev tile + fwd / act", "(a) W2 grad", "(b) h2bar", "(c) W1 grad", "(d) h1bar", "last (f,e)+epilogue"]
for kind, what in ((0, "elasticity adjoint of substep 0: a launch of its own (cycles from the kernel's start)"),
                   (1, "plasticity adjoint of substep 0: the SECOND body of the last pair launch")):
    b = buf.reshape(3, 2048, 8)[kind][:893]
    b = b[b.sum(1) > 0]
    print(what, "-", len(b), "waves")
    for i, nm in enumerate(names):
        print(f"  {nm:32s} mean {b[:, i].mean():9.0f} median {np.median(b[:, i]):9.0f} max {b[:, i].max():9.0f}")
    print("  total mean", b.sum(1).mean(), "max", b.sum(1).max())

b = buf.reshape(3, 2048, 8)[2][:893]
b = b[b.sum(1) > 0]
print("forward pair kernel of the last substep boundary -", len(b), "waves")
for i, nm in enumerate(["stage both nets' weights", "g2p (both rounds)", "plasticity rounds", "elasticity rounds"]):
    print(f"  {nm:32s} mean {b[:, i].mean():9.0f} median {np.median(b[:, i]):9.0f} max {b[:, i].max():9.0f}")
print("  total mean", b.sum(1).mean(), "max", b.sum(1).max())
