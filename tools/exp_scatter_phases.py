"""-DNM_PHASES build: per-workgroup phase cycles of wg_scatter in the LAST k_p2g / k_g2p_bwd launch of a roll-out."""
import os, subprocess, sys, ctypes as C
import numpy as np
sys.path.insert(0, ".")
src = "neuma_amd/csrc"
out = "/tmp/libneuma_phases.so"
if os.path.exists("tools/libneuma_phases.so"):      # prebuilt on the build host (make -C tools phases), shipped with the snapshot
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
rt = SceneRuntime(synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else "metric", override=dict(K=1000, **({"N": int(os.environ["NM_EXP_N"])} if os.environ.get("NM_EXP_N") else {}))), dev)
ms = (C.c_int * 4)()
if len(sys.argv) > 2 and sys.argv[2] == "bwd":      # the last wg_scatter launch is then k_g2p_bwd of substep 0
    for it in range(2):
        for p_ in rt.parameters():
            p_.grad = None
        o = rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)
        (o[0].sum() + o[3].sum()).backward()
        torch.cuda.synchronize()
with torch.no_grad():
    for it in range(0 if (len(sys.argv) > 2 and sys.argv[2] == "bwd") else 3):
        rt.rollout(rt.x0, rt.v0, rt.C0, rt.F0)      # last launch using wg_scatter = k_p2g of substep 20
        torch.cuda.synchronize()
        lib.nm_debug_markslow(ms)
        print("after rollout", it, "cumulative mark_block calls / flag misses / new blocks / carried:", list(ms)[:4])
print("active blocks", rt.model.grid_stats())
fn = lib.nm_debug_scatter; fn.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(8 * 4096, dtype=np.int64)
print("rc", fn(buf.ctypes.data, 8 * 4096))
b = buf.reshape(4096, 8)
b = b[b[:, :7].sum(1) > 0]      # (workgroups that ran: the chunk size follows the scatter mode)
print('workgroups', len(b))
names = ["bbox", "box select", "sort/scan | zero", "contrib write | atomics", "cell sums | addr+mark", "9 pushes", "flush(+mark)", "passes"]
for i, nm in enumerate(names):
    print(f"{nm:14s} mean {b[:, i].mean():9.1f} median {np.median(b[:, i]):9.1f} max {b[:, i].max():9.0f}")
print("total cycles mean", b[:, :7].sum(1).mean(), "max", b[:, :7].sum(1).max())
print("passes histogram", np.bincount(b[:, 7].astype(int)))
tot = b[:, :7].sum(1)
for w in np.argsort(-tot)[:6]:
    print("wg", int(w), "total", int(tot[w]), "phases", b[w, :7].astype(int).tolist(), "mode/passes", int(b[w, 7]))
