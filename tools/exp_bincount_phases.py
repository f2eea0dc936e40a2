"""-DNM_PHASES build: where k_bin_count's cycles go (per-wave clock64 phases summed over all waves of the renders of a few frames).
    python tools/exp_bincount_phases.py"""
import os, sys, ctypes as C
sys.path.insert(0, ".")
os.environ["NEUMA_HIP_LIB"] = os.path.abspath("tools/libneuma_phases.so")
import torch
from neuma_amd import _lib, synth
from neuma_amd.harness import SceneRuntime
lib = _lib.lib()
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene("metric"), dev)
rt.make_ground_truth()
for _ in range(3):
    rt.frame()
torch.cuda.synchronize()
fn = lib.nm_debug_bincount; fn.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 8)()
fn(buf, 1)
for _ in range(4):
    rt.frame()
torch.cuda.synchronize()
fn(buf, 0)
v = list(buf)
waves = v[7] if v[7] else 1
names = ["depth range scan", "rect + cull setup + prefix", "log reservation (1 atomic / workgroup) + barriers", "owner table",
         "rounds: operand fetch + tile loop + atomic issue", "rounds: previous round's log write (waits for ITS atomic)", "last log write"]
tot = sum(v[:7])
print("waves", waves, "launches", 12)
for n, x in zip(names, v[:7]):
    print(f"{n:52s} {x / waves:10.0f} cycles per wave  {100.0 * x / tot:5.1f} %")
print("total per wave", tot / waves)
