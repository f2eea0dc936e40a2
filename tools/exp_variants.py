"""Per-kernel times of the fused roll-out (forward + backward) for alternative builds of the library.
    NEUMA_HIP_LIB=variants/lib_x.so python tools/exp_variants.py [workload] [kernel-name filter ...]"""
import ctypes as C
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
filt = sys.argv[2:] or ["k_p2g", "k_g2p_bwd", "k_p2g_bwd", "k_material", "k_grid"]
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene(name), dev)
lib = _lib.lib()
g = torch.Generator().manual_seed(0)
wx, wF = torch.randn(rt.N, 3, generator=g).to(dev), torch.randn(rt.N, 3, 3, generator=g).to(dev)


def once():
    for p in rt.parameters():
        p.grad = None
    o = rt.rollout(*rt.start)
    ((o[0] * wx).sum() + (o[3] * wF).sum()).backward()


for _ in range(3):
    once()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    once()
b.record(); torch.cuda.synchronize()
wall = a.elapsed_time(b) / 10
lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
for _ in range(3):
    once()
torch.cuda.synchronize()
lib.nm_prof_enable(0, None)
buf = C.create_string_buffer(1 << 16)
lib.nm_prof_report(buf, len(buf))
out = []
for line in buf.value.decode().splitlines():
    nm, calls, ms = line.rsplit(" ", 2)
    if any(f in nm for f in filt):
        out.append(f"{nm.strip('()')[:28]} {1e3 * float(ms) / int(calls):.1f}")
print(f"{os.environ.get('NEUMA_HIP_LIB', 'default')}: roll-out fwd+bwd {1e3 * wall / rt.S:.1f} us/substep | " + " | ".join(out))
