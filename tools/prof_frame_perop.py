import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
lib = _lib.lib()
dev = torch.device("cuda", 0)
rt = SceneRuntime(synth.make_scene("metric"), dev, fused=False)
rt.make_ground_truth()
for _ in range(3):
    for p in rt.parameters(): p.grad = None
    rt.frame()
torch.cuda.synchronize()
lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
for p in rt.parameters(): p.grad = None
rt.frame()
torch.cuda.synchronize()
lib.nm_prof_enable(0, None)
buf = C.create_string_buffer(1 << 16)
lib.nm_prof_report(buf, len(buf))
rows = []
for line in buf.value.decode().splitlines():
    name, calls, ms = line.rsplit(" ", 2)
    rows.append((float(ms), int(calls), name))
tot = sum(r[0] for r in rows)
for ms, calls, name in sorted(rows, reverse=True)[:22]:
    print(f"{name:42s} {calls:5d} {ms:8.3f} ms {1e3 * ms / calls:8.2f} us")
print("total", round(tot, 3))
