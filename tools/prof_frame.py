"""Print the full per-kernel table (calls, total ms, avg us) of one profiled metric frame (library HIP-event profiler)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
lib = _lib.lib()
dev = torch.device("cuda", 0)
scene = synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else "metric")
rt = SceneRuntime(scene, dev)
rt.make_ground_truth()
for _ in range(3):
    for p in rt.parameters():
        p.grad = None
    rt.frame()
torch.cuda.synchronize()
lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
for p in rt.parameters():
    p.grad = None
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
for ms, calls, name in sorted(rows, reverse=True):
    print(f"{name:42s} {calls:5d} {ms:8.3f} ms {1e3 * ms / calls:8.2f} us  {100 * ms / tot:5.1f}%")
print("total", round(tot, 3), "ms")
