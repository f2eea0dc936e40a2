import os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from gpu_util import dev, rel_max
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
for kind in ("rest", "deformed"):
    rt = SceneRuntime(synth.make_scene("tiny"), dev(), fused=True)
    rt.set_start_state(kind)
    rt.make_ground_truth()
    res = {}
    for lean in ("1", "0", "1b", "0b"):
        os.environ["NEUMA_LEAN_FRAME"] = lean[0]
        for _ in range(2):
            for p in rt.parameters(): p.grad = None
            r = rt.frame()
        res[lean] = (r, [p.grad.clone() for p in rt.parameters()])
    for a, b in (("1", "0"), ("1", "1b"), ("0", "0b")):
        print(kind, a, b, "loss", float(res[a][0].loss), float(res[b][0].loss), ["%.1e" % rel_max(x, y) for x, y in zip(res[a][1], res[b][1])])
