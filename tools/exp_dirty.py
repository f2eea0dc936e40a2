"""Does a forward roll-out read memory it should not?  Freed device memory is filled with NaN / huge values first, then the same
roll-out runs several times (no-grad and with grad); results must agree run to run.   python tools/exp_dirty.py [scene]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
for fill in (float("nan"), 1e30):
    junk = [torch.full((64 << 20,), fill, device=dev) for _ in range(8)]
    del junk
    rt = SceneRuntime(synth.make_scene(name, override=dict(S=3, V=2)), dev, fused=True)
    rt.set_start_state("deformed")
    outs = []
    for it in range(4):
        with torch.no_grad():
            o = rt.rollout(*rt.start)
        outs.append([t.clone() for t in o])
    for it in range(2):
        o = rt.rollout(*[t.clone().requires_grad_(True) for t in rt.start])
        outs.append([t.detach().clone() for t in o])
    for k, o in enumerate(outs[1:], 1):
        d = [float((a - b).abs().max()) for a, b in zip(o, outs[0])]
        print(f"fill {fill}: run {k} vs run 0: max abs diff of (x, v, C, F) = {d}, finite {all(bool(torch.isfinite(t).all()) for t in o)}")
