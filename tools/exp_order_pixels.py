"""How many pixels of the ground-truth renders differ between the caller's Gaussian order and the runtime's spatial order
(rasterizer cut-off events: single pixels, one run in a few).   python tools/exp_order_pixels.py"""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from neuma_amd import synth
from neuma_amd.harness import SceneRuntime
dev = torch.device("cuda", 0)
# dirty the allocator like earlier tests of the file would
scene = synth.make_scene("tiny", override=dict(S=3, V=2))
for rep in range(3):
    res = {}
    for mode in ("given", "spatial"):
        os.environ["NEUMA_GAUSSIAN_ORDER"] = mode
        rt = SceneRuntime(scene, dev, fused=True)
        rt.set_start_state("deformed")
        rt.make_ground_truth()
        res[mode] = [g.clone() for g in rt.gt]
    for a, b in zip(res["given"], res["spatial"]):
        d = (a - b).abs()
        print(rep, "max", float(d.max()), "pixels > 1e-6:", int((d.amax(0) > 1e-6).sum()), "of", d[0].numel())
