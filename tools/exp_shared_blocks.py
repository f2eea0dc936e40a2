"""How many 4x4x4-node grid blocks do the ranks of a particle-sharded run list, and how many do they share?
Host-side count on the synthetic scenes (initial state), for DESIGN.md section 6.  python tools/exp_shared_blocks.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from neuma_amd import synth  # noqa: E402


def blocks_of(x, G):
    base = np.maximum((x * G - 0.5).astype(np.int64), 0)
    nb = (G + 2 + 3) // 4
    out = set()
    for a in (0, 2):
        for b in (0, 2):
            for c in (0, 2):
                n = base + np.array([a, b, c])
                blk = ((n[:, 0] >> 2) * nb + (n[:, 1] >> 2)) * nb + (n[:, 2] >> 2)
                out.update(np.unique(blk).tolist())
    return out


for name in ("metric", "stress"):
    sc = synth.make_scene(name)
    N, G = sc.x0.shape[0], sc.cfg["G"]
    for world in (2, 4, 8):
        lists = [blocks_of(sc.x0[(N * r) // world:(N * (r + 1)) // world], G) for r in range(world)]
        count = {}
        for l in lists:
            for b in l:
                count[b] = count.get(b, 0) + 1
        shared = sum(1 for c in count.values() if c >= 2)
        print(f"{name:7s} N={N} G={G} world={world}: blocks/rank max {max(len(l) for l in lists)}, union {len(count)}, "
              f"shared {shared} ({shared} KiB per all-reduce)")
