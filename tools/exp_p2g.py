"""Experiment: p2g / g2p_bwd time vs particle ordering (same particles)."""
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from neuma_amd import synth, _lib
from neuma_amd.harness import SceneRuntime
from bench import prof_table

lib = _lib.lib()
dev = torch.device("cuda", 0)
scene = synth.make_scene("metric", override=dict(K=1000))
x = scene.x0
G = 128
cell = np.floor(x * G).astype(np.int64)
blk = cell // 4
cib = (cell[:, 0] % 4) * 16 + (cell[:, 1] % 4) * 4 + cell[:, 2] % 4
bkey = (blk[:, 0] * 4096 + blk[:, 1]) * 4096 + blk[:, 2]
# rank within cell
ckey = bkey * 64 + cib
o = np.argsort(ckey, kind="stable")
ck_sorted = ckey[o]
first = np.r_[True, ck_sorted[1:] != ck_sorted[:-1]]
start = np.maximum.accumulate(np.where(first, np.arange(len(o)), 0))
rank = np.empty(len(o), dtype=np.int64); rank[o] = np.arange(len(o)) - start
orders = {
    "stencil order (base-cell sorted, block-major)": synth.stencil_order(x, G),
    "cell-sorted (block-major)": np.argsort(ckey, kind="stable"),
    "interleaved (block, rank-in-cell, cell)": np.lexsort((cib, rank, bkey)),
    "random": np.random.default_rng(0).permutation(len(x)),
    "x-axis sort (reference `sort`)": np.argsort(-x[:, 1], kind="stable"),
}
for name, perm in orders.items():
    scene.x0 = x[perm]
    rt = SceneRuntime(scene, dev, fused=True)
    for it in range(3):
        if it == 2:
            lib.nm_prof_reset(); lib.nm_prof_enable(1, None)
        xx, v, Cc, F = [t.clone().requires_grad_(True) for t in (rt.x0, rt.v0, rt.C0, rt.F0)]
        out = rt.rollout(xx, v, Cc, F)
        (out[0].sum() + out[3].sum()).backward()
        torch.cuda.synchronize()
    lib.nm_prof_enable(0, None)
    t = prof_table(lib)
    print(f"== {name}")
    for k in ("k_p2g", "k_g2p_bwd", "k_g2p", "k_p2g_bwd", "k_grid_op", "k_clear", "k_material_fwd<NM_ELASTICITY>", "k_material_bwd<NM_ELASTICITY>", "k_wgrad_reduce"):
        if k in t:
            print(f"   {k:36s} calls {t[k][0]:4d} avg {1e3*t[k][1]/t[k][0]:8.1f} us")
