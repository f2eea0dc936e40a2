"""Cost of the sharded-substep machinery on one GPU (one-rank RCCL group): per-substep time of the per-operator roll-out,
unsharded vs sharded.  python tools/exp_shard_overhead.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist

from neuma_amd import synth
from neuma_amd.harness import SceneRuntime

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
npart = int(sys.argv[2]) if len(sys.argv) > 2 else None       # e.g. 12500: what one of 8 ranks would hold
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ["NEUMA_SHARD_FORCE"] = "1"
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
scene = synth.make_scene(name, override=dict(N=npart, K=1000) if npart else None)
fused = (sys.argv[3] if len(sys.argv) > 3 else "fused") == "fused"      # fused: the library-level loops (what bench.py runs)
for shard in (False, True, False, True):
    rt = SceneRuntime(scene, dev, fused=fused, shard_sim=shard)
    rw = rt.rows

    def fwd():
        with torch.no_grad():
            return rt.rollout(rt.x0[rw], rt.v0[rw], rt.C0[rw], rt.F0[rw])

    def fwdbwd():
        for p in rt.parameters():
            p.grad = None
        o = rt.rollout(rt.x0[rw], rt.v0[rw], rt.C0[rw], rt.F0[rw])
        (o[0].sum() + o[3].sum()).backward()

    out = []
    for fn in (fwd, fwdbwd):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out.append(1e6 * (time.perf_counter() - t0) / reps / rt.S)
    print(f"{name} N={rt.N} fused={fused} shard={shard}: {out[0]:.1f} us/substep fwd, {out[1]:.1f} us/substep fwd+bwd", flush=True)
dist.destroy_process_group()
