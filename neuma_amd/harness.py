"""Frame-level driver: the caller side of the hot path.

Reproduces the operator order of /root/reference/experiments/finetune.py:331-414 (finetune_constitutive inner
loop) and render.py:304-332 for one video frame:
    S x [ stress = E(F); x,v,C,F = sim(statics, it, x,v,C,F, stress); F = P(F) ]            (finetune.py:362-364)
    de_x = denormalize(x); means3D = g_prev + B (de_x - de_x_prev); F_k = B F               (373-376)
    for view: render = diff_rasterization(...); loss += decay * pixel_loss(render, gt)      (378-389)
    loss.backward()                                                                        (413-414)
Multi-GPU (torch.distributed over RCCL): the V views x tile rows of a frame are split into contiguous stripes, one
per rank; each rank back-propagates its stripes to dL/dmeans3D and ONE all-reduce (sum, K x 3 fp32) per frame
merges them before the binding transpose.  The simulation is either replicated (default: a 100k-particle substep is
shorter than one xGMI collective) or, with shard_sim=True, particle-sharded (sim/shard.py): each rank
steps its contiguous range of the particle list and the ranks all-reduce the grid blocks their ranges share, twice
per substep; positions and deformation gradients are all-gathered once per frame for the bindings, and the LoRA
gradients are summed over the ranks after the backward pass.
"""
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import synth
from .material import InvariantFullMetaElasticity, InvariantFullMetaPlasticity
from .render.gaussian_model import GaussianModel
from .rollout import MPMFusedDiffSim
from .sim import MPMModelBuilder, MPMCacheDiffSim, MPMStatics
from .tune import Bindings, compute_bindings_xyz, compute_bindings_F, diff_rasterization, l1_loss, l2_loss, pixel_loss_rows

PIXEL_LOSSES = {"l1": l1_loss, "l2": l2_loss}


def stripe_plan(num_views: int, tile_rows: int, world: int, rank: int, weights=None, snap: float = 0.15) -> List[Tuple[int, int, int]]:
    """Split the V * tile_rows work units of a frame (a unit = one 16-pixel tile row of one view) into `world` contiguous
    chunks; return this rank's (view, row0, row1) stripes.  Pure host logic (covered by the gloo CPU tests).

    weights: None (every unit costs the same) or V x tile_rows non-negative costs - the compositing work each tile row had in
    the previous frame (sum of its tiles' list walks, from the cameras' walk records).  Cuts are placed where the running
    cost passes r / world of the total, then moved to a view boundary if one lies within `snap` of a rank's share: a rank
    that renders a whole view pays that view's preprocessing and binning once, a cut through a view makes two ranks pay
    it.  Same inputs -> same plan on every rank."""
    total = num_views * tile_rows
    if weights is None:
        cuts = [(total * r) // world for r in range(world + 1)]
    else:
        w = np.asarray(weights, dtype=np.float64).reshape(-1)
        if w.shape[0] != total:
            raise ValueError(f"weights must hold {num_views} x {tile_rows} entries")
        w = np.maximum(w, 0.0) + 1e-3 * max(float(w.sum()), 1.0) / total          # (empty rows still cost a launch's worth)
        acc = np.concatenate([[0.0], np.cumsum(w)])
        share = acc[-1] / world
        cuts = [0]
        for r in range(1, world):
            c = int(np.searchsorted(acc, r * share, side="left"))
            if c > 0 and r * share - acc[c - 1] < acc[min(c, total)] - r * share:
                c -= 1                                           # the cut whose running cost is nearest the target
            c = min(max(c, cuts[-1]), total)
            # snap to the nearest view boundary if that moves no more than snap * share of work across the cut
            for b in (round(c / tile_rows) * tile_rows,):
                if 0 < b < total and b >= cuts[-1] and abs(acc[b] - acc[c]) <= snap * share:
                    c = int(b)
            # every rank keeps at least one unit (one tile row that holds more than a rank's share of the work makes
            # consecutive cuts coincide otherwise) - as long as there are at least `world` units at all
            if total >= world:
                c = min(max(c, cuts[-1] + 1), total - (world - r))
            cuts.append(c)
        cuts.append(total)
    lo, hi = cuts[rank], cuts[rank + 1]
    out = []
    for v in range(num_views):
        a, b = max(lo, v * tile_rows), min(hi, (v + 1) * tile_rows)
        if b > a:
            out.append((v, a - v * tile_rows, b - v * tile_rows))
    return out


class _AllReduceSum(torch.autograd.Function):
    """Identity in forward; all-reduce(sum) of the gradient in backward — merges per-rank dL/dmeans3D."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from .sim.shard import all_reduce_sum_
        g = g.contiguous()
        if not g.is_cuda:
            g = g.clone()
        all_reduce_sum_(g, ctx.group)
        return g, None


def merge_grad_across_ranks(x: torch.Tensor, group=None) -> torch.Tensor:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    return _AllReduceSum.apply(x, group)


class _FrameTail(torch.autograd.Function):
    """Everything of a frame behind the roll-out as ONE autograd node: binding (means3D = g_prev + B (de_x - de_x_prev),
    cov' = (B F) cov (B F)^T, tune/utils.py:424-472 + simulation_utils.py:25-48 in one launch), then per render job
    rasterizer forward + pixel loss (value and dL/dimage in one pass), finetune.py:367-389; the reverse sweep runs the
    rasterizer adjoints, sums dL/dmeans3D over the jobs (and the ranks) and applies B^T.  Counterpart, for the frame's
    render half, of what nm_rollout_* is for its substeps: per frame it replaces two binding nodes and per view a rasterizer
    node, a loss node and the additions between them - on the single-substep configurations the host time of those nodes
    was the frame time.  Same kernels, same results as the per-operator composition (SceneRuntime(fused_tail=False))."""

    @staticmethod
    def forward(ctx, rt, de_x, F, weight, jobs, streams):
        p_cur = de_x.detach().float().contiguous()
        Fc = F.detach().float().reshape(-1, 9).contiguous()
        loss, recs, grads, keep = _tail_forward(rt, p_cur, Fc, weight, jobs, streams)
        ctx.rt, ctx.recs, ctx.grads, ctx.streams = rt, recs, grads, streams
        ctx.keep = keep                    # (referenced by the records' raw pointers)
        return loss

    @staticmethod
    def backward(ctx, g):
        rt = ctx.rt
        dx = torch.empty(rt.bindings.N, 3, dtype=torch.float32, device=rt.device)
        _tail_backward(rt, ctx.recs, ctx.grads, ctx.streams, dx)
        ctx.recs = ctx.grads = ctx.keep = None
        return None, dx * g, None, None, None, None


def _tail_forward(rt, p_cur, Fc, weight, jobs, streams, gt=None, step=None, eager=None):
    """Binding + covariance push-forward (one launch), then per render job rasterizer forward + pixel loss (value and dL/dimage
    in one pass).  p_cur (N, 3), Fc (N, 9): contiguous fp32, outside autograd.  gt / step: the ground-truth images and the
    dataset step of THIS frame (multi-frame epochs; default: the runtime's single frame).  Returns (loss, records, dL/dimage
    per job, tensors the records point into).
    eager: a list = the caller WILL run the reverse sweep of exactly this forward pass with dL/dloss = const (SceneRuntime.frame:
    forward and reverse are two calls back to back): dL/dimage of a job depends on nothing but that job's image, so its
    rasterizer adjoint is enqueued right behind its loss, on the job's own stream - no join of the streams between the sweeps
    (two event round trips and the loss sum were ~70 us of an otherwise idle device per frame) and the adjoint of the job that
    finishes first fills the chip while the others' forward tails drain.  The jobs' dL/dmeans3D are appended to the list."""
    from . import _lib as L
    from .render import get_rasterizer, raster_forward_raw, raster_backward_raw
    lib, dev = L.lib(), p_cur.device
    b = rt.bindings
    K = b.K
    means3D = torch.empty(K, 3, dtype=torch.float32, device=dev)
    cov = torch.empty(K, 6, dtype=torch.float32, device=dev)
    L.check(lib.nm_bind_frame(K, L.ptr(b.rowptr), L.ptr(b.col), L.ptr(b.val), L.ptr(p_cur), L.ptr(rt._de_x_prev), L.ptr(rt._g_prev),
                              L.ptr(Fc), L.ptr(rt._cov6), L.ptr(means3D), L.ptr(cov), None, L.stream_ptr(dev)), "nm_bind_frame")
    loss = torch.zeros((), dtype=torch.float32, device=dev)
    kind = 0 if rt.pixel_loss is l1_loss else 1
    mask = getattr(rt, "force_mask_data", False)
    sh = None if mask else rt._shs
    cp = rt._ones_rgb(K) if mask else None
    main = torch.cuda.current_stream(dev) if streams else None
    recs, grads, parts = [], [], []

    def job(vi, rows, part):
        rast = get_rasterizer(rt.camera_at(vi) if step is None else rt.camera_at(vi, step), rt.gaussians.active_sh_degree, False,
                              rt.background, tile_rows=rows)
        img, _, rec = raster_forward_raw(rast._cam, means3D, sh, cp, rt._opacity, cov)
        h, w = int(img.shape[-2]), int(img.shape[-1])
        r0, r1 = (0, 0) if rows is None else (rows[0] * 16, min(h, rows[1] * 16))
        gimg = torch.empty_like(img)
        L.check(lib.nm_pixel_loss(kind, float(weight), h, w, r0, r1, L.ptr(img), L.ptr((rt.gt if gt is None else gt)[vi]), L.ptr(part), L.ptr(gimg),
                                  L.stream_ptr(dev)), "nm_pixel_loss")
        recs.append(rec); grads.append(gimg); parts.append(part)
        if eager is not None:
            eager.append(raster_backward_raw(rec, gimg)[0])

    inline_last = streams and eager is not None and os.environ.get("NEUMA_INLINE_LAST_VIEW", "1") != "0"
    for i, (vi, rows) in enumerate(jobs):
        if inline_last and i == len(jobs) - 1:
            # the last job stays on the caller's stream: the frame's critical path then crosses no queue at the fork, and none at the
            # join either unless one of the other jobs outlasts it (a hop between two HIP queues costs the device ~15 us)
            job(vi, rows, loss)
        elif streams:
            st = streams[i]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                job(vi, rows, loss)         # every job's workgroups add to the ONE loss word (zeroed on main in front of the fork)
                loss.record_stream(st)
        else:
            job(vi, rows, loss)
    if streams:
        if eager is None:                   # (eager: _tail_backward joins, behind whatever the caller enqueues on main meanwhile)
            for st in set(streams):
                main.wait_stream(st)
        for t in [means3D, cov] + grads + (eager or []):
            t.record_stream(main)
    return loss, recs, grads, (means3D, cov)


def _tail_backward(rt, recs, grads, streams, dx_out, outs=None, scale=1.0):
    """Rasterizer adjoints of every job, dL/dmeans3D summed over the jobs (and the ranks), then scale * B^T of the sum into
    dx_out (N, 3).  outs: the jobs' dL/dmeans3D if _tail_forward already enqueued the adjoints (eager; its join covers them)."""
    import torch.distributed as dist
    from . import _lib as L
    from .render import raster_backward_raw
    lib, dev = L.lib(), rt.device
    total = None
    b = rt.bindings
    multi = rt.world > 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size(rt.group) > 1
    if outs is not None:
        if streams:
            main = torch.cuda.current_stream(dev)
            for st in set(streams):
                main.wait_stream(st)
    elif streams:
        outs = []
        main = torch.cuda.current_stream(dev)
        for st, rec, gimg in zip(streams, recs, grads):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs.append(raster_backward_raw(rec, gimg)[0])
        for st in set(streams):
            main.wait_stream(st)
        for t in outs:
            t.record_stream(main)
    else:
        outs = [raster_backward_raw(rec, gimg)[0] for rec, gimg in zip(recs, grads)]
    if not multi and 1 <= len(outs) <= 4:       # one GPU: the sum of the jobs, B^T and the scale in ONE launch
        ins = [L.ptr(d) for d in outs] + [None] * (4 - len(outs))
        L.check(lib.nm_spmm_csr_sum3(b.N, L.ptr(b.t_rowptr), L.ptr(b.t_col), L.ptr(b.t_val), ins[0], ins[1], ins[2], ins[3],
                                     float(scale), L.ptr(dx_out), L.stream_ptr(dev)), "nm_spmm_csr_sum3")
        return
    for d in outs:
        total = d if total is None else total.add_(d)
    if total is None:       # a rank without render jobs (more ranks than tile rows): zeros, and it still joins the all-reduce
        total = torch.zeros(rt.bindings.K, 3, dtype=torch.float32, device=dev)
    if multi:
        from .sim.shard import all_reduce_sum_
        all_reduce_sum_(total, rt.group)        # the frame's one K x 3 all-reduce (ncclAllReduce on the library's communicator)
    L.check(lib.nm_spmm_csr_sum3(b.N, L.ptr(b.t_rowptr), L.ptr(b.t_col), L.ptr(b.t_val), L.ptr(total), None, None, None, float(scale),
                                 L.ptr(dx_out), L.stream_ptr(dev)), "nm_spmm_csr_sum3")


# nm_rollout_cfg.last_gF_zero of the frame / epoch drivers: the reverse sweep leaves out the last substep's plasticity adjoint (all
# zeros when only dL/dx flows into the last record).  NEUMA_LAST_GF_ZERO=0: launch it all the same (A/B, tests)
_LAST_GF_ZERO = 0 if os.environ.get("NEUMA_LAST_GF_ZERO", "1") == "0" else 1


def _status_words_out(rt, lib, gptr, cfg, status, ev, gcache):
    """The grid cache records' status words on their way to pinned memory behind the forward sweep, `ev` recorded behind them.
    The library writes them from a one-wave kernel straight into the pinned words (nm_rollout_cache_status): the strided
    device-to-host copy it replaces was a 20 us hole on the frame's stream between the roll-out and the frame's tail.
    (Round 5 also tried the copy on a stream of its own: neutral for the single frame, -5 % for the epoch - a fifth stream
    shares a hardware queue with one of the four the epoch already uses.)"""
    import ctypes as C
    from . import _lib as L
    L.check(lib.nm_rollout_cache_status(gptr, C.byref(cfg), C.c_void_p(status.data_ptr()), L.stream_ptr(rt.device)), "nm_rollout_cache_status")
    ev.record()


class _FrameState(object):
    """What the forward half of a one-node frame keeps for its reverse sweep."""
    __slots__ = ("recs", "grads", "streams", "keep", "states", "eff", "gcache", "svdc", "actc", "status", "ev", "cache_blocks",
                 "adj", "weight", "ws_token", "ws_ptr", "outs")


def _frame_static(rt):
    """ctypes arguments of a one-node frame that do not change from frame to frame (statics, the LoRA layers' job templates),
    rebuilt when one of the tensors behind them is replaced."""
    from . import _lib as L
    from . import rollout as R
    layers = rt._lora_layers
    st_t = (rt.statics.vol, rt.statics.rho, rt.statics.clip_bound, rt.statics.enabled)
    key = tuple(t.data_ptr() for t in st_t) + tuple(t.data_ptr() for l in layers for t in (l.weight, l.lora_B, l.lora_A))
    c = rt.__dict__.get("_frame_static_cache")
    if c is None or c[0] != key:
        fwd, bwd = (L.nm_lora_layer * 6)(), (L.nm_lora_layer * 6)()
        woff, goff, sizes = [], [], []
        wo = go = 0
        for i, l in enumerate(layers):
            W = l.weight
            fwd[i] = L.nm_lora_layer(W.shape[0], W.shape[1], l.r, float(l.scaling), L.ptr(W), L.ptr(l.lora_B), L.ptr(l.lora_A), None, None)
            bwd[i] = L.nm_lora_layer(W.shape[0], W.shape[1], l.r, float(l.scaling), None, L.ptr(l.lora_B), L.ptr(l.lora_A), None, None)
            woff.append(wo); goff.append(go)
            sizes += [l.lora_B.numel(), l.lora_A.numel()]
            wo += W.numel()
            go += l.lora_B.numel() + l.lora_A.numel()
        shapes = [tuple(t.shape) for l in layers for t in (l.lora_B, l.lora_A)]
        c = rt._frame_static_cache = (key, rt.statics.c_struct(), fwd, bwd, woff, goff, sizes, shapes, go, sum(R._WSZ))
    return c


def _frame_forward(rt, weight, jobs, streams, eager=False):
    """Forward half of the whole frame of a one-GPU runtime, outside autograd: effective weights of both nets (one launch), the
    S-substep roll-out (nm_rollout_forward), binding + covariance push-forward, every render job + loss (finetune.py:331-389).
    Returns (loss, x, F (N, 9), _FrameState)."""
    import ctypes as C
    from . import _lib as L
    from . import rollout as R
    lib, dev = L.lib(), rt.device
    sim = rt.sim_fused
    n, S = rt.n_local, int(sim.substeps)
    stream = L.stream_ptr(dev)
    _, st, jf, _jb, woff, _goff, _sizes, _shapes, _gtot, nw = _frame_static(rt)
    # effective weights e0 | e1 | e2 | p0 | p1 | p2
    eff = torch.empty(2 * nw, dtype=torch.float32, device=dev)
    base = eff.data_ptr()
    for i in range(6):
        jf[i].o0 = base + 4 * woff[i]
    L.check(lib.nm_lora_merge_layers(6, jf, stream), "nm_lora_merge_layers")
    # roll-out: record 0 = the (packed) start state
    states = torch.empty(S + 1, 33 * n, dtype=torch.float32, device=dev)
    states[0, :24 * n].copy_(rt._start_packed())
    sz = rt.__dict__.get("_roll_sizes")
    cache_blocks = int(sim.grid_cache_blocks())
    if sz is None or sz[0] != (n, S, cache_blocks):
        sz = rt._roll_sizes = ((n, S, cache_blocks), int(lib.nm_rollout_workspace(n, S)),
                               int(lib.nm_rollout_gridcache_bytes(S, cache_blocks)) if cache_blocks > 0 else 0,
                               int(lib.nm_rollout_svdcache_bytes(n, S)), int(lib.nm_rollout_actcache_bytes(n, S)))
    _, ws_bytes, gc_bytes, svd_bytes, act_bytes = sz
    ws = rt._scratch("ws", ws_bytes)
    gcache = torch.empty(gc_bytes, dtype=torch.uint8, device=dev) if gc_bytes > 0 else None
    svdc = R.lease_cache(svd_bytes, dev) if R._SVD_CACHE else None
    actc = R.lease_cache(act_bytes, dev, force=R._ACT_CACHE == '1') if R._ACT_CACHE != '0' else None
    adj = L.SVD_ADJOINT[sim.svd_adjoint]
    cfg = L.nm_rollout_cfg(S, float(sim.plasticity.alpha), cache_blocks if gcache is not None else 0, 0, adj,
                           svdc.t.data_ptr() if svdc is not None else None, actc.t.data_ptr() if actc is not None else None, 0,
                           _LAST_GF_ZERO)      # (the reverse sweep will say the same: see _frame_backward)
    w0, w1 = R._WSZ[0], R._WSZ[0] + R._WSZ[1]
    mle = L.nm_mlp(base, base + 4 * w0, base + 4 * w1)
    pb = base + 4 * nw
    mlp = L.nm_mlp(pb, pb + 4 * w0, pb + 4 * w1)
    gptr = gcache.data_ptr() if gcache is not None else None
    L.check(lib.nm_rollout_forward(rt.model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp), states.data_ptr(),
                                   gptr, ws.data_ptr(), ws_bytes, stream), "nm_rollout_forward")
    fs = _FrameState()
    fs.ws_token = rt._ws_token = rt.__dict__.get("_ws_token", 0) + 1      # (whose operand-order weights the workspace holds)
    fs.ws_ptr = ws.data_ptr()
    fs.status = fs.ev = None
    if gcache is not None and R._CACHE_STATUS:
        pool = rt.__dict__.setdefault("_status_pool", [])      # (pinned words + event: handed back by the backward pass)
        if pool and pool[-1][0].numel() == S:
            fs.status, fs.ev = pool.pop()
        else:
            fs.status, fs.ev = torch.empty(S, dtype=torch.int32, pin_memory=True), torch.cuda.Event()
        _status_words_out(rt, lib, gptr, cfg, fs.status, fs.ev, gcache)
    if sim._cache_blocks is None:      # first roll-out: size the grid cache from what the scene touches (one host sync)
        blocks, _ = rt.model.grid_stats()
        sim._cache_blocks = int(1.5 * blocks) + 64
    last = states[S]
    x, Fl = last[:3 * n].view(n, 3), last[15 * n:24 * n].view(n, 9)
    p_cur = x if rt._unit_frame() else ((x - rt.center) / rt.size).contiguous()      # finetune.py:373
    fs.outs = [] if eager else None      # eager: the rasterizer adjoints ride behind their own forward pass (_tail_forward)
    loss, fs.recs, fs.grads, fs.keep = _tail_forward(rt, p_cur, Fl, weight, jobs, streams, eager=fs.outs)
    fs.streams, fs.states, fs.eff, fs.gcache, fs.svdc, fs.actc = streams, states, eff, gcache, svdc, actc
    fs.cache_blocks, fs.adj = (cache_blocks if gcache is not None else 0), adj
    return loss, x, Fl, fs


def _frame_backward(rt, fs, g=None):
    """Reverse sweep of _frame_forward down to dL/dB, dL/dA of the six LoRA layers (finetune.py:413-414), for dL/dloss = g (None:
    1).  Returns the twelve gradients, views of one buffer, in the order B, A per layer (elasticity's three, plasticity's three)."""
    import ctypes as C
    from . import _lib as L
    from . import rollout as R
    lib, dev = L.lib(), rt.device
    sim = rt.sim_fused
    n, S = rt.n_local, int(sim.substeps)
    _, st, _jf, jb, woff, goff, sizes, shapes, gtot, nw = _frame_static(rt)
    # dL/d(x, v, C, F of the last record): B^T of the summed rasterizer adjoints lands in the head of a buffer whose
    # tail (v, C, F: nothing downstream of the roll-out reads them) stays zero
    glast = rt._scratch("glast", 4 * 24 * n, zero=True)
    dx = glast[:12 * n].view(torch.float32).view(n, 3)
    _tail_backward(rt, fs.recs, fs.grads, fs.streams, dx, outs=fs.outs)
    if not rt._unit_frame():
        dx.div_(rt.size)
    if g is not None:
        dx.mul_(g)
    stream = L.stream_ptr(dev)
    verified = 0
    gcache = fs.gcache
    if gcache is not None and fs.ev is not None:
        # the record headers travel back right behind the forward sweep: they are here long before the renders are through
        if R._CACHE_WAIT and not fs.ev.query():
            fs.ev.synchronize()
        if fs.ev.query():
            verified = int(min(fs.status.tolist()) >= 0)
            rt._status_pool.append((fs.status, fs.ev))
    gfirst = rt._scratch("gfirst", 4 * 24 * n)
    gw = torch.empty(2 * nw, dtype=torch.float32, device=dev)
    ws_bytes = rt._roll_sizes[1]
    ws = rt._scratch("ws", ws_bytes)
    svdc, actc = fs.svdc, fs.actc
    cfg = L.nm_rollout_cfg(S, float(sim.plasticity.alpha), fs.cache_blocks, verified, fs.adj,
                           svdc.t.data_ptr() if svdc is not None else None, actc.t.data_ptr() if actc is not None else None,
                           1 if rt.__dict__.get("_ws_token") == fs.ws_token and ws.data_ptr() == fs.ws_ptr else 0,
                           _LAST_GF_ZERO)      # (glast holds dL/dx and zeros - the loss sees the last record's positions only)
    base = fs.eff.data_ptr()
    w0, w1 = R._WSZ[0], R._WSZ[0] + R._WSZ[1]
    mle = L.nm_mlp(base, base + 4 * w0, base + 4 * w1)
    pb = base + 4 * nw
    mlp = L.nm_mlp(pb, pb + 4 * w0, pb + 4 * w1)
    gbase = gw.data_ptr()
    L.check(lib.nm_rollout_backward(rt.model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp), fs.states.data_ptr(),
                                    gcache.data_ptr() if gcache is not None else None, glast.data_ptr(), gfirst.data_ptr(), gbase,
                                    gbase + 4 * nw, ws.data_ptr(), ws_bytes, stream), "nm_rollout_backward")
    for lease in (svdc, actc):
        if lease is not None:
            lease.release()
    # dL/dW_eff -> dL/dB, dL/dA of the six layers: ONE launch
    gba = torch.empty(gtot, dtype=torch.float32, device=dev)
    ob = gba.data_ptr()
    for i in range(6):
        j = jb[i]
        j.W = gbase + 4 * woff[i]
        j.o0 = ob + 4 * goff[i]
        j.o1 = ob + 4 * (goff[i] + sizes[2 * i])
    L.check(lib.nm_lora_merge_layers_bwd(6, jb, stream), "nm_lora_merge_layers_bwd")
    fs.recs = fs.grads = fs.keep = fs.states = fs.eff = fs.gcache = fs.svdc = fs.actc = fs.outs = None
    return [v.view(sh) for v, sh in zip(gba.split(sizes), shapes)]


def _gt_image(rt, img, vi):
    """A ground-truth frame as nm_pixel_loss reads it: (3, H, W) fp32, contiguous, on the runtime's device (train.video_loss
    accepted slices / other dtypes through torch ops; the native epoch hands the pointer to the library)."""
    cam = rt.cameras[vi]
    want = (3, int(cam.image_height), int(cam.image_width))
    if tuple(img.shape) != want:
        raise ValueError(f"ground-truth image of view {vi} has shape {tuple(img.shape)}, the render is {want}")
    if img.device != rt.device or img.dtype != torch.float32 or not img.is_contiguous():
        img = img.to(rt.device).float().contiguous()
    return img


class _EpochState(object):
    """What the forward half of a native multi-frame epoch keeps for its reverse sweep."""
    __slots__ = ("frames", "states", "eff", "n", "S", "adj", "side", "loss_parts", "peak_note")


def _epoch_forward(rt, gt_frames, weights, views=None, frame_steps=None, start=None, overlap: bool = True):
    """Forward half of a whole BPTT epoch of a one-GPU runtime, outside autograd (finetune.py:331-392): F frames of S substeps,
    the state flowing from frame to frame, and per frame binding against the DETACHED previous frame (de_x_prev, g_prev:
    finetune.py:392-393) + V renders + decayed pixel loss.  The counterpart, for the reference's real unit of work, of
    _frame_forward: the same library calls per frame (nm_rollout_forward on consecutive records of ONE checkpoint buffer - frame
    f's record 0 IS frame f-1's last record, nothing is copied -, nm_bind_frame, rasterizer + loss), no autograd nodes, no
    engine thread.  overlap: frame f's binding / renders / loss run on a second HIP stream while frame f+1 simulates.
    weights[f]: the frame's loss weight (decay_rate ** ((f) // decay_steps)), None = the frame is excluded (exclude_steps:
    simulated, not rendered, and - as in the reference - it does not become the next frame's de_x_prev / g_prev).
    Returns (loss, _EpochState)."""
    import ctypes as C
    from . import _lib as L
    from . import rollout as R
    lib, dev = L.lib(), rt.device
    sim = rt.sim_fused
    n, S, nf = rt.n_local, int(sim.substeps), len(weights)
    main = torch.cuda.current_stream(dev)
    stream = L.stream_ptr(dev)
    _, st, jf, _jb, woff, _goff, _sizes, _shapes, _gtot, nw = _frame_static(rt)
    eff = torch.empty(2 * nw, dtype=torch.float32, device=dev)
    base = eff.data_ptr()
    for i in range(6):
        jf[i].o0 = base + 4 * woff[i]
    L.check(lib.nm_lora_merge_layers(6, jf, stream), "nm_lora_merge_layers")
    states = torch.empty(nf * S + 1, 33 * n, dtype=torch.float32, device=dev)
    start = (rt.x0, rt.v0, rt.C0, rt.F0) if start is None else start
    torch.cat([t.detach().float().reshape(-1) for t in start], out=states[0][:24 * n])
    if sim._cache_blocks is None:      # size the grid cache from what the scene touches (one throw-away roll-out, one host sync)
        with torch.no_grad():
            rt.rollout(*start)
    cache_blocks = int(sim.grid_cache_blocks())
    ws_bytes = int(lib.nm_rollout_workspace(n, S))
    gc_bytes = int(lib.nm_rollout_gridcache_bytes(S, cache_blocks)) if cache_blocks > 0 else 0
    svd_bytes, act_bytes = int(lib.nm_rollout_svdcache_bytes(n, S)), int(lib.nm_rollout_actcache_bytes(n, S))
    ws = rt._scratch("ws", ws_bytes)
    # every frame's pair of caches goes back to the pool after the reverse sweep and is found there by the next epoch (GB-sized
    # buffers: handing them to the caching allocator makes it release and re-acquire device memory inside the training loop);
    # the cap is per buffer size, so an epoch at another N or S, a single frame or an evaluation render keeps the default
    for nb_ in (svd_bytes, act_bytes):
        if nb_ > 0:
            R._POOL_CAPS[(str(dev), int(nb_))] = max(R._POOL_CAPS.get((str(dev), int(nb_)), R._POOL_CAP[0]), nf)
    adj = L.SVD_ADJOINT[sim.svd_adjoint]
    w0, w1 = R._WSZ[0], R._WSZ[0] + R._WSZ[1]
    mle = L.nm_mlp(base, base + 4 * w0, base + 4 * w1)
    pb = base + 4 * nw
    mlp = L.nm_mlp(pb, pb + 4 * w0, pb + 4 * w1)
    views = list(range(rt.V)) if views is None else list(views)
    jobs = [(vi, None) for vi in views]
    streams = rt._frame_streams(jobs) if len(jobs) > 1 else None
    side = None
    if overlap:
        side = getattr(rt, "_render_stream", None)
        if side is None:
            side = rt._render_stream = torch.cuda.Stream(device=dev)
    unit = rt._unit_frame()
    de_prev = start[0].detach().float().contiguous() if unit else ((start[0] - rt.center) / rt.size).detach().float().contiguous()
    g_prev = rt.gaussians.get_xyz.detach().float().contiguous()
    rt._cov6 = rt._cov.detach().float().reshape(-1, 6).contiguous()
    rt._tail_key = None                 # (the single-frame constants are rebuilt by the next frame())
    es = _EpochState()
    es.frames, es.loss_parts = [], []
    cached = recomputed = 0
    rec_bytes = 33 * n * 4
    for f in range(nf):
        gcache = torch.empty(gc_bytes, dtype=torch.uint8, device=dev) if gc_bytes > 0 else None
        svdc = R.lease_cache(svd_bytes, dev) if R._SVD_CACHE else None
        actc = R.lease_cache(act_bytes, dev, force=R._ACT_CACHE == '1') if R._ACT_CACHE != '0' else None
        cached += actc is not None
        recomputed += actc is None
        cfg = L.nm_rollout_cfg(S, float(sim.plasticity.alpha), cache_blocks if gcache is not None else 0, 0, adj,
                               svdc.t.data_ptr() if svdc is not None else None, actc.t.data_ptr() if actc is not None else None, 0,
                               _LAST_GF_ZERO if f == nf - 1 else 0)      # (as the reverse sweep of the last frame will: _epoch_backward)
        sptr = states.data_ptr() + f * S * rec_bytes
        gptr = gcache.data_ptr() if gcache is not None else None
        L.check(lib.nm_rollout_forward(rt.model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp), sptr, gptr,
                                       ws.data_ptr(), ws_bytes, stream), "nm_rollout_forward")
        status = ev = None
        if gcache is not None and R._CACHE_STATUS:
            status, ev = torch.empty(S, dtype=torch.int32, pin_memory=True), torch.cuda.Event()
            _status_words_out(rt, lib, gptr, cfg, status, ev, gcache)
        fr = {"gcache": gcache, "svdc": svdc, "actc": actc, "status": status, "ev": ev, "cache_blocks": cache_blocks if gcache is not None else 0,
              "tail": None}
        es.frames.append(fr)
        if weights[f] is None:
            continue
        last = states[(f + 1) * S]
        x, Fl = last[:3 * n].view(n, 3), last[15 * n:24 * n].view(n, 9)
        if side is not None:
            side.wait_stream(main)
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            p_cur = x if unit else ((x - rt.center) / rt.size).contiguous()
            rt._de_x_prev, rt._g_prev = de_prev, g_prev
            loss_f, recs, grads, keep = _tail_forward(rt, p_cur, Fl, float(weights[f]), jobs, streams,
                                                      gt={vi: _gt_image(rt, gt_frames[f][i], vi) for i, vi in enumerate(views)},      # (gt_frames[f][i] belongs to views[i])
                                                      step=None if frame_steps is None else frame_steps[f])
            es.loss_parts.append(loss_f)
            fr["tail"] = (recs, grads, keep, p_cur)
            de_prev, g_prev = p_cur, keep[0]          # (detached by construction: nothing here is in a graph)
    if side is not None:
        main.wait_stream(side)
    loss = torch.stack(es.loss_parts).sum() if es.loss_parts else torch.zeros((), dtype=torch.float32, device=dev)
    es.states, es.eff, es.n, es.S, es.adj, es.side = states, eff, n, S, adj, side
    es.peak_note = {"frames_with_activation_cache": int(cached), "frames_recomputing": int(recomputed),
                    "activation_cache_budget_GB": round(R.act_cache_budget(dev) / 2 ** 30, 1)}
    es.frames[0]["streams"] = streams
    return loss, es


class _NullCtx(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _epoch_backward(rt, es):
    """Reverse sweep of _epoch_forward (the one loss.backward() of finetune.py:413-414): frame by frame from the last, the
    rasterizer adjoints + B^T of frame f (on the second stream, under the roll-out adjoint of frame f + 1), dL/dx_f added to what
    flows back from frame f + 1, nm_rollout_backward, the LoRA weight gradients summed over the frames, one
    nm_lora_merge_layers_bwd at the end.  Returns the twelve gradients (B, A per layer)."""
    import ctypes as C
    from . import _lib as L
    from . import rollout as R
    lib, dev = L.lib(), rt.device
    n, S, nf = es.n, es.S, len(es.frames)
    sim = rt.sim_fused
    _, st, _jf, jb, woff, goff, sizes, shapes, gtot, nw = _frame_static(rt)
    main = torch.cuda.current_stream(dev)
    stream = L.stream_ptr(dev)
    side = es.side
    streams = es.frames[0].get("streams")
    base = es.eff.data_ptr()
    w0, w1 = R._WSZ[0], R._WSZ[0] + R._WSZ[1]
    mle = L.nm_mlp(base, base + 4 * w0, base + 4 * w1)
    pb = base + 4 * nw
    mlp = L.nm_mlp(pb, pb + 4 * w0, pb + 4 * w1)
    ws_bytes = int(lib.nm_rollout_workspace(n, S))
    ws = rt._scratch("ws", ws_bytes)
    bufs = [torch.zeros(24 * n, dtype=torch.float32, device=dev), torch.empty(24 * n, dtype=torch.float32, device=dev)]
    gw_tot = torch.zeros(2 * nw, dtype=torch.float32, device=dev)
    gw = torch.empty(2 * nw, dtype=torch.float32, device=dev)
    rec_bytes = 33 * n * 4
    unit = rt._unit_frame()

    def tail_bwd(f):
        """dL/dx of frame f's renders (None: the frame was excluded), enqueued on the second stream."""
        fr = es.frames[f]
        if fr["tail"] is None:
            return None
        recs, grads, keep, _p = fr["tail"]
        if side is not None:
            side.wait_stream(main)
        with torch.cuda.stream(side) if side is not None else _NullCtx():
            dx = torch.empty(n, 3, dtype=torch.float32, device=dev)
            _tail_backward(rt, recs, grads, streams, dx)
            if not unit:
                dx.div_(rt.size)
            ev = torch.cuda.Event()
            ev.record()
        fr["tail"] = None
        return dx, ev

    pending = tail_bwd(nf - 1)
    cur = 0
    for f in range(nf - 1, -1, -1):
        fr = es.frames[f]
        glast, gfirst = bufs[cur], bufs[cur ^ 1]
        nxt = tail_bwd(f - 1) if f > 0 else None        # (runs under this frame's roll-out adjoint)
        if pending is not None:
            dx, ev = pending
            main.wait_event(ev)
            glast[:3 * n].add_(dx.view(-1))
            dx.record_stream(main)
        verified = 0
        if fr["gcache"] is not None and fr["ev"] is not None:
            if R._CACHE_WAIT and not fr["ev"].query():
                fr["ev"].synchronize()
            if fr["ev"].query():
                verified = int(min(fr["status"].tolist()) >= 0)
        svdc, actc = fr["svdc"], fr["actc"]
        cfg = L.nm_rollout_cfg(S, float(sim.plasticity.alpha), fr["cache_blocks"], verified, es.adj,
                               svdc.t.data_ptr() if svdc is not None else None, actc.t.data_ptr() if actc is not None else None, 0,
                               _LAST_GF_ZERO if f == nf - 1 else 0)     # (the epoch's last frame: nothing flows into its last record but dL/dx)
        gbase = gw.data_ptr()
        L.check(lib.nm_rollout_backward(rt.model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp),
                                        es.states.data_ptr() + f * S * rec_bytes,
                                        fr["gcache"].data_ptr() if fr["gcache"] is not None else None, glast.data_ptr(), gfirst.data_ptr(),
                                        gbase, gbase + 4 * nw, ws.data_ptr(), ws_bytes, stream), "nm_rollout_backward")
        for lease in (svdc, actc):
            if lease is not None:
                lease.release()
        fr["gcache"] = fr["svdc"] = fr["actc"] = None
        torch.nan_to_num_(gfirst, 0.0, 0.0, 0.0)       # interface.py:65-74 at the boundary between two frames' roll-outs
        gw_tot.add_(gw)
        pending = nxt
        cur ^= 1
    gba = torch.empty(gtot, dtype=torch.float32, device=dev)
    ob, gbase = gba.data_ptr(), gw_tot.data_ptr()
    for i in range(6):
        j = jb[i]
        j.W = gbase + 4 * woff[i]
        j.o0 = ob + 4 * goff[i]
        j.o1 = ob + 4 * (goff[i] + sizes[2 * i])
    L.check(lib.nm_lora_merge_layers_bwd(6, jb, stream), "nm_lora_merge_layers_bwd")
    es.frames, es.states = [], None
    return [v.view(sh) for v, sh in zip(gba.split(sizes), shapes)]


class _Frame(torch.autograd.Function):
    """The whole frame of a one-GPU runtime as ONE autograd node over the LoRA factors (_frame_forward / _frame_backward): the
    same library calls, kernels and results as the composition LoRA merge -> _Rollout -> _FrameTail that it replaces; what goes
    away is the host time between them (four autograd nodes each way, the packing / unpacking of their inputs and outputs) -
    the frame time of the small configurations (bb, jd, sf) was host time.  SceneRuntime.frame(backward=True) calls the two
    halves directly and adds the gradients to the parameters' .grad itself (no graph, no engine thread); this node is the same
    thing for callers that want the loss inside a larger graph."""

    @staticmethod
    def forward(ctx, rt, weight, jobs, streams, *ba):
        loss, x, Fl, fs = _frame_forward(rt, weight, jobs, streams)
        ctx.rt, ctx.fs = rt, fs
        ctx.mark_non_differentiable(x, Fl)
        return loss, x, Fl

    @staticmethod
    def backward(ctx, g, _gx, _gF):
        grads = _frame_backward(ctx.rt, ctx.fs, g)
        ctx.fs = None
        return (None, None, None, None) + tuple(grads)


def measure_substep_us(workload: str, num_particles: int, device, reps: int = 5) -> float:
    """Microseconds per substep (forward + backward) of the fused roll-out of `workload` scaled to `num_particles`, measured on
    this GPU: one input of sim.shard.shard_cost_model at start-up (bench.py --shard-sim auto) - what the one-GPU table
    SUBSTEP_US used to stand for."""
    import time
    scene = synth.make_scene(workload, override=dict(N=int(num_particles), K=1000))
    rt = SceneRuntime(scene, device)
    params = rt.parameters()

    def once():
        for p in params:
            p.grad = None
        o = rt.rollout(*rt.start)
        (o[0].sum() + o[3].sum()).backward()

    for _ in range(3):
        once()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize(device)
    return 1e6 * (time.perf_counter() - t0) / reps / rt.S


def make_material_cfg(alpha=1e-3):
    return dict(layer_widths=[64, 64], norm=None, nonlinearity="gelu", no_bias=True, normalize_input=True, alpha=alpha)


@dataclass
class FrameResult:
    loss: torch.Tensor
    x: torch.Tensor
    F: torch.Tensor


class SceneRuntime(object):
    """Everything resident on one GPU for a synthetic scene: particles, statics, nets (+LoRA), Gaussians, bindings,
    cameras and ground-truth images."""

    def __init__(self, scene: synth.Scene, device, lora_r: int = 16, lora_alpha: int = 16, fused: bool = True,
                 bc: str = "noslip", gravity=(0.0, -9.8, 0.0), pixel_loss: str = "l2", white_bg: Optional[bool] = None,
                 rank: int = 0, world: int = 1, group=None, shard_sim: bool = False):
        self.scene, self.device, self.fused = scene, torch.device(device), fused
        self.rank, self.world, self.group = rank, world, group
        # (NEUMA_SHARD_FORCE=1: keep the exchange machinery on with a single rank, to measure its overhead on one GPU)
        self.shard_sim = bool(shard_sim) and (world > 1 or os.environ.get("NEUMA_SHARD_FORCE") == "1")
        # (shard_sim with fused=True: nm_rollout_forward_sharded runs the substep loop, phases and collectives, inside the
        # library; fused=False: the per-operator classes drive the phases from Python)
        # the render jobs of a frame (views, or view stripes on several GPUs) go round-robin over HIP streams so that one job's
        # binning (small sort / scan kernels) runs under another job's compositing kernel - same results, ~9 % shorter frame
        self.overlap_views = os.environ.get("NEUMA_OVERLAP_VIEWS", "1") != "0"
        self.num_view_streams = int(os.environ.get("NEUMA_VIEW_STREAMS", "0")) or None      # default: one stream per view
        cfg = scene.cfg
        self.S, self.V = int(cfg["S"]), int(cfg["V"])
        sim_cfg = dict(gravity=list(gravity), bc=bc, num_grids=cfg["G"], dt=cfg["dt"], bound=1, eps=6e-7)   # finetune.py:47,270
        self.model = MPMModelBuilder().parse_cfg(sim_cfg).finalize(self.device, requires_grad=True)
        N = scene.x0.shape[0]
        self.N = N
        self.x0 = torch.tensor(scene.x0, device=self.device)
        self.v0 = torch.tensor(scene.v0, device=self.device)
        self.C0 = torch.zeros(N, 3, 3, device=self.device)
        self.F0 = torch.eye(3, device=self.device).repeat(N, 1, 1)
        # the particle rows this rank simulates: all of them, or its contiguous range of the (Hilbert-ordered) list
        self.rows = slice(0, N)
        if world > 1:
            # the one collective point at which the library's RCCL communicator of this group comes into being (every rank builds
            # its runtime): the frame's collectives (sim.shard.all_reduce_sum_ / all_gather_rows_, reached from autograd backward
            # functions) only look it up
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
                from .sim.shard import create_library_comm
                create_library_comm(group, self.device)
        if self.shard_sim:
            from .sim.shard import shard_range
            self.rows = slice(*shard_range(N, world, rank))
            self.model.shard(group)
        n_local = self.n_local = self.rows.stop - self.rows.start
        st = MPMStatics()
        st.init(n_local, self.device)
        st.vol.fill_(scene.vol); st.rho.fill_(1000.0); st.clip_bound.fill_(0.1); st.enabled.fill_(1)
        self.statics = st
        # constitutive nets: shipped checkpoint + LoRA (finetune.py:295-313)
        w = synth.load_base_weights(cfg["mat"])
        mcfg = make_material_cfg()
        self.elasticity = InvariantFullMetaElasticity(mcfg).to(self.device)
        self.plasticity = InvariantFullMetaPlasticity(mcfg).to(self.device)
        for net, tag in ((self.elasticity, "e"), (self.plasticity, "p")):
            net.layers[0].fc.weight.data.copy_(torch.tensor(w[tag][0]))
            net.layers[1].fc.weight.data.copy_(torch.tensor(w[tag][1]))
            net.final_layer.fc.weight.data.copy_(torch.tensor(w[tag][2]))
        if lora_r > 0:
            g = torch.Generator().manual_seed(0)
            for net in (self.elasticity, self.plasticity):
                net.init_lora_layers(r=lora_r, lora_alpha=lora_alpha)
                net.freeze_all_except_lora()
                for lin in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc):
                    # lora_A ~ kaiming_uniform (a = sqrt 5, i.e. U(+-1/sqrt(fan_in))) and a non-zero lora_B, both
                    # from the seeded generator so that every runtime / rank builds the same adaptor
                    bound = 1.0 / (lin.lora_A.shape[1] ** 0.5)
                    lin.lora_A.data.copy_((2.0 * torch.rand(lin.lora_A.shape, generator=g) - 1.0) * bound)
                    lin.lora_B.data.copy_(0.01 * torch.randn(lin.lora_B.shape, generator=g))
        self.sim_cached = MPMCacheDiffSim(self.model, 4096)
        self.sim_fused = MPMFusedDiffSim(self.model, self.elasticity, self.plasticity, self.S)
        # Gaussians.  A trained 3DGS checkpoint lists its kernels in training order - random in space - while the particles are
        # Hilbert-ordered: every binding row then gathers particle rows from all over the array (k_bind_frame: 4.6x its
        # compulsory traffic) and every particle's B^T row gathers Gaussians from all over theirs.  The runtime keeps its OWN
        # order - the Hilbert order of the kernels' rest positions, `gaussian_perm[i]` = the caller's index of kernel i -
        # for every per-kernel array and the binding matrix's rows; nothing per-kernel leaves the runtime (images and particle
        # gradients do), and the depth sort's tie break is by this index instead of the caller's.  NEUMA_GAUSSIAN_ORDER=given
        # keeps the caller's order.
        self.gaussian_perm = None
        if os.environ.get("NEUMA_GAUSSIAN_ORDER", "spatial") == "spatial" and scene.g_xyz.shape[0] > 1:
            import copy
            lo, hi = scene.g_xyz.min(0), scene.g_xyz.max(0)
            cells = np.clip(((scene.g_xyz - lo) / np.maximum(hi - lo, 1e-12).max() * 1023.0).astype(np.int64), 0, 1023)
            perm = np.argsort(synth.hilbert_index(cells, 10), kind="stable")
            scene = copy.copy(scene)
            for k in ("g_xyz", "g_sh", "g_logscale", "g_rot", "g_opacity_logit", "bind_idx", "bind_w"):
                setattr(scene, k, np.ascontiguousarray(getattr(scene, k)[perm]))
            self.gaussian_perm = torch.from_numpy(perm)
        gm = GaussianModel(cfg["sh"])
        sh = torch.tensor(scene.g_sh, device=self.device)
        gm.set_params(torch.tensor(scene.g_xyz, device=self.device), sh[:, :1].contiguous(), sh[:, 1:].contiguous(),
                      torch.tensor(scene.g_logscale, device=self.device), torch.tensor(scene.g_rot, device=self.device),
                      torch.tensor(scene.g_opacity_logit, device=self.device))
        self.gaussians = gm
        self._opacity = gm.get_opacity.contiguous()
        self._shs = gm.get_features.contiguous()
        self._cov = gm.get_covariance(1.0)
        K, nb = scene.bind_idx.shape
        rows = torch.arange(K).repeat_interleave(nb)
        ind = torch.stack([rows, torch.tensor(scene.bind_idx.reshape(-1))], 0)
        self.bindings = Bindings(ind, torch.tensor(scene.bind_w.reshape(-1)), (K, N), self.device)
        self.K = K
        if white_bg is None:        # synthetic scenes render on white, the real-world ones (burger) on black
            white_bg = cfg.get("bg", "white") == "white"
        self.background = (torch.ones(3) if white_bg else torch.zeros(3)).to(self.device)
        self.cameras = synth.ring_cameras(self.V, cfg["W"], cfg["H"], device=self.device)
        self.pixel_loss = PIXEL_LOSSES[pixel_loss]
        self.tile_rows = (cfg["H"] + 15) // 16
        self.gt: List[Optional[torch.Tensor]] = [None] * self.V
        self.center = torch.zeros(3, device=self.device)
        self.size = torch.ones(3, device=self.device)
        # the state every frame() starts from (default: the rest state) and the Gaussian centres that belong to it
        self._start = None
        self._g_start = None
        self.state_kind = "rest"

    @property
    def start(self):
        return (self.x0, self.v0, self.C0, self.F0) if self._start is None else self._start

    @property
    def g_start(self):
        return self.gaussians.get_xyz if self._g_start is None else self._g_start

    @torch.no_grad()
    def set_start_state(self, kind: str = "rest", seed: int = 3, max_steps: int = 4000):
        """Choose the state the timed frame starts from.
        rest      x0, v0, C = 0, F = I (the SURVEY §8d generator: free fall, F stays ~ I - the best case for the Jacobi SVD)
        deformed  rest positions with F = I + 0.05 N(0,1): every particle needs the full SVD sweeps, stresses are large
        impact    forward-simulate (no autograd) until the body has hit the floor and rebounded into a compressed state:
                  real contact, wall boundary conditions active, spatially varying F
        The Gaussians move with the particles: g_start = g_0 + B (x_start - x_0) (tune/utils.py:424-448)."""
        if kind == "rest":
            self._start = self._g_start = None
            self.state_kind = kind
            return self.start
        elif kind == "deformed":
            g = torch.Generator().manual_seed(seed)
            F = self.F0 + (0.05 * torch.randn(self.F0.shape, generator=g)).to(self.device)
            self._start = (self.x0, self.v0, self.C0, F.contiguous())
        elif kind == "impact":
            if self.shard_sim:
                raise ValueError("start state 'impact' is prepared on an unsharded runtime")
            x, v, C, F = self.x0, self.v0, self.C0, self.F0
            was, S = self.fused, self.S
            self.fused, self.S = False, 1
            dx = 1.0 / self.scene.cfg["G"]
            hit = None
            for it in range(max_steps):
                x, v, C, F = self.rollout(x, v, C, F, step0=3000)
                if it % 20 == 19:
                    if hit is None and float(x[:, 1].min()) < 2.5 * dx:
                        hit = it
                    # keep going until the bulk is clearly compressed (mean vertical stretch below 0.97) or 400 steps after contact
                    if hit is not None and (float(F[:, 1, 1].mean()) < 0.97 or it > hit + 400):
                        break
            self.fused, self.S = was, S
            self._start = tuple(t.contiguous() for t in (x, v, C, F))
        else:
            raise ValueError(f"unknown start state {kind!r}")
        self.state_kind = kind
        self._g_start = compute_bindings_xyz(self.start[0], self.x0, self.gaussians.get_xyz, self.bindings).detach()
        return self.start

    # ---- pieces
    def parameters(self):
        return [p for net in (self.elasticity, self.plasticity) for p in net.parameters() if p.requires_grad]

    # ---- the one-node frame (_Frame)
    def _lean_ok(self) -> bool:
        """The frame can run as ONE autograd node: simulation not sharded (one GPU, or several with the simulation replicated and
        the render in stripes), library roll-out, particles in scatter order, and the only trainable
        tensors are the LoRA factors of the six layers (finetune.py:295-313, the configuration NeuMA trains in).  Decided once
        per set of layer modules (adding LoRA replaces them); `_lean = None` forgets the decision after a manual (un)freeze."""
        fcs = [fc for net in (self.elasticity, self.plasticity) for fc in (net.layers[0].fc, net.layers[1].fc, net.final_layer.fc)]
        key = tuple(id(fc) for fc in fcs)
        ok = getattr(self, "_lean", None)
        if ok is None or ok[0] != key:
            from .material.loralib import LinearLoRA
            good = (not self.shard_sim and self.model.exchange is None
                    and all(isinstance(fc, LinearLoRA) and fc.r > 0 and not fc.merged and fc.weight.is_cuda and not fc.weight.requires_grad
                            and fc.lora_A.requires_grad and fc.lora_B.requires_grad and fc.bias is None
                            and fc.weight.dtype == torch.float32 and fc.weight.is_contiguous() for fc in fcs)
                    and len(self.parameters()) == 12 and not self.sim_fused.order.active(self.start[0]))
            ok = self._lean = (key, good)
            self._lora_layers = fcs
        # (a start state that wants gradients - the initial-velocity stage - goes through the roll-out node, which returns them)
        return (ok[1] and self.fused and getattr(self, "fused_tail", True) and torch.is_grad_enabled()
                and not any(t.requires_grad for t in self.start) and os.environ.get("NEUMA_FUSED_TAIL", "1") != "0" and os.environ.get("NEUMA_LEAN_FRAME", "1") != "0")

    def _start_packed(self):
        """x | v | C | F of the start state as one contiguous record (rebuilt when the start state changes)."""
        st = self.start
        key = tuple(t.data_ptr() for t in st) + tuple(t._version for t in st)
        if getattr(self, "_packed_key", None) != key:
            self._packed = torch.cat([t.detach().float().reshape(-1) for t in st])
            self._packed_key = key
        return self._packed

    def _unit_frame(self) -> bool:
        """de_x = (x - center) / size is the identity (synthetic scenes)."""
        key = (self.center.data_ptr(), self.center._version, self.size.data_ptr(), self.size._version)
        if getattr(self, "_unit_key", None) != key:
            self._unit = bool((self.center == 0).all()) and bool((self.size == 1).all())
            self._unit_key = key
        return self._unit

    def _scratch(self, name: str, nbytes: int, zero: bool = False):
        """Per-runtime device buffers that live across frames (stream-ordered reuse): the roll-out workspace, dL/dstate records."""
        pool = self.__dict__.setdefault("_scratch_pool", {})
        t = pool.get(name)
        if t is None or t.numel() != max(int(nbytes), 4):
            t = pool[name] = (torch.zeros if zero else torch.empty)(max(int(nbytes), 4), dtype=torch.uint8, device=self.device)
        return t

    def rollout(self, x, v, C, F, step0: int = 0):
        """S substeps (finetune.py:360-364)."""
        if self.fused:
            return self.sim_fused(self.statics, x, v, C, F)
        for it in range(self.S):
            stress = self.elasticity(F)
            x, v, C, F = self.sim_cached(self.statics, step0 + it, x, v, C, F, stress)
            F = self.plasticity(F)
        return x, v, C, F

    def camera_at(self, view: int, step=None):
        """Camera of view index `view` at dataset frame `step` (synthetic scenes: static cameras, `step` is ignored)."""
        return self.cameras[view]

    def render_view(self, means3D, deform_grad, view: int, tile_rows=None, cov=None, step=None):
        """cov: covariances already pushed forward by the deformation gradients (deform_grad is then ignored)."""
        if getattr(self, "force_mask_data", False):       # silhouette supervision: constant colour 1 (tune/utils.py:390-404)
            return diff_rasterization(means3D, deform_grad if cov is None else None, None, self.camera_at(view, step), self.background,
                                      gaussians_active_sh=self.gaussians.active_sh_degree, guassians_cov=self._cov if cov is None else cov,
                                      gaussians_opa=self._opacity, gaussians_shs=self._shs, force_mask_data=True, tile_rows=tile_rows)
        return diff_rasterization(means3D, deform_grad if cov is None else None, None, self.camera_at(view, step), self.background,
                                  gaussians_active_sh=self.gaussians.active_sh_degree,
                                  guassians_cov=self._cov if cov is None else cov,
                                  gaussians_opa=self._opacity, gaussians_shs=self._shs, tile_rows=tile_rows)

    @torch.no_grad()
    def make_ground_truth(self, perturb: float = 0.02, steps: int = 5, seed: int = 2):
        """GT image per view = render of the state after a few perturbed substeps."""
        g = torch.Generator().manual_seed(seed)
        x, v, C, F = self.start
        v = v + (perturb * torch.randn(self.v0.shape, generator=g)).to(self.device)
        was = self.fused
        self.fused = False
        S = self.S
        self.S = steps
        x, v, C, F = self.rollout(x[self.rows], v[self.rows], C[self.rows], F[self.rows], step0=2048)
        x, F = self.all_rows(x), self.all_rows(F)
        self.S, self.fused = S, was
        means3D = compute_bindings_xyz(x, self.start[0], self.g_start, self.bindings)
        dg = compute_bindings_F(F, self.bindings)
        for vi in range(self.V):
            self.gt[vi] = self.render_view(means3D, dg, vi).detach().clone()

    def all_rows(self, t: torch.Tensor, differentiable: bool = False) -> torch.Tensor:
        """This rank's particle rows -> all particles (identity unless the simulation is sharded).  The gradient that
        comes back through it is already summed over the ranks (merge_grad_across_ranks sits downstream)."""
        if not self.shard_sim:
            return t
        from .sim.shard import gather_rows
        return gather_rows(t if differentiable else t.detach(), self.N, self.group, grad_is_summed=True)

    # ---- one frame, forward + backward
    def frame(self, weight: float = 1.0, backward: bool = True) -> FrameResult:
        self._frame_no = getattr(self, "_frame_no", 0) + 1      # (the same on every rank: the stripe plans are keyed to it)
        if getattr(self, "_de_prev_key", None) != self.start[0].data_ptr():      # (constant between start-state changes)
            self._de_prev = ((self.start[0] - self.center) / self.size).detach()
            self._de_prev_key = self.start[0].data_ptr()
        de_x_prev = self._de_prev
        g_prev = self.g_start
        if backward and self._lean_ok():
            # LoRA training, simulation on this GPU: the whole frame is two plain calls (_frame_forward / _frame_backward).
            # Several ranks with the simulation replicated: every rank rolls out all particles, renders its stripes, and the
            # reverse sweep's one K x 3 all-reduce (inside _tail_backward) gives every rank the full gradient
            if self.world == 1:
                jobs = self._lean_jobs
            else:
                jobs = [(vi, (r0, r1)) for (vi, r0, r1) in stripe_plan(self.V, self.tile_rows, self.world, self.rank, self._stripe_weights())]
            streams = self._frame_streams(jobs)
            self._tail_constants(de_x_prev, g_prev)
            ba = [t for l in self._lora_layers for t in (l.lora_B, l.lora_A)]
            if os.environ.get("NEUMA_LEAN_GRAPH") == "1":       # (the same through the autograd engine: tests)
                loss, x, F = _Frame.apply(self, float(weight), jobs, streams, *ba)
                loss.backward()
                return FrameResult(loss.detach(), x, F.view(-1, 3, 3))
            # forward, reverse sweep, and what loss.backward() would do with the result: accumulate into .grad
            with torch.no_grad():
                loss, x, F, fs = _frame_forward(self, float(weight), jobs, streams, eager=os.environ.get("NEUMA_EAGER_RENDER_BWD", "1") != "0")
                for p, gr in zip(ba, _frame_backward(self, fs)):
                    if p.grad is None:
                        p.grad = gr
                    else:
                        p.grad.add_(gr)
            if self.world > 1:
                self._collect_stripe_work(jobs)
            return FrameResult(loss, x, F.view(-1, 3, 3))
        rows = self.rows
        x, v, C, F = (t[rows] for t in self.start)
        x, v, C, F = self.rollout(x, v, C, F)
        if self.shard_sim:
            self.model.exchange.defer_check()                                 # checked once, at the end of the frame
            x, F = self.all_rows(x, differentiable=True), self.all_rows(F)    # the bindings need every particle
        de_x = (x - self.center) / self.size                                  # finetune.py:373
        H = self.scene.cfg["H"]
        # render jobs of this rank: whole views on one GPU, (view, tile-row stripe) pieces when the frame is split
        if self.world == 1:
            jobs = [(vi, None) for vi in range(self.V)]                         # :378-389
        else:
            jobs = [(vi, (r0, r1)) for (vi, r0, r1) in stripe_plan(self.V, self.tile_rows, self.world, self.rank, self._stripe_weights())]
        # the jobs go round-robin over HIP streams: one job's binning (counts, scans, cell sorts: small latency-bound kernels)
        # executes under another job's compositing kernel - same results, ~9 % shorter frame
        streams = self._frame_streams(jobs)
        if getattr(self, "fused_tail", True) and os.environ.get("NEUMA_FUSED_TAIL", "1") != "0":
            # one autograd node for binding + covariance push-forward + every render job + loss (_FrameTail)
            self._tail_constants(de_x_prev, g_prev)
            loss = _FrameTail.apply(self, de_x, F, float(weight), jobs, streams)
        else:
            loss = self._tail_per_operator(de_x, de_x_prev, g_prev, F, weight, jobs, streams, H)
        if backward:
            loss.backward()
            if self.shard_sim:
                from .sim.shard import reduce_param_grads
                reduce_param_grads(self.parameters(), self.group)             # each rank saw only its particles
        if self.shard_sim:
            # raises if a substep's block exchange was incomplete (capacity exceeded, a particle outside its rank's announced
            # neighbourhood) - BEFORE the caller can step an optimizer with this frame's gradients.  Fused roll-outs report
            # through pinned status words copied out right behind the forward sweep: the host waits for that sweep here while the
            # device still has the renders and the reverse sweep queued, so the frame loop does not drain
            self.model.exchange.check(wait="watched" if self.fused else True)
        if self.world > 1:
            self._collect_stripe_work(jobs)
        return FrameResult(loss.detach(), x.detach(), F.detach())

    def epoch(self, gt_frames, weights, views=None, frame_steps=None, overlap: bool = True, backward: bool = True):
        """One BPTT epoch natively (finetune.py:331-414): len(weights) frames of S substeps from (x0, v0, C0, F0), per frame binding
        + renders of `views` + weights[f] * pixel loss against gt_frames[f][i] (None: frame excluded), then the whole reverse
        sweep; the LoRA gradients are ADDED to .grad like loss.backward() would.  Two plain calls (_epoch_forward /
        _epoch_backward), no autograd graph: what train.video_loss + loss.backward() compute through one node per frame.
        Returns the (detached) loss.  Needs the configuration frame()'s two-call path needs (_lean_ok)."""
        if not self._lean_ok():
            raise RuntimeError("the native epoch needs a one-GPU runtime with LoRA on all six layers as the only trainable tensors "
                               "(SceneRuntime._lean_ok); use train.video_loss otherwise")
        ba = [t for l in self._lora_layers for t in (l.lora_B, l.lora_A)]
        with torch.no_grad():
            loss, es = _epoch_forward(self, gt_frames, weights, views, frame_steps, None, overlap)
            self.last_epoch_note = es.peak_note
            if backward:
                for p, gr in zip(ba, _epoch_backward(self, es)):
                    if p.grad is None:
                        p.grad = gr
                    else:
                        p.grad.add_(gr)
        return loss

    @property
    def _lean_jobs(self):
        jobs = getattr(self, "_jobs1", None)
        if jobs is None or len(jobs) != self.V:
            jobs = self._jobs1 = [(vi, None) for vi in range(self.V)]
        return jobs

    def _frame_streams(self, jobs):
        """HIP streams of the frame's render jobs (None: all on the caller's stream) and the compositing mode that goes with it."""
        streams = None
        if getattr(self, "overlap_views", False) and len(jobs) > 1:
            if not hasattr(self, "_view_streams"):
                self._view_streams = [torch.cuda.Stream(device=self.device) for _ in range(int(self.num_view_streams or self.V))]
            streams = [self._view_streams[k % len(self._view_streams)] for k in range(len(jobs))]
        if getattr(self, "_hint_mode", None) != bool(streams) and os.environ.get("NEUMA_HINT_FWD_LEN") is None:
            # several render jobs share the chip (streams): the forward pass walks its tiles front to back and only the reverse
            # sweep runs in segments - the T = 1 starts of parallel forward segments are extra work that then displaces another
            # view's (metric frame: 140.3 -> 142.2 frames/s); a view that has the chip to itself needs the segments for its
            # latency (sf 603 -> 1035 frames/s).  Process-wide knob of the library, set when the situation changes.
            from . import _lib as L
            L.check(L.lib().nm_raster_set_hinted(0x7FFFFFFF if streams else 0, 256), "nm_raster_set_hinted")      # (>= any capacity: the library then skips its second and third compositing pass)
            # ... and the reverse sweeps of views that share the chip composite two pixels per lane (metric frame 186.7 -> 188.9
            # frames/s; a lone view would lose 15 %)
            L.check(L.lib().nm_raster_set_reverse_px2(1 if streams else 0), "nm_raster_set_reverse_px2")
            self._hint_mode = bool(streams)
        return streams

    def flush(self):
        """Wait for everything enqueued so far and raise what is still unreported (sharded exchanges, rasterizer overflows)."""
        from .render import flush_pending
        if self.shard_sim:
            self.model.exchange.check()
        flush_pending()

    def _tail_constants(self, de_x_prev, g_prev):
        """Frame-invariant operands of _FrameTail as contiguous fp32 (rebuilt when the start state changes)."""
        key = (de_x_prev.data_ptr(), g_prev.data_ptr(), self._cov.data_ptr())
        if getattr(self, "_tail_key", None) != key:
            self._de_x_prev = de_x_prev.detach().float().contiguous()
            self._g_prev = g_prev.detach().float().contiguous()
            self._cov6 = self._cov.detach().float().reshape(-1, 6).contiguous()
            self._tail_key = key

    def _ones_rgb(self, K):
        if getattr(self, "_ones", None) is None or self._ones.shape[0] != K:
            self._ones = torch.ones(K, 3, device=self.device)
        return self._ones

    def _tail_per_operator(self, de_x, de_x_prev, g_prev, F, weight, jobs, streams, H):
        """The same through the drop-in operators, one autograd node each (tune.py / render): the reference's composition."""
        means3D = compute_bindings_xyz(de_x, de_x_prev, g_prev, self.bindings)  # :375
        deform_grad = compute_bindings_F(F, self.bindings)                      # :376
        if self.world > 1:          # a single-rank runtime inside a multi-rank job (tests) must not join the collective
            means3D = merge_grad_across_ranks(means3D, self.group)
        loss = torch.zeros((), device=self.device)
        # the covariance push-forward is the same for every view of the frame: once, not per view
        from .render import deform_cov_by_F
        cov = deform_cov_by_F(self._cov, deform_grad)

        def job_loss(vi, rows):
            if rows is None:
                return weight * self.pixel_loss(self.render_view(means3D, None, vi, cov=cov), self.gt[vi])
            render = self.render_view(means3D, None, vi, tile_rows=rows, cov=cov)
            y0, y1 = rows[0] * 16, min(H, rows[1] * 16)
            # partial sums of the mean over the full image: the ranks' losses add up to the 1-GPU loss
            return weight * pixel_loss_rows(render, self.gt[vi], 0 if self.pixel_loss is l1_loss else 1, y0, y1)

        if streams:
            main = torch.cuda.current_stream(self.device)
            terms = []
            for s, (vi, rows) in zip(streams, jobs):
                s.wait_stream(main)
                for t in (means3D, cov):
                    t.record_stream(s)
                with torch.cuda.stream(s):
                    terms.append(job_loss(vi, rows))
            for s in self._view_streams:
                main.wait_stream(s)
            for t in terms:
                t.record_stream(main)
                loss = loss + t
        else:
            for vi, rows in jobs:
                loss = loss + job_loss(vi, rows)
        return loss

    # ---- stripes balanced by the compositing work of the previous frame
    def _stripe_weights(self):
        """V x tile_rows work estimates all ranks agree on (None until a frame has been measured).  A new plan is adopted only
        when the one in use would leave some rank with > 10 % more than its share: stripes that move every frame would make
        the rasterizer re-size its lists for every new cut.
        WHEN a measurement is adopted is a function of the frame counter alone - the table collected in frame n is looked at
        in frame n + STRIPE_ADOPT_AFTER, after waiting for its copy (long finished by then) - never of how far this rank's
        host happens to run ahead of its GPU: ranks that switched plans in different frames would render overlapping or
        missing tile rows and pair a V x tile_rows all-reduce with another rank's K x 3 one."""
        pend = getattr(self, "_stripe_pending", None)
        if pend is not None and self._frame_no >= pend[2]:
            pend[1].synchronize()
            self._stripe_pending = None
            flat = pend[0].numpy().copy()
            w, invalid = flat[:-1].reshape(self.V, self.tile_rows), float(flat[-1])
            cur = getattr(self, "_stripe_w", None)
            if invalid > 0.0:          # some rank had no walk record for one of its stripes: keep the plan in use, everywhere
                pass
            elif cur is None:
                self._stripe_w = w
            else:
                parts = [sum(float(w[v, a:b].sum()) for (v, a, b) in stripe_plan(self.V, self.tile_rows, self.world, r, cur))
                         for r in range(self.world)]
                if max(parts) > 1.10 * (sum(parts) / self.world):
                    self._stripe_w = w
        return getattr(self, "_stripe_w", None)

    STRIPE_ADOPT_AFTER = 2

    def _collect_stripe_work(self, jobs):
        """Per (view, tile row): the list entries its tiles walked in this frame's renders (the cameras' walk records; every rank
        knows its own stripes), summed over the ranks by one small all-reduce and copied to the host asynchronously.  Whether a
        frame collects depends only on state every rank shares (no measurement in flight, i.e. the frame counter)."""
        import torch.distributed as dist
        if getattr(self, "_stripe_pending", None) is not None or os.environ.get("NEUMA_STRIPE_BALANCE", "1") == "0":
            return
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) < 2:
            return              # (a runtime that only pretends to be one of several ranks: tests)
        W = torch.zeros(self.V * self.tile_rows + 1, dtype=torch.float32, device=self.device)
        Wv = W[:-1].view(self.V, self.tile_rows)
        gx = (int(self.scene.cfg["W"]) + 15) // 16
        for vi, rows in jobs:
            stores = getattr(self.camera_at(vi), "_nm_walk_stores", None)
            walk = stores[1].get(self.device) if stores else None
            if walk is None:
                W[-1] = 1.0         # (hinting switched off / no record yet: this rank still joins the collective, the
                continue            #  measurement is discarded on every rank)
            r0, r1 = rows if rows is not None else (0, self.tile_rows)
            t = walk.view(-1, gx)[r0:r1].float()
            Wv[vi, r0:r1] = t.sum(1) + 32.0 * (t > 0).float().sum(1)
        from .sim.shard import all_reduce_sum_
        all_reduce_sum_(W, self.group)
        host = torch.empty(self.V * self.tile_rows + 1, dtype=torch.float32, pin_memory=True)
        host.copy_(W, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stripe_pending = (host, ev, self._frame_no + self.STRIPE_ADOPT_AFTER)


class DiskRuntime(SceneRuntime):
    """The same runtime assembled from a NeuMA experiment on disk instead of a synthetic scene: what
    experiments/finetune.py:491-663 builds before it enters the two training stages.

        dataset     dataset.VideoDataset (cameras + ground-truth images per (view, frame id), init_x / init_v)
        gaussians   GaussianModel from <assets>/<sim_data_name>/kernels.ply
        bindings    tune.Bindings from bindings.pt
        init_data   MPMInitData of the particle set (positions in simulation space, vol, rho, clip_bound, size, center)
        nets        InvariantFullMetaElasticity / Plasticity with the base checkpoint loaded (LoRA is added by the trainer)
    `views`: names of the dataset views used for the loss (cfg.constitution.views / cfg.velocity.views)."""

    def __init__(self, sim_cfg, dataset, gaussians, bindings, init_data, elasticity, plasticity, background, device, substeps: int,
                 views, scaling_modifier: float = 1.0, pixel_loss: str = "l2", fused: bool = True, eps: Optional[float] = 6e-7):
        from types import SimpleNamespace
        self.device = torch.device(device)
        self.fused, self.rank, self.world, self.group, self.shard_sim = fused, 0, 1, None, False
        self.overlap_views = os.environ.get("NEUMA_OVERLAP_VIEWS", "1") != "0"
        self.num_view_streams = None
        self.dataset = dataset
        self.view_names = sorted(views)                                          # finetune.py:262
        self.S, self.V = int(substeps), len(self.view_names)
        sim_cfg = dict(sim_cfg)
        if eps is not None:
            sim_cfg["eps"] = eps                                                 # finetune.py:105 / 270: cfg.sim.eps = EPS
        self.model = MPMModelBuilder().parse_cfg(sim_cfg).finalize(self.device, requires_grad=True)
        N = int(init_data.pos.shape[0])
        self.N = self.n_local = N
        self.rows = slice(0, N)
        first = dataset.getCameras(self.view_names[0], dataset.steps[0])
        self.scene = SimpleNamespace(name="disk", cfg=dict(N=N, G=int(sim_cfg["num_grids"]), K=int(gaussians.get_xyz.shape[0]),
                                                            W=first.image_width, H=first.image_height, dt=float(sim_cfg["dt"]), S=self.S,
                                                            V=self.V, sh=gaussians.active_sh_degree, mat="checkpoint"))
        self.x0 = dataset.get_init_x.to(self.device).float().contiguous()
        self.v0 = dataset.get_init_v.detach().to(self.device).float().contiguous()
        self.C0 = torch.zeros(N, 3, 3, device=self.device)
        self.F0 = torch.eye(3, device=self.device).repeat(N, 1, 1)
        from .sim import MPMStaticsInitializer
        self.statics_initializer = MPMStaticsInitializer(self.model)
        self.statics_initializer.add_group(init_data)
        self.statics = self.statics_initializer.finalize()
        self.init_data = init_data
        self.elasticity, self.plasticity = elasticity.to(self.device), plasticity.to(self.device)
        self.sim_cached = MPMCacheDiffSim(self.model, 1 << 16)
        self._sim_fused = None
        self.gaussians = gaussians
        self.gaussian_perm = None       # (the caller's GaussianModel is used as it is: its order is the caller's)
        self._opacity = gaussians.get_opacity.contiguous()
        self._shs = gaussians.get_features.contiguous()
        self._cov = gaussians.get_covariance(scaling_modifier)
        self.bindings = Bindings.of(bindings)
        self.K = int(gaussians.get_xyz.shape[0])
        self.background = background.to(self.device)
        self.cameras = [dataset.getCameras(v, dataset.steps[0]) for v in self.view_names]
        self.pixel_loss = PIXEL_LOSSES[pixel_loss]
        self.tile_rows = (first.image_height + 15) // 16
        self.gt = [None] * self.V
        # de-normalisation of particle positions (nclaw/utils.py:110-118): (x - center) / size
        self.center = torch.as_tensor(np.asarray(init_data.center), dtype=torch.float32, device=self.device)
        self.size = torch.as_tensor(np.asarray(init_data.size), dtype=torch.float32, device=self.device)
        self._start = self._g_start = None
        self.state_kind = "rest"

    @property
    def sim_fused(self):
        # built on first use: the trainer adds the LoRA layers to the nets after this runtime exists
        if self._sim_fused is None or self._sim_fused.substeps != self.S:
            self._sim_fused = MPMFusedDiffSim(self.model, self.elasticity, self.plasticity, self.S)
        return self._sim_fused

    def camera_at(self, view: int, step=None):
        if step is None:
            return self.cameras[view]
        return self.dataset.getCameras(self.view_names[view], step)

    def ground_truth(self, num_frames: int):
        """gt[f-1][i] = image of view i at the f-th frame AFTER the first one (finetune.py:369, 386)."""
        out = []
        for f in range(1, num_frames + 1):
            step = self.dataset.steps[f]
            out.append([self.dataset.getCameras(v, step).original_image.to(self.device) for v in self.view_names])
        return out
