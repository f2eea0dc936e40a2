"""Fused roll-out fast path: S substeps of  stress=E(F); x,v,C,F=sim(...); F=P(F)  (finetune.py:360-364) as ONE
autograd node backed by nm_rollout_forward / nm_rollout_backward.  Keeps 96 B/particle/substep of checkpoints
and recomputes everything else in the reverse sweep (the substep's stress is checkpointed too: 132 B/particle).  Results equal the per-operator path
(MPMCacheDiffSim + material modules) up to fp32 summation order; tests/test_gpu_rollout.py checks that."""
import ctypes as C

import torch
import torch.autograd as autograd
import torch.nn as nn
from torch import Tensor

from . import _lib as L
from .sim.mpm import MPMModel, MPMStatics
from .sim.order import ParticleOrder, hilbert_index_torch  # noqa: F401 (re-exported)

_WSZ = (64 * 13, 64 * 64, 9 * 64)
_CACHE_STATUS = __import__('os').environ.get('NEUMA_CACHE_STATUS', '1') != '0'
_SVD_CACHE = __import__('os').environ.get('NEUMA_SVD_CACHE', '1') != '0'
_CACHE_WAIT = __import__('os').environ.get('NEUMA_CACHE_WAIT', '1') != '0'
# activation cache of the fused roll-out (1.2 KB per particle and substep; nm_rollout_cfg.act_cache): the forward kernels keep
# the second hidden layer of the MLPs, the reverse sweep loads it instead of recomputing (metric workload: 127.5 -> 132.3 frames/s,
# 2.3 GB per 20-substep node).  'auto' (default): on while the caches of all live roll-out nodes stay inside the budget
# (act_cache_budget: NEUMA_ACT_CACHE_GB if set, else half of the device memory that is free when the first cache is asked
# for); '1': always; '0': never (recompute, the reference's memory profile)
_ACT_CACHE = __import__('os').environ.get('NEUMA_ACT_CACHE', 'auto')
# NEUMA_FWD_PAIR=0: one launch per net in the forward sweep instead of plasticity(t) + elasticity(t+1) in one (A/B runs)
_FWD_PAIR = __import__('os').environ.get('NEUMA_FWD_PAIR')
_FWD_PAIR_SET = [False]
_ACT_CACHE_GB = __import__('os').environ.get('NEUMA_ACT_CACHE_GB')       # None: derived from the free device memory, once
_BUDGET = {}            # device -> bytes
_ACT_LIVE = {}          # device -> bytes of activation cache held by live roll-out nodes (live_bytes)
_POOL_CAP = [4]         # idle buffers kept per size ...
_POOL_CAPS = {}         # ... unless the size has a cap of its own: (device, bytes) -> idle buffers kept (a multi-frame epoch
                        # keeps one pair of caches per frame between epochs: harness.SceneRuntime.epoch)
_POOL = {}              # (device, bytes) -> idle cache buffers.  The caches are GB-sized: handing them back to the caching
                        # allocator every frame makes it release and re-acquire device memory now and then (tens of
                        # milliseconds inside a training loop), so a node returns them here after its backward pass


def act_cache_budget(device) -> int:
    """Bytes the SVD / activation caches of all live roll-out nodes may hold together: NEUMA_ACT_CACHE_GB, or half of what
    torch.cuda.mem_get_info reports free at the first call (MI355X, 288 GB: ~135 GB; the rest is for the checkpoints, the
    grid cache records, the rasterizer state kept per frame and view and the caller's own tensors)."""
    key = str(device)
    b = _BUDGET.get(key)
    if b is None:
        if _ACT_CACHE_GB is not None:
            b = int(float(_ACT_CACHE_GB) * (1 << 30))
        else:
            idle = sum(k[1] * len(v) for k, v in _POOL.items() if k[0] == key)
            b = (int(torch.cuda.mem_get_info(device)[0]) + idle) // 2
        _BUDGET[key] = b
    return b


def live_bytes(device=None) -> int:
    """Cache bytes held by live roll-out nodes on `device` (None: on all devices)."""
    return sum(_ACT_LIVE.values()) if device is None else _ACT_LIVE.get(str(device), 0)


def trim_pool(device=None) -> int:
    """Drop the idle pooled cache buffers (of `device`; None: of every device) and hand the memory back to the driver; returns the
    bytes released.  The pool keeps GB-sized buffers alive across epochs on purpose (allocating them costs more than an epoch's
    reverse sweep); call this when the process moves on to something else on the same GPU - an evaluation render, a larger
    scene.  lease_cache calls it by itself when an allocation fails."""
    freed = 0
    for k in [k for k in _POOL if device is None or k[0] == str(device)]:
        freed += k[1] * len(_POOL[k])
        del _POOL[k]
    if freed:
        torch.cuda.empty_cache()
    return freed


def lease_cache(nbytes: int, device, force: bool = False):
    """A cache buffer of nbytes for one roll-out node, or None when the budget does not allow it (the node then recomputes).
    Idle pooled buffers of OTHER sizes count as held - they are device memory outside torch's caching allocator - and are
    dropped when they are what stands in the way (an epoch at another N or S left them behind)."""
    nbytes = int(nbytes)
    if nbytes <= 0:
        return None
    key = (str(device), nbytes)
    if not force:
        budget = act_cache_budget(device)
        others = sum(k[1] * len(v) for k, v in _POOL.items() if k[0] == key[0] and k != key)
        live = _ACT_LIVE.get(key[0], 0)
        if live + nbytes > budget:
            return None
        if live + others + nbytes > budget:
            for k in [k for k in _POOL if k[0] == key[0] and k != key]:
                del _POOL[k]
    try:
        return _Lease(nbytes, device, True)
    except torch.cuda.OutOfMemoryError:
        # the budget was sampled once, at the first call; what is free NOW decides (another workload, another process on the
        # GPU): give the idle buffers back and try once more - and without the memory the node recomputes, as over budget
        trim_pool(device)
        try:
            return _Lease(nbytes, device, True)
        except torch.cuda.OutOfMemoryError:
            if force:
                raise
            return None


class _Lease(object):
    """A cache buffer held by one roll-out node: back to the pool after the node's backward pass; if the node is dropped
    without one (inference, an abandoned graph) the buffer just dies with it."""

    def __init__(self, nbytes: int, device, counted: bool):
        key = (str(device), int(nbytes))
        free = _POOL.get(key)
        self.key, self.counted = key, counted
        self.t = None           # (an allocation that raises leaves a half-built object behind: __del__ must find the attribute)
        self.t = free.pop() if free else torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        if counted:
            _ACT_LIVE[key[0]] = _ACT_LIVE.get(key[0], 0) + int(nbytes)

    def release(self):
        if self.t is not None:
            free = _POOL.setdefault(self.key, [])
            if len(free) < _POOL_CAPS.get(self.key, _POOL_CAP[0]):
                free.append(self.t)
            self._forget()

    def _forget(self):
        if self.t is not None and self.counted:
            _ACT_LIVE[self.key[0]] = _ACT_LIVE.get(self.key[0], 0) - self.key[1]
        self.t = None

    def __del__(self):
        self._forget()


_ZEROS = {}


def _zeros(count: int, device) -> torch.Tensor:
    """A cached read-only zero vector (stands in for a gradient autograd did not produce)."""
    key = (count, str(device))
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(count, dtype=torch.float32, device=device)
    return _ZEROS[key]


class _ShardLink(object):
    """nm_comm of one sharded roll-out: the two collectives the library calls back for (include/neuma_hip.h, "Particle-sharded
    roll-out"), issued through torch.distributed on views of the roll-out's shard workspace.  The library enqueues
    everything else itself; the callbacks run on the host, in program order, while it does."""

    def __init__(self, exchange, ws: torch.Tensor):
        import torch.distributed as dist
        self.ws, self.group, self.world = ws, exchange.group, exchange.world
        self.error = None
        base = ws.data_ptr()

        def view(ptr, count, dtype):
            off, nbytes = int(ptr) - base, 4 * int(count)
            if off < 0 or off + nbytes > ws.numel():
                raise ValueError(f"collective buffer [{off}, {off + nbytes}) lies outside the roll-out's shard workspace of {ws.numel()} bytes")
            return ws[off:off + nbytes].view(dtype)

        def all_gather(user, send, recv, count, stream):
            try:
                dist.all_gather_into_tensor(view(recv, count * self.world, torch.int32), view(send, count, torch.int32), group=self.group)
                return 0
            except Exception as e:      # an exception must not unwind through the C frames
                self.error = e
                return 1

        def all_reduce(user, buf, count, stream):
            try:
                dist.all_reduce(view(buf, count, torch.float32), op=dist.ReduceOp.SUM, group=self.group)
                return 0
            except Exception as e:
                self.error = e
                return 1

        def exchange_peers(user, send, recv, count, peers, stream):
            # transport of the tests (gloo has no device send / recv): every rank's buffer is all-gathered and the peers' rows are
            # copied behind one another - the same bytes in the same places as ncclSend / ncclRecv leave them
            try:
                mine = view(send, count, torch.float32)
                rows = torch.empty(self.world, int(count), dtype=torch.float32, device=mine.device)
                dist.all_gather_into_tensor(rows.view(-1), mine, group=self.group)
                k = 0
                for q in range(self.world):
                    if q != exchange.rank and (int(peers) >> q) & 1:
                        view(int(recv) + 4 * k * int(count), count, torch.float32).copy_(rows[q])
                        k += 1
                return 0
            except Exception as e:
                self.error = e
                return 1

        rccl = exchange.library_comm() if hasattr(exchange, "library_comm") else None
        peers = getattr(exchange, "peers", None)
        if rccl is not None:
            # the library's own communicator: ncclAllGather / ncclAllReduce issued from the C loop, no Python in between
            self._cbs = None
            self.comm = L.nm_comm()
            L.check(L.lib().nm_rccl_comm(rccl, C.byref(self.comm)), "nm_rccl_comm")
            self.backend = "rccl (library-owned communicator)"
        else:
            self._cbs = (L.COMM_ALL_GATHER(all_gather), L.COMM_ALL_REDUCE(all_reduce), L.COMM_EXCHANGE_PEERS(exchange_peers))      # keep the thunks alive
            self.comm = L.nm_comm(exchange.world, exchange.rank, self._cbs[0], self._cbs[1], None, self._cbs[2], L.COMM_ALL_RANKS)
            self.backend = "torch.distributed callbacks"
        # neighbour-only exchange of the shared blocks (GridExchange.peers, derived with the frame-level capacities), else all-reduce
        self.comm.peers = L.COMM_ALL_RANKS if peers is None else int(peers)
        if peers is not None:
            self.backend += f", shared blocks swapped with ranks {[q for q in range(exchange.world) if (int(peers) >> q) & 1]}"
        try:
            exchange.link_backend = self.backend
        except AttributeError:
            pass

    def check(self, rc: int, what: str):
        if rc and self.error is not None:
            e, self.error = self.error, None
            raise L.NeumaHipError(f"{what}: collective failed: {type(e).__name__}: {e}") from e
        L.check(rc, what)


class _Rollout(autograd.Function):

    @staticmethod
    def forward(ctx, model: MPMModel, statics: MPMStatics, substeps: int, alpha: float, cache_blocks: int, svd_adjoint: int,
                x, v, C_, F, e0, e1, e2, p0, p1, p2):
        lib = L.lib()
        dev = x.device
        n = x.size(0)
        S = int(substeps)
        states = torch.empty(S + 1, 33 * n, dtype=torch.float32, device=dev)   # x|v|C|F|stress per record
        # record 0 = the inputs, packed x|v|C|F by one concatenation kernel
        torch.cat([t.detach().float().reshape(-1) for t in (x, v, C_, F)], out=states[0][:24 * n])
        we = [t.detach().float().contiguous() for t in (e0, e1, e2)]
        wp = [t.detach().float().contiguous() for t in (p0, p1, p2)]
        ws_bytes = int(lib.nm_rollout_workspace(n, S))
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        ex = model.exchange
        # grid cache: only when a backward pass can follow (ground-truth / inference roll-outs skip it)
        cache_blocks = int(cache_blocks) if any(ctx.needs_input_grad) else 0
        gc_bytes = int(lib.nm_rollout_gridcache_bytes(S, cache_blocks)) if (cache_blocks > 0 and ex is None) else 0
        gcache = torch.empty(gc_bytes, dtype=torch.uint8, device=dev) if gc_bytes > 0 else None
        # SVD cache (U, sigma, V of both nets' inputs per substep, 168 B/particle/substep) and activation cache (1.2 KB): only
        # when a backward pass can follow, and only while the caches of all live roll-out nodes together stay inside the
        # budget (act_cache_budget) - a BPTT loop over hundreds of frames falls back to the recompute (the reference's memory
        # profile) instead of running out of memory
        svdc = None
        if _SVD_CACHE and n > 0 and any(ctx.needs_input_grad):
            svdc = lease_cache(int(lib.nm_rollout_svdcache_bytes(n, S)), dev)
        actc = None
        if _ACT_CACHE != '0' and n > 0 and any(ctx.needs_input_grad):
            actc = lease_cache(int(lib.nm_rollout_actcache_bytes(n, S)), dev, force=_ACT_CACHE == '1')
        cfg = L.nm_rollout_cfg(S, float(alpha), cache_blocks if gcache is not None else 0, 0, int(svd_adjoint),
                               L.ptr(svdc.t) if svdc is not None else None, L.ptr(actc.t) if actc is not None else None)
        ctx.svdc, ctx.actc = svdc, actc
        if ex is not None:
            return _Rollout._forward_sharded(ctx, lib, model, ex, statics, S, float(alpha), int(svd_adjoint), n, states, we, wp, ws,
                                             ws_bytes, svdc, actc)
        st = statics.c_struct()
        mle = L.nm_mlp(*[L.ptr(t) for t in we])
        mlp = L.nm_mlp(*[L.ptr(t) for t in wp])
        if _FWD_PAIR is not None and not _FWD_PAIR_SET[0]:
            lib.nm_rollout_set_forward_pair(0 if _FWD_PAIR == '0' else 1)
            _FWD_PAIR_SET[0] = True
        L.check(lib.nm_rollout_forward(model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp), L.ptr(states),
                                       L.ptr(gcache) if gcache is not None else None, L.ptr(ws), ws_bytes, L.stream_ptr(dev)),
                "nm_rollout_forward")
        ctx.model, ctx.statics, ctx.S, ctx.alpha, ctx.n = model, statics, S, float(alpha), n
        ctx.set_materialize_grads(False)      # (backward tells an absent gradient from a zero one: nm_rollout_cfg.last_gF_zero)
        ctx.svd_adjoint = int(svd_adjoint)
        ctx.cache_blocks, ctx.gcache = int(cfg.grid_cache_blocks), gcache
        ctx.cache_status = ctx.cache_event = None
        if gcache is not None and _CACHE_STATUS:      # asynchronous read-back of the record headers: by the time the backward pass runs the
            status = torch.empty(S, dtype=torch.int32, pin_memory=True)       # host usually knows that every record is valid
            L.check(lib.nm_rollout_cache_status(L.ptr(gcache), C.byref(cfg), C.c_void_p(status.data_ptr()), L.stream_ptr(dev)),
                    "nm_rollout_cache_status")
            ev = torch.cuda.Event()
            ev.record()
            ctx.cache_status, ctx.cache_event = status, ev
        ctx.save_for_backward(states, *we, *wp)
        last = states[S]
        return (last[:3 * n].view(n, 3), last[3 * n:6 * n].view(n, 3), last[6 * n:15 * n].view(n, 3, 3),
                last[15 * n:24 * n].view(n, 3, 3))

    @staticmethod
    def _forward_sharded(ctx, lib, model, ex, statics, S, alpha, svd_adjoint, n, states, we, wp, ws, ws_bytes, svdc, actc):
        """This rank's share of the particles (model.shard(group)): the library runs the whole S-substep loop, phases and
        collectives, and calls back for the latter (nm_rollout_forward_sharded): one all-gather per roll-out (the frame's
        exchange list is negotiated at the first substep) and one all-reduce per substep and direction."""
        dev = states.device
        st = statics.c_struct()
        def probe_start_state():
            # one p2g of the inputs with zero stress: the handle then holds the blocks the start state touches
            r0 = states[0]
            r0[24 * n:].zero_()
            base = r0.data_ptr()
            cur = L.nm_particles(base, base + 12 * n, base + 24 * n, base + 60 * n, base + 96 * n)
            L.check(lib.nm_mpm_p2g(model.handle(), n, C.byref(st), C.byref(cur), L.stream_ptr(dev)), "nm_mpm_p2g")

        probed = False
        if ex.cap is None or ex.cap_shared is None:
            probe_start_state()          # capacities from the blocks the start state touches (two host reads)
            probed = True
            ex._ensure_sized()
        if ex.needs_frame_sizing() if hasattr(ex, "needs_frame_sizing") else (ex.cap_dil is None or ex.cap_frame is None):
            # The frame-level capacities are sized from the grid the handle holds NOW: it must be the start state's, whatever
            # the handle did before (model.shard(cap=..., cap_shared=...) with a fused roll-out as the first operation used to
            # size them from an empty grid: 64 each, and every later frame overflowed - ADVICE r3)
            if not probed:
                probe_start_state()
            ex.size_frame_lists()
        cap_rec, cap, cap_shared = int(ex.cap), int(ex.cap_dil), int(ex.cap_frame)
        gcache = torch.empty(int(lib.nm_rollout_gridcache_bytes(S, cap_rec)), dtype=torch.uint8, device=dev)
        sws_bytes = int(lib.nm_rollout_shard_workspace(ex.world, cap, cap_shared, S))
        sws = torch.empty(sws_bytes, dtype=torch.uint8, device=dev)
        link = _ShardLink(ex, sws)
        cfg = L.nm_rollout_cfg(S, alpha, cap_rec, 0, svd_adjoint, L.ptr(svdc.t) if svdc is not None else None,
                               L.ptr(actc.t) if actc is not None else None)
        mle = L.nm_mlp(*[L.ptr(t) for t in we])
        mlp = L.nm_mlp(*[L.ptr(t) for t in wp])
        link.check(lib.nm_rollout_forward_sharded(model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp), L.ptr(states),
                                                  L.ptr(gcache), L.ptr(ws), ws_bytes, C.byref(link.comm), cap, cap_shared, L.ptr(sws),
                                                  sws_bytes, L.stream_ptr(dev)), "nm_rollout_forward_sharded")
        ex.watch(lib, sws, dev)          # status word -> pinned host memory; ex.check() raises on a capacity overflow
        ctx.model, ctx.statics, ctx.S, ctx.alpha, ctx.n = model, statics, S, alpha, n
        ctx.set_materialize_grads(False)
        ctx.svd_adjoint = svd_adjoint
        ctx.cache_blocks, ctx.gcache = cap_rec, gcache
        ctx.cache_status = ctx.cache_event = None
        ctx.shard = (ex, sws, sws_bytes, cap, cap_shared)
        ctx.save_for_backward(states, *we, *wp)
        last = states[S]
        return (last[:3 * n].view(n, 3), last[3 * n:6 * n].view(n, 3), last[6 * n:15 * n].view(n, 3, 3),
                last[15 * n:24 * n].view(n, 3, 3))

    @staticmethod
    def backward(ctx, gx, gv, gC, gF):
        lib = L.lib()
        states, e0, e1, e2, p0, p1, p2 = ctx.saved_tensors
        dev = states.device
        n, S = ctx.n, ctx.S
        # incoming gradients packed x|v|C|F by one concatenation kernel (absent ones are zeros, kept per size)
        parts = []
        for g, cnt in ((gx, 3 * n), (gv, 3 * n), (gC, 9 * n), (gF, 9 * n)):
            parts.append(g.float().reshape(-1) if g is not None else _zeros(cnt, dev))
        glast = torch.cat(parts)
        gfirst = torch.empty(24 * n, dtype=torch.float32, device=dev)
        gwe = torch.empty(sum(_WSZ), dtype=torch.float32, device=dev)
        gwp = torch.empty(sum(_WSZ), dtype=torch.float32, device=dev)
        ws_bytes = int(lib.nm_rollout_workspace(n, S))
        ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=dev)
        gcache = ctx.gcache
        verified = 0
        if gcache is not None and ctx.cache_event is not None:
            # the record headers travel back right behind the forward sweep.  A host that gets here before the device has
            # finished that sweep waits for it (the device still has whatever was enqueued in between - the frame's render -
            # or idles for the few launches it takes to get the reverse sweep going): the unverified sweep it would otherwise
            # run costs four extra launches per substep (+20 us each at 100k particles, 0.4 ms per 20-substep node)
            if _CACHE_WAIT and not ctx.cache_event.query():
                ctx.cache_event.synchronize()
            verified = int(bool((ctx.cache_status >= 0).all()))
        svdc, actc = getattr(ctx, "svdc", None), getattr(ctx, "actc", None)
        # (no gradient arrived for F of the last record: the last substep's plasticity adjoint would propagate zeros - the
        #  library leaves that launch out, nm_rollout_cfg.last_gF_zero; the sharded sweep ignores the word)
        cfg = L.nm_rollout_cfg(S, ctx.alpha, ctx.cache_blocks, verified, ctx.svd_adjoint, L.ptr(svdc.t) if svdc is not None else None,
                               L.ptr(actc.t) if actc is not None else None, 0, 1 if gF is None else 0)
        st = ctx.statics.c_struct()
        mle = L.nm_mlp(L.ptr(e0), L.ptr(e1), L.ptr(e2))
        mlp = L.nm_mlp(L.ptr(p0), L.ptr(p1), L.ptr(p2))
        if getattr(ctx, "shard", None) is not None:
            ex, sws, sws_bytes, cap, cap_shared = ctx.shard
            link = _ShardLink(ex, sws)
            link.check(lib.nm_rollout_backward_sharded(ctx.model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp),
                                                       L.ptr(states), L.ptr(gcache), L.ptr(glast), L.ptr(gfirst), L.ptr(gwe), L.ptr(gwp),
                                                       L.ptr(ws), ws_bytes, C.byref(link.comm), cap, cap_shared, L.ptr(sws), sws_bytes,
                                                       L.stream_ptr(dev)), "nm_rollout_backward_sharded")
        else:
            L.check(lib.nm_rollout_backward(ctx.model.handle(), n, C.byref(cfg), C.byref(st), C.byref(mle), C.byref(mlp),
                                            L.ptr(states), L.ptr(gcache) if gcache is not None else None, L.ptr(glast), L.ptr(gfirst),
                                            L.ptr(gwe), L.ptr(gwp), L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "nm_rollout_backward")
        ctx.gcache = None
        for lease in (svdc, actc):
            if lease is not None:
                lease.release()
        ctx.svdc = ctx.actc = None
        torch.nan_to_num_(gfirst, 0.0, 0.0, 0.0)   # interface.py:65-74 at the boundary of the fused node
        a, b = _WSZ[0], _WSZ[0] + _WSZ[1]
        return (None, None, None, None, None, None,
                gfirst[:3 * n].view(n, 3), gfirst[3 * n:6 * n].view(n, 3), gfirst[6 * n:15 * n].view(n, 3, 3),
                gfirst[15 * n:].view(n, 3, 3),
                gwe[:a].view(64, 13), gwe[a:b].view(64, 64), gwe[b:].view(9, 64),
                gwp[:a].view(64, 13), gwp[a:b].view(64, 64), gwp[b:].view(9, 64))


class MPMFusedDiffSim(nn.Module):
    """sim(statics, x, v, C, F) -> (x, v, C, F) after `substeps` substeps, constitutive nets included.

    grid_cache: "auto" (default) keeps each substep's touched grid blocks for the backward pass (the reverse sweep then
    restores the grid instead of re-running p2g); the per-substep capacity is sized once from the number of blocks the
    first roll-out touches (x1.5 + 64; one host sync) - a substep that outgrows it falls back to the recompute on its
    own.  An int fixes the capacity in 4x4x4-node blocks; 0 / None disables the cache (the reference's behaviour)."""

    def __init__(self, model: MPMModel, elasticity: nn.Module, plasticity: nn.Module, substeps: int, grid_cache="auto",
                 reorder="auto", svd_adjoint: str = "reference") -> None:
        super().__init__()
        # svd_adjoint: how dL/dF flows through R = U V^T of the constitutive nets in the reverse sweep.
        #   "reference" (default)  the reference's gradient: warp's adj_svd3 behind warp/svd.py:41-57, whose 1/(s_j^2 - s_i^2) is
        #                          clamped at 1e-6 - the rotation path vanishes where singular values coincide (F = I, the state every
        #                          roll-out starts from) and is the exact polar derivative elsewhere;
        #   "polar"                the exact derivative U [(A - A^T) / (s_i + s_j)] V^T everywhere (an improvement, not parity).
        if svd_adjoint not in L.SVD_ADJOINT:
            raise ValueError(f"svd_adjoint must be one of {sorted(L.SVD_ADJOINT)}")
        self.svd_adjoint = svd_adjoint
        self.model, self.elasticity, self.plasticity, self.substeps = model, elasticity, plasticity, int(substeps)
        self.grid_cache = grid_cache
        self._cache_blocks = None if grid_cache == "auto" else int(grid_cache or 0)
        # reorder: the scatter kernels are ~20x faster when consecutive particles are spatial neighbours, and NeuMA's
        # data preparation shuffles the particles (tune/utils.py:270-272).  "auto": on the first call measure how many
        # consecutive particles have adjacent stencil origins; below 80 % the roll-out runs on a Hilbert-sorted copy
        # (inputs gathered, outputs scattered back - plain differentiable indexing, so callers never see the order).
        # True / False force it.  The permutation is fixed at the first call (particles move a fraction of a cell per step).
        self.order = ParticleOrder(int(model.constant.num_grids), reorder)

    @property
    def reorder(self):
        return self.order.mode

    @reorder.setter
    def reorder(self, mode):
        self.order.mode = mode
        self.order.perm = None

    @property
    def _perm(self):
        return self.order.perm

    def order_quality(self, x: Tensor) -> float:
        return self.order.quality(x)

    def grid_cache_blocks(self) -> int:
        return int(self._cache_blocks or 0)

    def forward(self, statics: MPMStatics, x: Tensor, v: Tensor, C_: Tensor, F: Tensor):
        e = self.elasticity.effective_weights()
        p = self.plasticity.effective_weights()
        if self.order.active(x):
            pm, inv = self.order.perm, self.order.inv
            out = _Rollout.apply(self.model, self.order.statics(statics), self.substeps, self.plasticity.alpha,
                                 self.grid_cache_blocks(), L.SVD_ADJOINT[self.svd_adjoint], x[pm], v[pm], C_[pm], F[pm], *e, *p)
            out = tuple(o[inv] for o in out)
        else:
            out = _Rollout.apply(self.model, statics, self.substeps, self.plasticity.alpha, self.grid_cache_blocks(),
                                 L.SVD_ADJOINT[self.svd_adjoint], x, v, C_, F, *e, *p)
        if self._cache_blocks is None and self.model.exchange is None:      # first roll-out: size the cache from what the scene touches
            blocks, _ = self.model.grid_stats()
            self._cache_blocks = int(1.5 * blocks) + 64
        return out
