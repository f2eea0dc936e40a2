"""Particle-sharded simulation: one process per GPU, every rank steps a fixed subset of the particles.

No reference counterpart (the reference is single-device).  Ownership is by particle, not by region:
rank r keeps a contiguous range of the (Hilbert-ordered) particle list for the whole roll-out, so nothing migrates and
autograd needs no routing.  Each rank scatters its own particles into its own block-sparse grid; the only data-path
collective of a substep is an all-reduce (sum) over the 4x4x4-node blocks that MORE than one rank touches - the
overlap of the ranks' halos - once for {mv, m} in the forward pass and once for the adjoint of the node velocities in
the backward pass (include/neuma_hip.h, "Particle-sharded substep"; kernels in csrc/nm_shard.hip).

    model.shard(group)          every MPMModel.forward / backward on this model is now the sharded substep
    shard_range(N, world, r)    the particle rows rank r owns
    gather_rows(t, N, group)    local rows -> all rows (differentiable; the rasterizer needs every particle)
    reduce_param_grads(params)  sum the LoRA gradients of the ranks after loss.backward()
"""
import ctypes as C
import os
from typing import Iterable, Optional, Tuple

import torch

from .. import _lib as L


def shard_range(num_particles: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of the particle list that `rank` owns: contiguous, sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    return (num_particles * rank) // world, (num_particles * (rank + 1)) // world


def size_with_slack(count: int) -> int:
    """Capacity chosen from an observed count: 1.5x + 64 (the same rule as the grid cache)."""
    return int(1.5 * int(count)) + 64


STATUS_TEXT = {1: "a rank touched more grid blocks than the list capacity `cap`",
               2: "more blocks are shared between ranks than `cap_shared`",
               4: "a substep touched more blocks than its grid cache record holds",
               8: "a particle left the block neighbourhood its rank announced at the frame's first substep (it moved more than "
                  "a block - 4 grid cells - within one roll-out): use fewer substeps per roll-out node",
               16: "a grid block is shared with a rank that is not among the peers of the neighbour-only exchange (the bodies' "
                   "neighbourhoods have come to touch since the peers were derived)"}


def dilate_blocks_host(ids, nb: int):
    """Host statement of nm_mpm_dilated_list (csrc/nm_shard.hip): the 27-neighbourhood, in 4x4x4-node blocks, of the blocks
    `ids` (block id = (bi * nb + bj) * nb + bk), clipped to the nb^3 blocks of the grid; returned as a sorted array (the
    device list is unordered).  This is what a rank announces once per frame in the fused sharded roll-out: every block one of
    its particles can reach while it moves less than one block.  The frame's exchange list is then
    shared_blocks_host(all-gathered neighbourhoods): two ranks can both touch only blocks that lie in both neighbourhoods."""
    import numpy as np
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    bi, bj, bk = ids // (nb * nb), (ids // nb) % nb, ids % nb
    out = set()
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for dk in (-1, 0, 1):
                i, j, k = bi + di, bj + dj, bk + dk
                ok = (i >= 0) & (i < nb) & (j >= 0) & (j < nb) & (k >= 0) & (k < nb)
                out.update(((i[ok] * nb + j[ok]) * nb + k[ok]).tolist())
    return np.asarray(sorted(out), dtype=np.int32)


# ---------------------------------------------------------------- replicate or shard the simulation? (bench.py --shard-sim auto)
# Measured on 1x MI355X, fused roll-out, forward + backward, microseconds per substep at N particles (tools/exp_shard_overhead.py
# <workload> <N>; profiles/r04_shard_overhead.txt - round 4: monotone since the scatter kernels cut a chunk of the particle
# list at its jumps instead of falling back to per-wave boxes, which had cost 84 us per scatter at 25 000 particles).  Between
# the points: linear in N.  The small sizes are latency floors - a substep is ~7 dependent launches - which is why dividing
# 100k particles by 8 buys 2.4x, not 8x.  bench.py --shard-sim auto no longer decides from this table when it runs on several
# GPUs: it measures both sizes and the all-reduce at start-up (shard_cost_model(measured=...)); the table is the fallback.
SUBSTEP_US = ((12_500, 94.0), (25_000, 117.0), (50_000, 147.5), (100_000, 223.0), (1_000_000, 1700.0))
SHARD_MARGIN = 0.10
MACHINERY_US = 7.0       # measured with a one-rank RCCL group and the library-owned communicator (csrc/nm_rccl.hip): 2 pack launches
#                          + 2 ncclAllReduce per substep, fwd + bwd (profiles/r04_shard_overhead.txt: +4.5 forward, +5..7 both; with
#                          the Python callback table of round 3 it was 23; at 1 M particles / 256^3 the packs grow to ~100 us)
# NOT measured (no multi-GPU box so far): latency of one all-reduce of <= 1 MiB over xGMI at `world` ranks.  Assumption: a ring /
# tree step costs ~6 us and RCCL's launch ~10 us.  NEUMA_XGMI_ALLREDUCE_US overrides the per-collective figure.
def allreduce_us(world: int) -> float:
    import math
    import os
    env = os.environ.get("NEUMA_XGMI_ALLREDUCE_US")
    if env:
        return float(env)
    return 0.0 if world <= 1 else 10.0 + 6.0 * math.ceil(math.log2(world)) * 2


def substep_us(n: int) -> float:
    pts = SUBSTEP_US
    if n <= pts[0][0]:
        return pts[0][1]
    for (n0, t0), (n1, t1) in zip(pts, pts[1:]):
        if n <= n1:
            return t0 + (t1 - t0) * (n - n0) / (n1 - n0)
    return pts[-1][1] * n / pts[-1][0]


def shard_cost_model(num_particles: int, world: int, substeps: int, measured: Optional[dict] = None) -> dict:
    """Estimated simulation time of one frame (S substeps, forward + backward) on `world` GPUs with the simulation replicated
    (every rank steps all particles) and particle-sharded (every rank steps N / world of them and pays, per substep and
    direction, one all-reduce of the exchange blocks, plus per frame the all-gather of x and F, the all-gather of the block
    neighbourhoods and the reduction of the LoRA gradients).  `shard` = the sharded estimate is smaller by at least
    SHARD_MARGIN (10 %).
    measured (calibrate(), bench.py at start-up): {"substep_us_full", "substep_us_shard", "allreduce_us", "machinery_us"} taken on
    THIS box at the real world size replace the table / the assumed all-reduce latency; without it the model runs on the
    one-GPU table below and an ASSUMED xGMI latency (`allreduce_us_assumed`), which is all that exists until a multi-GPU box
    has been measured."""
    m = measured or {}
    rep = substeps * float(m.get("substep_us_full", substep_us(num_particles)))
    ar = float(m["allreduce_us"]) if "allreduce_us" in m else allreduce_us(world)
    mach = float(m.get("machinery_us", MACHINERY_US))
    per_frame = 4.0 * ar if world > 1 else 0.0                     # x, F, neighbourhoods, parameter gradients
    sh = substeps * (float(m.get("substep_us_shard", substep_us(-(-num_particles // world)))) + mach + 2.0 * ar) + per_frame
    out = {"replicated_us": rep, "sharded_us": sh, "shard": world > 1 and sh < (1.0 - SHARD_MARGIN) * rep,
           "inputs": "measured at start-up" if measured else "one-GPU table + assumed all-reduce latency"}
    out["allreduce_us" if "allreduce_us" in m else "allreduce_us_assumed"] = ar
    if measured:
        out["measured"] = dict(measured)
    return out


def time_all_reduce_us(group, device, count: int = 1 << 16, reps: int = 20, rccl=None) -> float:
    """Mean microseconds of one in-place all-reduce (sum) of `count` floats over `group` - the latency term of the cost model,
    measured at the real world size: through the library's communicator when there is one (nm_rccl_time_all_reduce), else
    through torch.distributed with HIP events (gloo: host time).  Collective."""
    import time
    import torch.distributed as dist
    buf = torch.zeros(count, dtype=torch.float32, device=device)
    if rccl is not None:
        us = C.c_float(0.0)
        L.check(L.lib().nm_rccl_time_all_reduce(rccl, L.ptr(buf), count, 5, reps, C.byref(us), L.stream_ptr(device)), "nm_rccl_time_all_reduce")
        return float(us.value)
    for _ in range(5):
        dist.all_reduce(buf, group=group)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf, group=group)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    return 1e6 * (time.perf_counter() - t0) / reps


def time_exchange_peers_us(device, rank: int, world: int, count: int = 1 << 16, reps: int = 20, rccl=None) -> Optional[float]:
    """Mean microseconds of one neighbour-only exchange of `count` floats with the ranks next to this one in rank order (r - 1,
    r + 1: what contiguous ranges of a space-filling particle order mostly share blocks with) through the library's
    communicator - the figure the start-up calibration holds against the all-reduce's.  None without a library communicator
    (the callback table is a test transport: its time means nothing).  Collective."""
    if rccl is None:
        return None
    peers = sum(1 << q for q in (rank - 1, rank + 1) if 0 <= q < world)
    send = torch.zeros(count, dtype=torch.float32, device=device)
    recv = torch.zeros(2 * count, dtype=torch.float32, device=device)
    us = C.c_float(0.0)
    # (an RCCL without the point-to-point entry points answers with an error - on every rank alike: no exchange mode to offer)
    if L.lib().nm_rccl_time_exchange_peers(rccl, L.ptr(send), L.ptr(recv), count, peers, 5, reps, C.byref(us), L.stream_ptr(device)):
        return None
    return float(us.value)


def peers_mask(adj, rank: int) -> int:
    """nm_comm.peers of `rank` from its adjacency row (adj[q] != 0: rank q shares a block of the neighbourhoods with it)."""
    return sum(1 << q for q, a in enumerate(adj) if a and q != rank)


def peer_ranks_host(lists, rank: int):
    """Host statement of nm_mpm_peer_ranks: lists[q] = the (dilated) block ids rank q announced; adj[q] = 1 iff one of them lies
    in `rank`'s own list.  Symmetric in (rank, q)."""
    mine = set(int(b) for b in lists[rank])
    return [1 if (q != rank and any(int(b) in mine for b in lists[q])) else 0 for q in range(len(lists))]


def exchange_peers_host(bufs, peers):
    """What the neighbour-only exchange leaves in every rank's buffer: bufs[r] (array per rank, same layout) summed over r and its
    peers in ascending rank order.  For a slot whose owners are all among each other's peers this equals the all-reduce."""
    out = []
    for r, mine in enumerate(bufs):
        acc = None
        for q in range(len(bufs)):
            if q == r or (peers[r] >> q) & 1:
                acc = bufs[q].copy() if acc is None else acc + bufs[q]
        out.append(acc)
    return out


def explain_status(bits: int) -> str:
    return "; ".join(text for bit, text in STATUS_TEXT.items() if bits & bit)


def shared_blocks_host(gathered, cap: int, nblocks: int, cap_shared: int, rank: Optional[int] = None):
    """Host statement of the rule nm_mpm_shared_blocks implements on the device (csrc/nm_shard.hip) - the specification the
    kernels are tested against, and what the CPU (gloo) tests run the exchange bookkeeping with.

    gathered: int array (world, 1 + cap), row r = [count_r, id_0 .. id_{cap-1}] as all-gathered from the ranks.  An entry
    counts if it lies inside the rank's (capacity-clipped) count and names a block in [0, nblocks).  A block is SHARED if
    at least two entries name it; the shared list orders the blocks by the first position at which they appear in the
    flattened gathered array.  Every rank evaluates the same rule on the same array, hence the same list in the same
    order everywhere, with no coordination.  Returns (ids (n,), mine (n,) bool: `rank` lists the block itself, bits) with
    bits = 1 if some rank's count exceeds cap, | 2 if more than cap_shared blocks are shared (the list is then cut)."""
    import numpy as np
    g = np.asarray(gathered).reshape(-1, 1 + cap)
    world = g.shape[0]
    first, count, lists = {}, {}, []
    bits = 0
    for r in range(world):
        n = int(g[r, 0])
        if n > cap:
            bits |= 1
        ids = [int(b) for b in g[r, 1:1 + min(n, cap)] if 0 <= int(b) < nblocks]
        lists.append(set(ids))
        for j, b in enumerate(g[r, 1:1 + min(n, cap)]):
            b = int(b)
            if not 0 <= b < nblocks:
                continue
            count[b] = count.get(b, 0) + 1
            first.setdefault(b, r * (1 + cap) + 1 + j)
    shared = sorted((b for b in count if count[b] >= 2), key=lambda b: first[b])
    if len(shared) > cap_shared:
        bits |= 2
        shared = shared[:cap_shared]
    mine = [rank is not None and b in lists[rank] for b in shared]
    return np.asarray(shared, dtype=np.int32), np.asarray(mine, dtype=bool), bits


def exchange_blocks_host(values, shared_ids, mine, all_reduce):
    """Host statement of nm_mpm_blocks_pack -> all-reduce -> nm_mpm_blocks_unpack.  values: (nblocks, 64, 4) float array of
    this rank's block contents, modified in place: blocks of `shared_ids` this rank lists (`mine`) are replaced by the sum
    over the ranks; the rank contributes zeros for shared blocks it does not list and never touches any other block.
    all_reduce: callable summing a float array over the ranks in place (dist.all_reduce on a tensor view)."""
    import numpy as np
    buf = np.zeros((len(shared_ids), 64, 4), dtype=values.dtype)
    for i, (b, m) in enumerate(zip(shared_ids, mine)):
        if m:
            buf[i] = values[b]
    all_reduce(buf)
    for i, (b, m) in enumerate(zip(shared_ids, mine)):
        if m:
            values[b] = buf[i]
    return values


class ShardTape(object):
    """`tape` of a sharded substep: the grid cache record plus the list of blocks that were summed over the ranks.
    Filled by MPMModel.forward, consumed by MPMModel.backward."""

    def __init__(self, generation: int) -> None:
        self.generation = generation
        self.cap = 0
        self.buf: Optional[torch.Tensor] = None
        self.shared: Optional[torch.Tensor] = None


class GridExchange(object):
    """Per-model state of the sharded substep: capacities, exchange buffers, the status word."""

    def __init__(self, model, group=None, cap: Optional[int] = None, cap_shared: Optional[int] = None,
                 cap_dil: Optional[int] = None, cap_frame: Optional[int] = None) -> None:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise L.NeumaHipError("model.shard() needs an initialised torch.distributed process group")
        self.model, self.group = model, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.cap = int(cap) if cap else None                    # blocks per rank list / grid cache record
        self.cap_shared = int(cap_shared) if cap_shared else None
        self.cap_dil = int(cap_dil) if cap_dil else None        # fused roll-out: blocks in a rank's announced neighbourhood ...
        self.cap_frame = int(cap_frame) if cap_frame else None  # ... and in the frame's exchange list (size_frame_lists)
        self.device = model.device
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.generation = 0
        self._mine = self._gathered = self._ws = self._buf = self._scratch_shared = None
        self._watched = []          # (pinned int32, event) of fused sharded roll-outs whose status word has not been examined
        # fused roll-out: which ranks this one swaps exchange buffers with (bit q = rank q), or None = all-reduce over the world.
        # NEUMA_SHARD_EXCHANGE: "allreduce" (default), "peers" (neighbour-only: derived by size_frame_lists from the probe)
        self.exchange_mode = os.environ.get("NEUMA_SHARD_EXCHANGE", "allreduce")
        self.peers: Optional[int] = None
        self._rccl = None           # library-owned RCCL communicator (library_comm): None = not looked up yet, False = not available
        create_library_comm(group, self.device)      # COLLECTIVE, here and only here for a sharded model: model.shard() is called by every rank

    def library_comm(self):
        """The library's own RCCL communicator for this group (library_comm_for: one per process group, shared with the per-frame
        collectives).  With it the fused sharded roll-out issues its collectives from the C loop on the stream
        (rollout._ShardLink); None when the group's backend is not nccl (the gloo tests keep the callback table) or
        NEUMA_COMM=python asks for the callbacks."""
        if self._rccl is None:
            self._rccl = library_comm_for(self.group, self.device) or False
        elif self._rccl and library_comm_for(self.group, self.device) is None:
            self._rccl = False          # released behind this object's back (close_library_comms, group destroyed)
        return self._rccl or None

    def close(self) -> None:
        """Forget the communicator (it belongs to the process group's cache: close_library_comms() releases it)."""
        self._rccl = False

    # -- sizing (first substep only: two host reads + two tiny collectives)
    def _ensure_sized(self) -> None:
        import torch.distributed as dist
        lib, h, s = L.lib(), self.model.handle(), self.model._stream()
        if self.cap is None:
            nblocks = (int(self.model.constant.num_grids) + 2 + 3) // 4
            probe = torch.empty(1 + nblocks ** 3, dtype=torch.int32, device=self.device)
            L.check(lib.nm_mpm_active_list(h, L.ptr(probe), nblocks ** 3, s), "nm_mpm_active_list")
            count = probe[:1].clone()
            dist.all_reduce(count, op=dist.ReduceOp.MAX, group=self.group)
            self.cap = size_with_slack(int(count.item()))
        if self._gathered is None:
            self._mine = torch.empty(1 + self.cap, dtype=torch.int32, device=self.device)
            self._gathered = torch.empty(self.world, 1 + self.cap, dtype=torch.int32, device=self.device)
            nbytes = int(lib.nm_mpm_shared_workspace(self.world, self.cap))
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        if self.cap_shared is None:
            upper = max(1, (self.world * self.cap) // 2)          # a shared block is in at least two lists
            probe = torch.zeros(2 + 2 * upper, dtype=torch.int32, device=self.device)
            self._find_shared(probe, upper, None)
            self.cap_shared = size_with_slack(int(probe[0].item()))
        if self._buf is None:
            self._buf = torch.empty(self.cap_shared * 64 * 4, dtype=torch.float32, device=self.device)
            self._scratch_shared = self.new_shared()

    def needs_frame_sizing(self) -> bool:
        """The fused roll-out must probe the start state first: a frame-level capacity, or the peer set of the neighbour-only
        exchange, is unknown (first roll-out, or forgotten after a status bit)."""
        return (self.cap_dil is None or self.cap_frame is None or
                (self.exchange_mode == "peers" and self.peers is None and self.world <= 32))

    def size_frame_lists(self) -> None:
        """Capacities of the fused roll-out's frame-level negotiation, from the grid the handle holds now (one probe: the
        ranks' neighbourhoods all-gathered, the exchange list selected, two host reads)."""
        import torch.distributed as dist
        if not self.needs_frame_sizing():
            return
        given = (self.cap_dil, self.cap_frame) if (self.cap_dil is not None and self.cap_frame is not None) else None
        self.cap_dil = self.cap_frame = None
        lib, h, s = L.lib(), self.model.handle(), self.model._stream()
        nb3 = ((int(self.model.constant.num_grids) + 2 + 3) // 4) ** 3
        probe_cap = min(nb3, 27 * int(self.cap))
        mine = torch.empty(1 + probe_cap, dtype=torch.int32, device=self.device)
        L.check(lib.nm_mpm_dilated_list(h, L.ptr(mine), probe_cap, s), "nm_mpm_dilated_list")
        count = mine[:1].clone()
        dist.all_reduce(count, op=dist.ReduceOp.MAX, group=self.group)
        self.cap_dil = size_with_slack(int(count.item()))
        if self.cap_dil < probe_cap:
            mine = mine[:1 + self.cap_dil].contiguous()
        else:
            self.cap_dil = probe_cap
        gathered = torch.empty(self.world, 1 + self.cap_dil, dtype=torch.int32, device=self.device)
        dist.all_gather_into_tensor(gathered.view(-1), mine, group=self.group)
        upper = max(1, (self.world * self.cap_dil) // 2)
        probe = torch.zeros(2 + 2 * upper, dtype=torch.int32, device=self.device)
        ws = torch.empty(int(lib.nm_mpm_shared_workspace(self.world, self.cap_dil)), dtype=torch.uint8, device=self.device)
        L.check(lib.nm_mpm_shared_blocks(h, L.ptr(gathered), self.world, self.cap_dil, L.ptr(probe), upper, None, L.ptr(ws),
                                         ws.numel(), s), "nm_mpm_shared_blocks")
        self.cap_frame = size_with_slack(int(probe[0].item()))
        self.peers = None
        if self.exchange_mode == "peers" and self.world <= 32:
            # neighbour-only exchange: the ranks whose neighbourhood meets this rank's, from the same gathered lists (the rule is
            # symmetric, so rank q finds this rank among ITS peers); a rank that becomes a neighbour later raises status bit 16
            adj = torch.zeros(self.world, dtype=torch.int32, device=self.device)
            L.check(lib.nm_mpm_peer_ranks(h, L.ptr(gathered), self.world, self.cap_dil, self.rank, L.ptr(adj), s), "nm_mpm_peer_ranks")
            self.peers = peers_mask(adj.tolist(), self.rank)
        if given is not None:       # (the caller's capacities stand; the probe was for the peers)
            self.cap_dil, self.cap_frame = max(given[0], self.cap_dil), max(given[1], self.cap_frame)

    def new_shared(self) -> torch.Tensor:
        return torch.empty(2 + 2 * self.cap_shared, dtype=torch.int32, device=self.device)

    # -- the exchange
    def _find_shared(self, shared: torch.Tensor, cap_shared: int, status) -> None:
        import torch.distributed as dist
        lib, h, s = L.lib(), self.model.handle(), self.model._stream()
        L.check(lib.nm_mpm_active_list(h, L.ptr(self._mine), self.cap, s), "nm_mpm_active_list")
        dist.all_gather_into_tensor(self._gathered.view(-1), self._mine, group=self.group)
        L.check(lib.nm_mpm_shared_blocks(h, L.ptr(self._gathered), self.world, self.cap, L.ptr(shared), cap_shared,
                                         L.ptr(status), L.ptr(self._ws), self._ws.numel(), s), "nm_mpm_shared_blocks")

    def find_shared(self, shared: torch.Tensor) -> None:
        self._find_shared(shared, self.cap_shared, self.status)

    def sum_blocks(self, which: int, shared: torch.Tensor) -> None:
        """All-reduce the shared blocks of {mv, m} (which = 0) or of the grid adjoint (which = 1)."""
        import torch.distributed as dist
        lib, h, s = L.lib(), self.model.handle(), self.model._stream()
        L.check(lib.nm_mpm_blocks_pack(h, which, L.ptr(shared), self.cap_shared, L.ptr(self._buf), s), "nm_mpm_blocks_pack")
        dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=self.group)
        L.check(lib.nm_mpm_blocks_unpack(h, which, L.ptr(shared), self.cap_shared, L.ptr(self._buf), s), "nm_mpm_blocks_unpack")

    # -- status
    def watch(self, lib, shard_ws: torch.Tensor, device) -> None:
        """A fused sharded roll-out (rollout._Rollout) keeps its own status word at the head of its workspace: copy it to
        pinned host memory in stream order; check() looks at it."""
        host = torch.zeros(1, dtype=torch.int32, pin_memory=True)
        L.check(lib.nm_rollout_shard_status(L.ptr(shard_ws), C.c_void_p(host.data_ptr()), L.stream_ptr(device)), "nm_rollout_shard_status")
        ev = torch.cuda.Event()
        ev.record()
        self._watched.append((host, ev))

    def _raise(self, bits: int) -> None:
        """An incomplete exchange: say which capacity it was, and forget the frame-level capacities so that the next fused
        roll-out probes them again (they are sized from a start state; cap / cap_shared are the constructor's)."""
        if bits & (1 | 2 | 8 | 16):
            self.cap_dil = self.cap_frame = self.peers = None
        hint = []
        if bits & 1:
            hint.append(f"cap={self.cap} (per-rank block list; fused roll-outs: cap_dil, re-probed at the next roll-out)")
        if bits & 2:
            hint.append(f"cap_shared={self.cap_shared} (exchange list; fused roll-outs: cap_frame, re-probed at the next roll-out)")
        if bits & 4:
            hint.append(f"cap={self.cap} (blocks per grid cache record): model.shard(group, cap=...)")
        if bits & 8:
            hint.append("fewer substeps per roll-out node (the neighbourhood is negotiated once per node)")
        if bits & 16:
            hint.append("nothing (the peer ranks of the neighbour-only exchange are derived again at the next roll-out)")
        raise L.NeumaHipError(f"sharded substep incomplete ({explain_status(bits)}); adjust: " + "; ".join(hint) +
                              ".  The gradients of that roll-out are wrong: discard them (do not step the optimizer) and re-run it")

    def check(self, wait=True) -> None:
        """Raise if any substep since the last check exceeded a capacity (one host read).  Called once per backward
        pass by MPMModel.backward and by the frame driver after the forward roll-out.
        wait=False: look only at the status words of fused roll-outs that have FINISHED - no host synchronisation.
        wait="watched" (frame driver, fused roll-outs): wait for the status words of the fused roll-outs enqueued so far - each
        is copied out right behind its FORWARD sweep, so the host waits for that sweep while the device still has the frame's
        renders and reverse sweeps queued - and raise before the caller can use the frame's gradients; the per-operator status
        word on the device is not read (no stream synchronisation).
        wait=True: everything, including the device-side word of the per-operator substeps (synchronises the stream)."""
        if wait is False or wait == "watched":
            bits, keep = 0, []
            for host, ev in self._watched:
                if wait == "watched":
                    ev.synchronize()
                if ev.query():
                    bits |= int(host[0])
                else:
                    keep.append((host, ev))
            self._watched = keep
            if bits:
                self.generation += 1
                self._raise(bits)
            return
        bits = int(self.status.item())
        for host, ev in self._watched:
            ev.synchronize()
            bits |= int(host[0])
        self._watched = []
        if self.world > 1:
            # the per-operator substeps' word is this rank's own (bit 8 / bit 4 are raised by the rank it happens to): every
            # rank must reach the same verdict, or the one that raises re-probes capacities in collectives nobody else is in.
            # (Every rank calls check(wait=True) at the same point - once per backward pass / frame; fused roll-outs have
            #  already OR-ed their word over the ranks on the device, nm_rollout_forward_sharded.)
            import torch.distributed as dist
            flags = torch.tensor([(bits >> b) & 1 for b in range(8)], dtype=torch.int32, device=self.device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            bits = sum(int(f) << b for b, f in enumerate(flags.tolist()))
        self.generation += 1
        if bits:
            self.status.zero_()
            self._raise(bits)

    def defer_check(self) -> None:
        """The caller promises to call check() itself before it uses any result of the substeps recorded so far (the
        frame driver does so once, after the backward pass, instead of stalling the GPU between the two sweeps)."""
        self.generation += 1

    # -- the substep (called by MPMModel.forward / backward)
    def forward(self, statics, state_curr, state_next, tape) -> None:
        lib, model = L.lib(), self.model
        h, s = model.handle(), model._stream()
        n = state_curr.particle.x.shape[0]
        st = statics.c_struct()
        cur, nxt = state_curr.particle.c_struct(), state_next.particle.c_struct()
        L.check(lib.nm_mpm_p2g(h, n, C.byref(st), C.byref(cur), s), "nm_mpm_p2g")
        self._ensure_sized()
        record = None
        if isinstance(tape, ShardTape):
            tape.cap = self.cap
            tape.buf = torch.empty(int(lib.nm_mpm_gridcache_bytes(self.cap)), dtype=torch.uint8, device=self.device)
            tape.shared = self.new_shared()
            tape.generation = self.generation
            shared, record = tape.shared, tape.buf
        elif tape is None:
            shared = self._scratch_shared
        else:
            raise L.NeumaHipError("a sharded MPMModel takes the tape from model.new_tape() (or None)")
        self.find_shared(shared)
        self.sum_blocks(0, shared)
        L.check(lib.nm_mpm_forward_finish(h, n, C.byref(st), C.byref(cur), C.byref(nxt), L.ptr(record), self.cap,
                                          L.ptr(self.status), s), "nm_mpm_forward_finish")

    def backward(self, statics, state_curr, state_next, gnext, gcur, tape) -> None:
        if not isinstance(tape, ShardTape) or tape.buf is None:
            raise L.NeumaHipError("the sharded reverse sweep needs the ShardTape its forward substep filled "
                                  "(model.new_tape()); there is no recompute fallback across ranks")
        if tape.generation == self.generation:
            self.check()                # first substep of this reverse sweep: were all forward substeps complete?
        lib, model = L.lib(), self.model
        h, s = model.handle(), model._stream()
        n = state_curr.particle.x.shape[0]
        st = statics.c_struct()
        cur, nxt = state_curr.particle.c_struct(), state_next.particle.c_struct()
        L.check(lib.nm_mpm_backward_begin(h, n, C.byref(st), C.byref(cur), C.byref(nxt), C.byref(gnext), C.byref(gcur),
                                          L.ptr(tape.buf), tape.cap, s), "nm_mpm_backward_begin")
        self.sum_blocks(1, tape.shared)
        L.check(lib.nm_mpm_backward_finish(h, n, C.byref(st), C.byref(cur), C.byref(gcur), s), "nm_mpm_backward_finish")


# ---------------------------------------------------------------- the library's RCCL communicator, one per process group
# id(pg) -> (pg, handle | False).  The entry keeps a strong reference to the ProcessGroup object, so its id cannot be handed to
# another object while the entry exists; an entry whose group has left torch.distributed's table (destroy_process_group, then a
# new group) is purged - and its communicator destroyed - at the next look-up instead of being returned for the new group.
_LIB_COMMS = {}


def _resolve_group(group):
    import torch.distributed as dist
    if group is not None:
        return group
    return dist.distributed_c10d._get_default_group()


def _group_alive(pg) -> bool:
    import torch.distributed as dist
    try:
        return pg in dist.distributed_c10d._world.pg_map
    except Exception:       # noqa: BLE001 - private table moved: assume alive (the strong reference still rules out an id clash)
        return True


def _purge_dead_comms() -> None:
    for key, (pg, h) in list(_LIB_COMMS.items()):
        if not _group_alive(pg):
            if h:
                try:
                    L.lib().nm_rccl_destroy(h)
                except Exception:       # noqa: BLE001
                    pass
            del _LIB_COMMS[key]


def create_library_comm(group, device):
    """COLLECTIVE: create (once) the library-owned RCCL communicator of a torch.distributed group (csrc/nm_rccl.hip) - rank 0's
    ncclUniqueId goes round through ONE torch.distributed broadcast, every rank runs ncclCommInitRank, one MIN all-reduce makes
    the ranks agree on whether it exists.  Every rank of the group must call this at the same point: it is called where a
    multi-rank object is set up (MPMModel.shard / GridExchange, SceneRuntime.__init__, bench.py's calibration), never from inside
    a collective wrapper or an autograd backward.  Returns the handle, or None when the group's backend is not nccl (the gloo
    tests), NEUMA_COMM=python asks for torch.distributed, or some rank could not create it (then no rank uses it)."""
    import os
    import torch.distributed as dist
    _purge_dead_comms()
    pg = _resolve_group(group)
    key = id(pg)
    ent = _LIB_COMMS.get(key)
    if ent is not None:
        return ent[1] or None
    h = False
    if os.environ.get("NEUMA_COMM", "rccl") != "python" and str(dist.get_backend(group)) == "nccl":
        lib = L.lib()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        idt = torch.zeros(128, dtype=torch.uint8)
        rc0 = 0
        if rank == 0:
            # a failure here (librccl cannot be bound, ncclGetUniqueId failed) must not raise: the other ranks are on their
            # way into the broadcast below.  Rank 0 sends zeros, skips its own create and the MIN all-reduce makes every
            # rank fall back to torch.distributed together
            rc0 = int(lib.nm_rccl_unique_id(C.c_void_p(idt.data_ptr())))
            if rc0:
                idt.zero_()
        dev_id = idt.to(device)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(dev_id, src=src, group=group)
        idt = dev_id.cpu()
        hh = C.c_void_p()
        rc = rc0
        if rc == 0:
            with torch.cuda.device(device):
                rc = lib.nm_rccl_create(C.c_void_p(idt.data_ptr()), world, rank, C.byref(hh))
        # every rank uses the library's communicator or none does: a rank on which it could not be created must not leave
        # the others waiting inside a collective it never issues
        ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 1:
            h = hh
        else:
            import warnings
            why = (lib.nm_last_error() or b"").decode() if rc else "another rank could not create it"
            warnings.warn(f"library-owned RCCL communicator not available ({why}): collectives go through torch.distributed")
            if rc == 0:
                lib.nm_rccl_destroy(hh)
    _LIB_COMMS[key] = (pg, h)
    return h or None


def library_comm_for(group, device=None):
    """LOOK-UP ONLY (never collective, never creates): the communicator create_library_comm made for this group, or None - the
    per-frame collectives below then go through torch.distributed, on every rank alike (creation is all-or-none)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    try:
        pg = _resolve_group(group)
    except Exception:       # noqa: BLE001 - no default group
        return None
    ent = _LIB_COMMS.get(id(pg))
    if ent is None:
        return None
    if ent[0] is not pg or not _group_alive(pg):
        _purge_dead_comms()
        return None
    return ent[1] or None


def close_library_comms() -> None:
    """Release every library-owned communicator of this process (call before dist.destroy_process_group(); entries of groups
    destroyed without it are released at the next create / look-up)."""
    for key, (_pg, h) in list(_LIB_COMMS.items()):
        if h:
            try:
                L.lib().nm_rccl_destroy(h)
            except Exception:       # noqa: BLE001 - interpreter shutdown
                pass
        del _LIB_COMMS[key]


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum of a contiguous fp32 CUDA tensor over the ranks: ncclAllReduce on the library's communicator, issued on the
    current stream (no torch.distributed call, no stream hop); torch.distributed when there is no such communicator (gloo,
    NEUMA_COMM=python, CPU tensors, other dtypes)."""
    import torch.distributed as dist
    h = library_comm_for(group, t.device) if (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()) else None
    if h is not None:
        L.check(L.lib().nm_rccl_all_reduce_sum_f32(h, L.ptr(t), t.numel(), L.stream_ptr(t.device)), "nm_rccl_all_reduce_sum_f32")
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_rows_(out: torch.Tensor, mine: torch.Tensor, group=None) -> torch.Tensor:
    """out (world x chunk rows) <- every rank's `mine` (chunk rows), 4-byte elements: ncclAllGather on the library's communicator
    (the bit patterns travel as int32) or dist.all_gather_into_tensor."""
    import torch.distributed as dist
    ok = mine.is_cuda and mine.element_size() == 4 and mine.is_contiguous() and out.is_contiguous()
    h = library_comm_for(group, mine.device) if ok else None
    if h is not None:
        L.check(L.lib().nm_rccl_all_gather_i32(h, L.ptr(mine), L.ptr(out), mine.numel(), L.stream_ptr(mine.device)), "nm_rccl_all_gather_i32")
    else:
        dist.all_gather_into_tensor(out, mine, group=group)
    return out


# ---------------------------------------------------------------- particle rows <-> all rows
class _GatherRows(torch.autograd.Function):
    """Forward: all-gather the ranks' row blocks into the full (N, ...) tensor.  Backward: this rank's rows of the
    incoming gradient (`summed` = the gradient is already the same on every rank, e.g. behind merge_grad_across_ranks)
    or of its all-reduced sum."""

    @staticmethod
    def forward(ctx, local, num_rows: int, group, summed: bool):
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        lo, hi = shard_range(num_rows, world, rank)
        if local.shape[0] != hi - lo:
            raise L.NeumaHipError(f"rank {rank} owns rows [{lo},{hi}) but holds {local.shape[0]}")
        chunk = -(-num_rows // world)
        tail = local.shape[1:]
        padded = local.new_zeros((chunk,) + tail)
        padded[:hi - lo] = local
        out = local.new_empty((world * chunk,) + tail)
        all_gather_rows_(out, padded, group)
        ctx.meta = (group, lo, hi, summed)
        if chunk * world == num_rows:
            return out
        pieces = [out[r * chunk:r * chunk + (shard_range(num_rows, world, r)[1] - shard_range(num_rows, world, r)[0])]
                  for r in range(world)]
        return torch.cat(pieces, 0)

    @staticmethod
    def backward(ctx, grad):
        import torch.distributed as dist
        group, lo, hi, summed = ctx.meta
        if not summed:
            grad = grad.contiguous().clone()
            all_reduce_sum_(grad, group)
        return grad[lo:hi].contiguous(), None, None, None


def gather_rows(local: torch.Tensor, num_rows: int, group=None, grad_is_summed: bool = False) -> torch.Tensor:
    return _GatherRows.apply(local.contiguous(), num_rows, group, grad_is_summed)


def reduce_param_grads(params: Iterable[torch.nn.Parameter], group=None) -> None:
    """Sum the parameter gradients over the ranks (each rank holds the contribution of its particles) - the one
    collective a data-parallel trainer adds after loss.backward().  One flat all-reduce."""
    import torch.distributed as dist
    if dist.get_world_size(group) == 1:
        return
    grads = []
    for p in params:
        if not p.requires_grad:
            continue
        if p.grad is None:                       # e.g. a rank without particles: it still takes part with zeros
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    all_reduce_sum_(flat, group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
