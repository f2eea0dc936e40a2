"""Particle-sharded simulation (neuma_amd/sim/shard.py, csrc/nm_shard.hip) against the unsharded model.

The GPU box has one device, so the ranks are processes sharing cuda:0 and the collectives go through gloo; what is under
test - which grid blocks are summed over the ranks, in the forward and in the reverse sweep - is transport-independent.
On an 8-GPU node the same code runs over RCCL (bench.py --shard-sim)."""
import socket

import pytest
import torch
import torch.multiprocessing as mp

import shard_worker

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(target, world, *args, timeout=600):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    [p.start() for p in ps]
    try:
        res = [q.get(timeout=timeout) for _ in ps]
    finally:
        [p.join(30) for p in ps]
        for p in ps:
            if p.is_alive():
                p.kill()
    for r in res:
        assert "error" not in r, f"rank {r['rank']}: {r['error']}\n{r.get('trace', '')}"
    return sorted(res, key=lambda r: r["rank"])


@pytest.mark.parametrize("world,cap", [(2, None), (3, None), (2, 20000)])
def test_sharded_substeps_match_the_unsharded_model(world, cap):
    """cap=None: capacities sized from the first substep, short lists -> the one-launch shared-list kernel;
    cap=20000: 2 x 20001 gathered entries -> the multi-kernel (rocPRIM select) path."""
    res = _run(shard_worker.gpu_substeps, world, 3, None, cap)
    for r in res:
        assert r["shared_blocks"] > 0, "the ranks' particle ranges must overlap in some grid blocks for this test to mean anything"
        assert r["status"] == 0
        # states: fp32 summation order differs (permuting particles moves x by 1e-7, C by 2e-5 over 50 steps)
        from gpu_util import parity
        for nm, val, bnd in zip(("x", "v", "C", "F"), r["out_abs"], (2e-7, 1.5e-6, 7e-5, 1.2e-6)):      # measured 6e-8 | 4.5e-7 | 2.2e-5 | 3.6e-7
            parity(f"3 sharded substeps, world={world}, cap={cap}, rank {r['rank']}, vs unsharded model (abs)", nm, val, bnd)
        from gpu_util import measured
        assert measured(max(r["out_err"]), "rel_max states, 3 sharded substeps") < 2e-6, r      # measured 5.9e-07
        assert measured(max(r["grad_err"]), "rel_max gradients, 3 sharded substeps") < 3e-6, r          # stated gradient tolerance, SURVEY.md §8d


def test_capacity_overflow_is_reported_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=shard_worker.gpu_substeps, args=(r, 2, port, q, 1, 1)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=600) for _ in ps]
    [p.join(30) for p in ps]
    for r in res:
        assert "error" in r and "cap_shared" in r["error"] and "NeumaHipError" in r["error"], r


@pytest.mark.parametrize("world,fused,preset,exchange", [(2, False, False, "allreduce"), (2, True, False, "allreduce"),
                                                         (3, True, False, "allreduce"), (2, True, True, "allreduce"),
                                                         (2, True, False, "peers"), (3, True, True, "peers")])
def test_sharded_frame_matches_the_single_process_frame(world, fused, preset, exchange):
    """fused: the library-level sharded roll-out (nm_rollout_forward_sharded: the loop over substeps, phases and collectives
    runs in C and calls back for the two collectives per substep); otherwise the phases are driven from Python.
    preset: model.shard(group, cap=..., cap_shared=...) given by the caller and a fused roll-out as the very first operation -
    the frame-level capacities used to be sized from an empty grid then."""
    res = _run(shard_worker.gpu_frame, world, "tiny", fused, preset, exchange)
    for r in res:
        from gpu_util import measured
        if exchange == "peers":
            # exchange="peers": nm_comm.exchange_peers_f32 - every rank swaps its exchange buffer with the ranks its neighbourhood
            # meets (here: all of them, the ball is small) and adds them in rank order - instead of the all-reduce
            assert "error" not in r, r
            assert r["peers"] == sum(1 << q for q in range(world) if q != r["rank"]) and "swapped with ranks" in r["backend"], r
        assert measured(abs(r["loss"] - r["ref_loss"]) / abs(r["ref_loss"]), "rel loss, sharded frame") <= 1e-6, r      # measured 2.4e-07
        assert measured(r["x_err"], "rel_max x") < 7e-7 and measured(r["F_err"], "rel_max F") < 7e-7, r      # measured 2.0e-7
        # (deformed start state: at F = I the gradients of this small scene carry 3e-3 of atomics-order noise even between two
        #  runs of the same unsharded path - tools/exp_grad_noise.py)
        assert measured(r["v0_err"], "rel_max dL/dv0") < 7e-6, r      # measured 2.2e-06
        assert measured(max(r["grad_err"]), "rel_max LoRA gradients") < 1e-5 and min(r["grad_mag"]) > 1e-9, r      # measured 2.7e-06


@pytest.mark.parametrize("world", [2, 3])
def test_striped_frame_with_replicated_simulation_matches_the_single_process_frame(world):
    """Render stripes + one gradient all-reduce per frame, the simulation replicated on every rank (what --shard-sim auto picks
    for the metric workload): summed loss, state and every rank's LoRA gradients against the one-GPU frame."""
    res = _run(shard_worker.gpu_stripes_frame, world, "tiny")
    for r in res:
        from gpu_util import measured
        assert measured(abs(r["loss"] - r["ref_loss"]) / abs(r["ref_loss"]), "rel loss, striped frame") <= 3e-6, r      # measured 3.9e-07
        assert measured(r["x_err"], "rel_max x") < 3e-7, r      # measured 9.7e-08
        assert measured(max(r["grad_err"]), "rel_max LoRA gradients") < 5e-6 and min(r["grad_mag"]) > 0, r      # measured 6.5e-07
        assert r["lean"], r          # (the ranks ran the two-call frame: harness._frame_forward / _frame_backward)


@pytest.mark.parametrize("world,cap", [(2, 300), (3, 12000)])
def test_device_shared_block_rule_equals_the_host_statement(world, cap):
    """nm_mpm_shared_blocks (one-workgroup path for short lists, mark / flag / select / finish path for long ones) against
    neuma_amd.sim.shard.shared_blocks_host - the statement of the rule the CPU (gloo) tests run the exchange with."""
    import ctypes as C
    import numpy as np
    from neuma_amd import _lib as L
    from neuma_amd.sim import MPMModelBuilder
    from neuma_amd.sim.shard import shared_blocks_host
    d = torch.device("cuda", 0)
    model = MPMModelBuilder().parse_cfg(dict(gravity=[0.0, -9.8, 0.0], bc="noslip", num_grids=128, dt=1e-3, bound=1, eps=6e-7)).finalize(d)
    nblocks = ((128 + 2 + 3) // 4) ** 3
    rng = np.random.default_rng(world)
    g = np.full((world, 1 + cap), -1, np.int32)
    pool = rng.permutation(nblocks)[: int(1.4 * cap)]
    for r in range(world):
        n = cap - 7 * r
        g[r, 0] = n
        g[r, 1:1 + n] = rng.permutation(pool)[:n]
    g[0, 5] = nblocks + 11                                   # invalid id: ignored
    g[world - 1, 0] = cap + 4                                # a rank that overflowed its list: status bit 1
    for cap_shared in (cap, 40):
        ids, _, bits = shared_blocks_host(g, cap, nblocks, cap_shared)
        lib = L.lib()
        gathered = torch.from_numpy(g).to(d)
        shared = torch.zeros(2 + 2 * cap_shared, dtype=torch.int32, device=d)
        status = torch.zeros(1, dtype=torch.int32, device=d)
        ws = torch.empty(int(lib.nm_mpm_shared_workspace(world, cap)), dtype=torch.uint8, device=d)
        for _ in range(2):                                   # twice: the per-block counters must be left reset
            L.check(lib.nm_mpm_shared_blocks(model.handle(), L.ptr(gathered), world, cap, L.ptr(shared), cap_shared, L.ptr(status),
                                             L.ptr(ws), ws.numel(), L.stream_ptr(d)), "nm_mpm_shared_blocks")
            out = shared.cpu().numpy()
            assert int(out[0]) == len(ids) and int(out[1]) == bits and int(status.item()) == bits
            assert np.array_equal(out[2:2 + len(ids)], ids)


@pytest.mark.parametrize("comm", ["rccl", "python"])
def test_one_rank_on_the_rccl_backend_runs_both_multi_gpu_collective_paths(comm):
    """backend "nccl" (= RCCL), world = 1: the fused sharded roll-out and the stripe mode's gradient all-reduce, against the
    unsharded frame.  comm = "rccl" (default): the roll-out's collectives are ncclAllGather / ncclAllReduce issued by the
    library itself from its C loop, through the communicator it owns (csrc/nm_rccl.hip; librccl bound with dlopen);
    "python": the nm_comm callback table on device-workspace views (what the gloo tests use)."""
    (r,) = _run(shard_worker.gpu_nccl_one_rank, 1, "tiny", comm)
    assert r["backend"] == "nccl" and r["allreduce_ok"]
    assert r["peers"] == 0, r
    if comm == "rccl":
        assert r["link"].startswith("rccl") and "librccl" in r["rccl_library"], r
        assert 0.0 < r["allreduce_us"] < 1e4, r
        assert r["self_swap_ok"] and 0.0 < r["exchange_us"] < 1e4, r      # ncclSend / ncclRecv inside one group call
    else:
        assert r["link"].startswith("torch.distributed"), r
    from gpu_util import measured
    assert measured(abs(r["loss"] - r["ref_loss"]) / abs(r["ref_loss"]), "rel loss, one-rank RCCL frame") <= 1.5e-6, r      # measured 4.0e-07
    assert measured(r["x_err"], "rel_max x") < 7e-7 and measured(r["F_err"], "rel_max F") < 7e-7, r      # measured 2.0e-7
    assert measured(r["v0_err"], "rel_max dL/dv0") < 2e-6, r      # measured 5.1e-07
    assert measured(max(r["grad_err"]), "rel_max LoRA gradients") < 5e-6, r      # measured 1.5e-06
    assert r["cap_frame"] >= 64 and r["cap_dil"] > 64


def test_a_particle_leaving_its_announced_neighbourhood_is_reported_on_every_rank():
    res = _run(shard_worker.gpu_neighbourhood_miss, 2, 40.0)      # (20 cells = 5 blocks by the second substep)
    for r in res:
        assert r["cells"] > 30 and r["raised"] and "neighbourhood" in r["text"], r
    res = _run(shard_worker.gpu_neighbourhood_miss, 2, 1.5)        # a normal speed: nothing to report
    for r in res:
        assert r["cells"] < 2 and not r["raised"], r
    # ONE rank's body leaves (status bit 8 is raised by the rank it happens to): the roll-out OR-s the word over the ranks, so
    # both raise, both forget their frame-level capacities, and the re-run - whose capacity probe is a collective - completes
    res = _run(shard_worker.gpu_neighbourhood_miss, 2, 40.0, 1, timeout=240)
    for r in res:
        assert r["raised"] and "neighbourhood" in r["text"] and r["rerun_ok"], r


def test_device_peer_rule_equals_the_host_statement_and_is_symmetric():
    """nm_mpm_peer_ranks (which ranks announce a block of THIS rank's neighbourhood) against sim.shard.peer_ranks_host, for every
    rank of a four-rank world laid out along a bar: ranks two apart do not meet, so the peer sets are proper subsets of the world
    - and rank r is among q's peers exactly when q is among r's."""
    import ctypes as C
    import numpy as np
    from neuma_amd import _lib as L
    from neuma_amd.sim import MPMModelBuilder
    from neuma_amd.sim.shard import dilate_blocks_host, peer_ranks_host, peers_mask
    d = torch.device("cuda", 0)
    G, world = 64, 4
    nb = (G + 2 + 3) // 4
    model = MPMModelBuilder().parse_cfg(dict(gravity=[0.0, -9.8, 0.0], bc="noslip", num_grids=G, dt=1e-3, bound=1, eps=6e-7)).finalize(d)
    lib, h, s = L.lib(), model.handle(), L.stream_ptr(d)
    from gpu_util import build_statics
    # rank r's particles: a slab of the bar, three blocks (12 cells) long with a one-block gap to the next - neighbourhoods
    # (+-1 block) of consecutive ranks overlap, those of ranks two apart do not
    rng = np.random.default_rng(3)
    parts = []
    for r in range(world):
        x0 = (1 + 16 * r + 0.5) / G
        parts.append(np.stack([x0 + rng.random(400) * (11.0 / G), 0.4 + rng.random(400) * 0.1, 0.4 + rng.random(400) * 0.1], 1).astype(np.float32))
    lists = []
    for r in range(world):       # every rank's dilated list, from the device (host statement checked elsewhere)
        n = parts[r].shape[0]
        x = torch.from_numpy(parts[r]).to(d)
        v, Cm, F, S = torch.zeros(n, 3, device=d), torch.zeros(n, 3, 3, device=d), torch.eye(3, device=d).repeat(n, 1, 1), torch.zeros(n, 3, 3, device=d)
        st = build_statics(model, torch.full((n,), 1e-6), torch.full((n,), 1000.0), torch.full((n,), 0.1), torch.ones(n, dtype=torch.int32), d)
        keep = [t.float().contiguous() for t in (x, v, Cm, F, S)]
        cur = L.nm_particles(*[L.ptr(t) for t in keep])
        for _ in range(2):       # (twice: the first clear carries the previous rank's blocks - one handle plays every rank here)
            L.check(lib.nm_mpm_p2g(h, n, C.byref(st.c_struct()), C.byref(cur), s), "nm_mpm_p2g")
        out = torch.zeros(1 + nb ** 3, dtype=torch.int32, device=d)
        L.check(lib.nm_mpm_dilated_list(h, L.ptr(out), nb ** 3, s), "nm_mpm_dilated_list")
        got = out.cpu().numpy()
        lists.append(np.sort(got[1:1 + int(got[0])]))
    cap = max(len(l) for l in lists) + 5
    g = np.full((world, 1 + cap), -1, np.int32)
    for r in range(world):
        g[r, 0] = len(lists[r]); g[r, 1:1 + len(lists[r])] = lists[r]
    gathered = torch.from_numpy(g).to(d)
    masks = []
    for r in range(world):       # rank r's turn: ITS neighbourhood must be the one the handle holds (dilated_list tags it)
        n = parts[r].shape[0]
        x = torch.from_numpy(parts[r]).to(d)
        keep = [t.float().contiguous() for t in (x, torch.zeros(n, 3, device=d), torch.zeros(n, 3, 3, device=d),
                                                 torch.eye(3, device=d).repeat(n, 1, 1), torch.zeros(n, 3, 3, device=d))]
        st = build_statics(model, torch.full((n,), 1e-6), torch.full((n,), 1000.0), torch.full((n,), 0.1), torch.ones(n, dtype=torch.int32), d)
        cur = L.nm_particles(*[L.ptr(t) for t in keep])
        for _ in range(2):
            L.check(lib.nm_mpm_p2g(h, n, C.byref(st.c_struct()), C.byref(cur), s), "nm_mpm_p2g")
        out = torch.zeros(1 + nb ** 3, dtype=torch.int32, device=d)
        L.check(lib.nm_mpm_dilated_list(h, L.ptr(out), nb ** 3, s), "nm_mpm_dilated_list")
        adj = torch.full((world,), 7, dtype=torch.int32, device=d)
        L.check(lib.nm_mpm_peer_ranks(h, L.ptr(gathered), world, cap, r, L.ptr(adj), s), "nm_mpm_peer_ranks")
        want = peer_ranks_host(lists, r)
        assert adj.tolist() == want, (r, adj.tolist(), want)
        masks.append(peers_mask(want, r))
    assert masks == [0b0010, 0b0101, 0b1010, 0b0100], [bin(m) for m in masks]
    for r in range(world):
        for q in range(world):
            assert ((masks[r] >> q) & 1) == ((masks[q] >> r) & 1)


def test_device_neighbourhood_list_equals_the_host_statement():
    """nm_mpm_dilated_list against sim.shard.dilate_blocks_host, incl. blocks at the grid's faces (clipped neighbourhoods)."""
    import ctypes as C
    import numpy as np
    from gpu_util import mpm_case, build_model, build_statics
    from neuma_amd import _lib as L
    from neuma_amd.sim.shard import dilate_blocks_host
    d = torch.device("cuda", 0)
    const, vol, rho, clip, en, x, v, Cm, F, S = mpm_case(N=3000, G=32, seed=5, near_wall=True, disabled=False)
    model = build_model(const, d)
    st = build_statics(model, vol, rho, clip, en, d)
    lib, h, s = L.lib(), model.handle(), L.stream_ptr(d)
    cur = L.nm_particles(*[L.ptr(t.float().to(d).contiguous()) for t in (x, v, Cm, F, S)])
    keep = [t.float().to(d).contiguous() for t in (x, v, Cm, F, S)]
    cur = L.nm_particles(*[L.ptr(t) for t in keep])
    L.check(lib.nm_mpm_p2g(h, x.shape[0], C.byref(st.c_struct()), C.byref(cur), s), "nm_mpm_p2g")
    nb = (32 + 2 + 3) // 4
    act = torch.zeros(1 + nb ** 3, dtype=torch.int32, device=d)
    L.check(lib.nm_mpm_active_list(h, L.ptr(act), nb ** 3, s), "nm_mpm_active_list")
    ids = act[1:1 + int(act[0])].cpu().numpy()
    want = dilate_blocks_host(ids, nb)
    assert len(ids) > 20 and len(want) > len(ids) and want.min() == 0            # (the low wall: a clipped neighbourhood)
    for cap in (nb ** 3, 16):
        out = torch.zeros(1 + cap, dtype=torch.int32, device=d)
        for _ in range(2):                                                        # twice: a new tag per negotiation
            out.fill_(-1)
            L.check(lib.nm_mpm_dilated_list(h, L.ptr(out), cap, s), "nm_mpm_dilated_list")
            got = out.cpu().numpy()
            assert int(got[0]) == len(want)                                       # the count is never clamped
            if cap >= len(want):
                assert np.array_equal(np.sort(got[1:1 + len(want)]), want)
            else:
                assert set(got[1:1 + cap].tolist()) <= set(want.tolist())
