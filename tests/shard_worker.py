"""Rank body of the particle-sharded tests (spawned by test_gpu_shard.py / test_shard_cpu.py; not a test module).

GPU cases run `world` processes on cuda:0 with the gloo backend: the box has one GPU, and what is checked is the
exchange logic (which blocks are summed, in which order, forward and backward), which does not depend on the transport.
Every rank also runs the unsharded model on all particles and compares its own rows against it."""
import os

import torch


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def _err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _deformed_start(run):
    """A start state away from F = I: at rest the invariants sigma - 1 and F^T F - I are differences of nearly equal numbers,
    and every gradient of a frame inherits that - two runs of the SAME path differ by 3e-3 there (order of fp32 atomics),
    against 4e-7 from this state (tools/exp_grad_noise.py)."""
    run.set_start_state("deformed")       # F = I + 0.05 N(0, 1), the same on every rank (seeded)


def gpu_substeps(rank, world, port, q, steps=3, cap_shared=None, cap=None):
    """`steps` chained substeps with given stresses: sharded rows vs the unsharded model, states and gradients."""
    try:
        dist = _init(rank, world, port)
        from gpu_util import mpm_case, build_model, build_statics
        from neuma_amd.sim import MPMDiffSim
        from neuma_amd.sim.shard import shard_range, gather_rows
        dev = torch.device("cuda", 0)
        const, vol, rho, clip, en, x, v, C, F, S = mpm_case(N=6000, G=32, seed=3)
        N = x.shape[0]
        lo, hi = shard_range(N, world, rank)
        g = torch.Generator().manual_seed(11)
        wts = [torch.randn(t.shape, generator=g).to(dev) for t in (x, v, C, F)]

        def run(model, statics, rows, gather):
            leaves = [t[rows].float().to(dev).requires_grad_(True) for t in (x, v, C, F)]
            stresses = [(S[rows] * (1.0 + 0.25 * k)).float().to(dev).requires_grad_(True) for k in range(steps)]
            sim = MPMDiffSim(model, reorder=False)
            state = leaves
            for k in range(steps):
                state = sim(statics, *state, stresses[k])
            full = [gather(t) for t in state]
            loss = sum((w * t).sum() for w, t in zip(wts, full))
            loss.backward()
            return [t.detach() for t in full], [t.grad for t in leaves + stresses]

        ref_model = build_model(const, dev)
        ref_out, ref_grad = run(ref_model, build_statics(ref_model, vol, rho, clip, en, dev), slice(0, N), lambda t: t)

        model = build_model(const, dev)
        ex = model.shard(None, cap=cap, cap_shared=cap_shared)
        rows = slice(lo, hi)
        st = build_statics(model, vol[rows], rho[rows], clip[rows], en[rows], dev)
        out, grad = run(model, st, rows, lambda t: gather_rows(t, N, None, grad_is_summed=True))
        ex.check()
        res = {"rank": rank, "cap": ex.cap, "cap_shared": ex.cap_shared}
        res["out_err"] = [_err(a, b) for a, b in zip(out, ref_out)]
        res["out_abs"] = [float((a - b).abs().max()) for a, b in zip(out, ref_out)]
        res["grad_err"] = [_err(a, b[rows]) for a, b in zip(grad, ref_grad)]
        # how many blocks were actually exchanged in a substep (forward-only call uses the scratch list)
        with torch.no_grad():
            sim = MPMDiffSim(model, reorder=False)
            sim(st, *[t[rows].float().to(dev) for t in (x, v, C, F)], S[rows].float().to(dev))
        res["shared_blocks"] = int(ex._scratch_shared[0])
        res["status"] = int(ex.status.item())
        q.put(res)
        dist.destroy_process_group()
    except Exception as e:      # report instead of leaving the parent waiting on the queue
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def gpu_frame(rank, world, port, q, name="tiny", fused=False, preset_caps=False, exchange="allreduce"):
    """One frame of the driver (materials + roll-out + bindings + render + loss, forward and backward): sharded
    simulation on `world` ranks vs the single-process frame.  fused: the sharded ranks run nm_rollout_forward_sharded /
    nm_rollout_backward_sharded (substep loop, phases and collectives inside the library) instead of the per-operator
    classes driven from Python."""
    try:
        os.environ["NEUMA_SHARD_EXCHANGE"] = exchange      # "peers": shared blocks swapped with the neighbour ranks only
        dist = _init(rank, world, port)
        from neuma_amd import synth
        from neuma_amd.harness import SceneRuntime
        dev = torch.device("cuda", 0)
        scene = synth.make_scene(name)
        ref = SceneRuntime(scene, dev, fused=False)
        ref.make_ground_truth()
        rt = SceneRuntime(scene, dev, rank=rank, world=world, shard_sim=True, fused=fused)
        assert rt.fused == fused and rt.model.exchange is not None and rt.model.exchange.exchange_mode == exchange
        if preset_caps:      # the caller fixes cap / cap_shared and the FIRST operation is a fused roll-out: the frame-level
            rt.model.shard(rt.group, cap=4096, cap_shared=4096)      # capacities must still be probed from the start state
        rt.gt = ref.gt
        for run in (ref, rt):
            # a material that is visibly wrong for the ground truth, so that the LoRA gradients are a signal and not the
            # rounding residue of a sum that cancels; and the initial velocities as leaves (per-particle gradients)
            with torch.no_grad():
                for p in run.parameters():
                    if p.shape[0] in (64, 9):       # the lora_B factors
                        p.mul_(-4.0)
            run.v0.requires_grad_(True)
            _deformed_start(run)
        r0 = ref.frame()
        r1 = rt.frame()
        tot = r1.loss.clone()
        dist.all_reduce(tot)
        res = {"rank": rank, "loss": float(tot), "ref_loss": float(r0.loss),
               "x_err": _err(r1.x, r0.x), "F_err": _err(r1.F, r0.F),
               "v0_err": _err(rt.v0.grad[rt.rows], ref.v0.grad[rt.rows]), "v0_mag": float(ref.v0.grad.abs().max())}
        # LoRA gradients: with the shipped weights this small ball sits near equilibrium and dL/dW is the residue of a sum
        # that cancels to 1e-4 of its terms (two unsharded paths differ by 0.4 % there), so they are compared for a
        # material that is visibly off (base weights x1.5, deformed F0) under random weights on the rolled-out x and F
        from neuma_amd.sim.shard import gather_rows, reduce_param_grads
        g = torch.Generator().manual_seed(5)
        wx, wF = torch.randn(rt.N, 3, generator=g).to(dev), torch.randn(rt.N, 3, 3, generator=g).to(dev)
        grads = []
        gf = torch.Generator().manual_seed(7)
        dF = 0.05 * torch.randn(rt.N, 3, 3, generator=gf).to(dev)
        for run in (ref, rt):
            with torch.no_grad():
                for net in (run.elasticity, run.plasticity):
                    for name, p in net.named_parameters():
                        if "lora" not in name:
                            p.mul_(1.5)
            run.F0 = run.F0 + dF
            for p in run.parameters():
                p.grad = None
            rw = run.rows
            x, v, C, F = run.rollout(run.x0[rw], run.v0[rw].detach(), run.C0[rw], run.F0[rw])
            if run is rt:
                x, F = gather_rows(x, rt.N, None, True), gather_rows(F, rt.N, None, True)
            ((wx * x).sum() + (wF * F).sum()).backward()
            if run is rt:
                reduce_param_grads(run.parameters(), None)
                rt.model.exchange.check()
            grads.append([p.grad.clone() for p in run.parameters()])
        res["grad_err"] = [_err(a, b) for a, b in zip(grads[1], grads[0])]
        res["grad_mag"] = [float(b.abs().max()) for b in grads[0]]
        res["peers"] = rt.model.exchange.peers
        res["backend"] = getattr(rt.model.exchange, "link_backend", None)
        q.put(res)
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def cpu_rows(rank, world, port, q):
    """Host-side pieces on CPU tensors: row ownership, gather_rows forward / backward, parameter-gradient reduction."""
    try:
        dist = _init(rank, world, port)
        from neuma_amd.sim.shard import shard_range, gather_rows, reduce_param_grads
        N = 11                                      # not divisible by the world size: exercises the padding
        torch.manual_seed(0)
        full = torch.randn(N, 3)
        lo, hi = shard_range(N, world, rank)
        local = full[lo:hi].clone().requires_grad_(True)
        w = torch.randn(N, 3)
        out = gather_rows(local, N, None, grad_is_summed=True)
        ok_fwd = bool(torch.equal(out, full))
        (out * w).sum().backward()
        ok_bwd = bool(torch.allclose(local.grad, w[lo:hi]))
        local2 = full[lo:hi].clone().requires_grad_(True)
        out2 = gather_rows(local2, N, None, grad_is_summed=False)
        (out2 * w * (rank + 1)).sum().backward()   # rank-dependent partial losses: gradients must be summed
        scale = sum(r + 1 for r in range(world))
        ok_sum = bool(torch.allclose(local2.grad, w[lo:hi] * scale))
        p = torch.nn.Parameter(torch.ones(4))
        p2 = torch.nn.Parameter(torch.ones(2, 2))
        p.grad = torch.full((4,), float(rank + 1))
        if rank == 0:
            p2.grad = torch.ones(2, 2)              # the other rank has no gradient for p2 yet
        reduce_param_grads([p, p2])
        ok_par = bool(torch.allclose(p.grad, torch.full((4,), float(scale))) and torch.allclose(p2.grad, torch.ones(2, 2)))
        # start-up calibration of the shard decision (bench.py): the all-reduce timer through torch.distributed (gloo here), and
        # the decision taken from the maxima over the ranks - every rank must decide alike
        from neuma_amd.sim.shard import time_all_reduce_us, shard_cost_model
        ar = time_all_reduce_us(None, "cpu", count=1 << 10, reps=3)
        vals = torch.tensor([223.0 + rank, 147.5 - rank, ar], dtype=torch.float64)
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        cost = shard_cost_model(100_000, world, 20, dict(substep_us_full=float(vals[0]), substep_us_shard=float(vals[1]), allreduce_us=float(vals[2])))
        ok_cal = bool(ar > 0.0 and cost["inputs"] == "measured at start-up" and float(vals[0]) == 223.0 + world - 1)
        # the communicator cache: look-ups never create (and are not collective); an entry dies with its process group - a group
        # created after destroy_process_group() must not inherit it, whatever id() the new object gets
        from neuma_amd.sim import shard as SH
        ok_lookup = SH.library_comm_for(None, "cpu") is None and len(SH._LIB_COMMS) == 0      # all of the above created nothing
        SH.create_library_comm(None, "cpu")          # gloo: recorded as "not available", no collective
        pg_old = dist.distributed_c10d._get_default_group()
        ok_entry = len(SH._LIB_COMMS) == 1 and SH.library_comm_for(None, "cpu") is None
        SH._LIB_COMMS[id(pg_old)] = (pg_old, False)
        dist.barrier()
        dist.destroy_process_group()
        ok_dead = not SH._group_alive(pg_old)
        SH._purge_dead_comms()
        ok_purged = len(SH._LIB_COMMS) == 0          # the entry of a destroyed group does not survive the next create / look-up
        q.put({"rank": rank, "ok": [ok_fwd, ok_bwd, ok_sum, ok_par, ok_cal, ok_lookup, ok_entry, ok_dead, ok_purged], "decision": bool(cost["shard"]), "sharded_us": cost["sharded_us"]})
    except Exception as e:
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def cpu_exchange(rank, world, port, q):
    """The shared-block rule and the pack / all-reduce / unpack bookkeeping of the sharded substep, on the host over gloo:
    every rank evaluates the rule on the all-gathered lists and must arrive at the same shared list, and after the exchange
    every block holds the dense (single-rank) sum on exactly the ranks that list it."""
    try:
        import numpy as np
        dist = _init(rank, world, port)
        from neuma_amd.sim.shard import shared_blocks_host, exchange_blocks_host
        nblocks, cap, cap_shared = 200, 24, 16
        rng = np.random.default_rng(7)
        # overlapping block sets: rank r lists a window of a shuffled block order, plus an out-of-range id and duplicates of nothing
        order = rng.permutation(nblocks)
        lists = [order[8 * r: 8 * r + 14].tolist() for r in range(world)]
        lists[0] = lists[0] + [nblocks + 5]                      # an invalid id must be ignored
        mine_list = lists[rank]
        row = np.full(1 + cap, -1, np.int32)
        row[0] = len(mine_list); row[1:1 + len(mine_list)] = mine_list
        t = torch.from_numpy(row.copy())
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        g = torch.stack(gathered).numpy()
        ids, mine, bits = shared_blocks_host(g, cap, nblocks, cap_shared, rank)
        # (1) the same list in the same order on every rank
        mine_ids = torch.full((cap_shared,), -1, dtype=torch.int64); mine_ids[:len(ids)] = torch.from_numpy(ids.astype(np.int64))
        all_ids = [torch.empty_like(mine_ids) for _ in range(world)]
        dist.all_gather(all_ids, mine_ids)
        ok_same = all(bool(torch.equal(a, all_ids[0])) for a in all_ids)
        # (2) it is what the definition says: blocks in >= 2 valid lists, ordered by first appearance in rank order
        valid = [[b for b in l if 0 <= b < nblocks] for l in lists]
        want = []
        for r in range(world):
            for b in valid[r]:
                if b not in want and sum(b in v for v in valid) >= 2:
                    want.append(b)
        ok_rule = ids.tolist() == want[:cap_shared] and bits == (2 if len(want) > cap_shared else 0)
        ok_mine = all(bool(m) == (int(b) in valid[rank]) for b, m in zip(ids, mine))
        # (3) pack / all-reduce / unpack: block values = (rank + 1) where the rank lists the block
        vals = np.zeros((nblocks, 64, 4), np.float32)
        for b in valid[rank]:
            vals[b] = rank + 1.0 + 0.001 * b
        before = vals.copy()

        def allreduce(buf):
            tb = torch.from_numpy(buf)
            dist.all_reduce(tb)

        exchange_blocks_host(vals, ids, mine, allreduce)
        ok_sum = True
        for b in range(nblocks):
            listed = [r for r in range(world) if b in valid[r]]
            if b in ids.tolist() and rank in listed:
                ok_sum &= bool(np.allclose(vals[b], sum(r + 1.0 + 0.001 * b for r in listed)))
            else:
                ok_sum &= bool(np.array_equal(vals[b], before[b]))       # untouched: unshared, or not listed by this rank
        # (4) capacity overflow is reported identically everywhere
        _, _, bits_small = shared_blocks_host(g, cap, nblocks, 2, rank)
        g_over = g.copy(); g_over[0, 0] = cap + 3
        _, _, bits_over = shared_blocks_host(g_over, cap, nblocks, cap_shared, rank)
        q.put({"rank": rank, "ok": [ok_same, ok_rule, ok_mine, ok_sum, bits_small & 2 == 2, bits_over & 1 == 1], "n_shared": len(ids)})
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def cpu_comm_link(rank, world, port, q):
    """The nm_comm callback table of the library-level sharded roll-out (rollout._ShardLink), invoked the way the library
    invokes it - through the C function pointers, with raw addresses inside the roll-out's workspace - on CPU tensors over
    gloo: rank-major all-gather of int32 lists and in-place float sum."""
    try:
        import ctypes as C
        import types
        dist = _init(rank, world, port)
        from neuma_amd import _lib as L
        from neuma_amd.rollout import _ShardLink
        ws = torch.zeros(4096, dtype=torch.uint8)
        ex = types.SimpleNamespace(group=None, world=world, rank=rank)
        link = _ShardLink(ex, ws)
        assert link.comm.world == world and link.comm.rank == rank
        base = ws.data_ptr()
        count = 5
        send_off, recv_off, buf_off = 256, 512, 2048
        ws[send_off:send_off + 4 * count].view(torch.int32).copy_(torch.arange(count, dtype=torch.int32) + 100 * rank)
        rc = link.comm.all_gather_i32(None, base + send_off, base + recv_off, count, None)
        got = ws[recv_off:recv_off + 4 * count * world].view(torch.int32).view(world, count)
        want = torch.stack([torch.arange(count, dtype=torch.int32) + 100 * r for r in range(world)])
        ok_gather = rc == 0 and bool(torch.equal(got, want))
        n = 7
        ws[buf_off:buf_off + 4 * n].view(torch.float32).copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        rc = link.comm.all_reduce_sum_f32(None, base + buf_off, n, None)
        tot = sum(r + 1 for r in range(world))
        ok_reduce = rc == 0 and bool(torch.allclose(ws[buf_off:buf_off + 4 * n].view(torch.float32), torch.arange(n, dtype=torch.float32) * tot))
        # a failing collective is reported by the return code (no exception may cross the C frames) and re-raised by check()
        rc = link.comm.all_reduce_sum_f32(None, base + ws.numel() + 64, n, None)       # address outside the workspace
        ok_err = rc != 0 and link.error is not None
        try:
            link.check(-1, "nm_rollout_forward_sharded")
            ok_err = False
        except L.NeumaHipError as e:
            ok_err = ok_err and "collective failed" in str(e)
        # neighbour-only exchange (nm_comm.exchange_peers_f32): a chain 0 - 1 - 2 ...: every rank swaps `n` floats with the ranks next
        # to it; the peers' buffers land behind one another, in rank order, at `recv`
        peers = sum(1 << q for q in (rank - 1, rank + 1) if 0 <= q < world)
        send2, recv2 = 3072, 3200
        ws[send2:send2 + 4 * n].view(torch.float32).copy_(torch.arange(n, dtype=torch.float32) + 10.0 * rank)
        ws[recv2:recv2 + 4 * n * 2].zero_()
        link.error = None
        rc = link.comm.exchange_peers_f32(None, base + send2, base + recv2, n, peers, None)
        got = ws[recv2:recv2 + 4 * n * 2].view(torch.float32).view(2, n)
        nbrs = [q for q in (rank - 1, rank + 1) if 0 <= q < world]
        ok_peers = rc == 0 and all(bool(torch.equal(got[k], torch.arange(n, dtype=torch.float32) + 10.0 * qq)) for k, qq in enumerate(nbrs))
        ok_peers = ok_peers and link.comm.peers == L.COMM_ALL_RANKS      # (no peer set on this exchange object: all-reduce mode)
        q.put({"rank": rank, "ok": [ok_gather, ok_reduce, ok_err, ok_peers]})
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def gpu_stripes_frame(rank, world, port, q, name="tiny"):
    """The replicated-simulation multi-GPU frame (what `--shard-sim auto` picks for the metric workload): every rank rolls out
    all particles, renders its stripes of the views' tile rows, the ranks sum dL/dmeans3D with ONE all-reduce per frame and
    finish the reverse sweep identically.  Against the single-process frame: the ranks' losses add up to its loss, and every
    rank ends with ITS full LoRA gradients (no parameter reduction in this mode)."""
    try:
        dist = _init(rank, world, port)
        from neuma_amd import synth
        from neuma_amd.harness import SceneRuntime
        dev = torch.device("cuda", 0)
        scene = synth.make_scene(name)
        ref = SceneRuntime(scene, dev, fused=True)
        ref.set_start_state("deformed")
        ref.make_ground_truth()
        rt = SceneRuntime(scene, dev, rank=rank, world=world, shard_sim=False, fused=True, group=dist.group.WORLD)
        rt.set_start_state("deformed")
        rt.gt = ref.gt
        out = {}
        for tag, run in (("ref", ref), ("rt", rt)):
            for _ in range(2):          # (second frame: stripe weights from the first, hinted plans, pooled buffers)
                for p in run.parameters():
                    p.grad = None
                r = run.frame()
            out[tag] = (r, [p.grad.clone() for p in run.parameters()])
        tot = out["rt"][0].loss.clone()
        dist.all_reduce(tot)
        res = {"rank": rank, "loss": float(tot), "ref_loss": float(out["ref"][0].loss),
               "x_err": _err(out["rt"][0].x, out["ref"][0].x),
               "grad_err": [_err(a, b) for a, b in zip(out["rt"][1], out["ref"][1])],
               "grad_mag": [float(b.abs().max()) for b in out["ref"][1]], "lean": bool(rt._lean_ok())}
        q.put(res)
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def gpu_nccl_one_rank(rank, world, port, q, name="tiny", comm="rccl"):
    """world = 1 on the RCCL backend ("nccl"): the collectives of both multi-GPU modes issued for real - the frame's K x 3
    gradient all-reduce (_AllReduceSum) and the fused sharded roll-out's callbacks on views of its device workspace
    (_ShardLink: one all-gather per roll-out, one all-reduce per substep and direction) - against the plain frame."""
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["NEUMA_SHARD_FORCE"] = "1"
        os.environ["NEUMA_COMM"] = comm          # "rccl": the library's own communicator; "python": the callback table
        os.environ["NEUMA_SHARD_EXCHANGE"] = "peers"     # (a one-rank world has no peers: the exchange is the empty one, no all-reduce)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from neuma_amd import synth
        from neuma_amd.harness import SceneRuntime, _AllReduceSum
        scene = synth.make_scene(name)
        ref = SceneRuntime(scene, dev, fused=True)
        ref.make_ground_truth()
        rt = SceneRuntime(scene, dev, rank=0, world=1, shard_sim=True, fused=True)
        assert rt.shard_sim and rt.model.exchange is not None and rt.fused
        rt.gt = ref.gt
        for run in (ref, rt):
            with torch.no_grad():
                for p in run.parameters():
                    if p.shape[0] in (64, 9):
                        p.mul_(-4.0)
            run.v0.requires_grad_(True)
            _deformed_start(run)
        r0, r1 = ref.frame(), rt.frame()
        rt.model.exchange.check()
        res = {"rank": rank, "backend": dist.get_backend(), "loss": float(r1.loss), "ref_loss": float(r0.loss),
               "x_err": _err(r1.x, r0.x), "F_err": _err(r1.F, r0.F), "v0_err": _err(rt.v0.grad, ref.v0.grad),
               "grad_err": [_err(a.grad, b.grad) for a, b in zip(rt.parameters(), ref.parameters())],
               "cap_frame": int(rt.model.exchange.cap_frame), "cap_dil": int(rt.model.exchange.cap_dil),
               "link": getattr(rt.model.exchange, "link_backend", None)}
        from neuma_amd import _lib as L
        from neuma_amd.sim.shard import time_all_reduce_us
        res["rccl_library"] = (L.lib().nm_rccl_library() or b"").decode()
        res["peers"] = rt.model.exchange.peers
        rc_comm = rt.model.exchange.library_comm()
        if rc_comm is not None:
            # the neighbour-only exchange's transport on a real communicator: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd
            # with the one pair a one-rank world has - this rank with itself
            import ctypes as C
            send, recv = torch.arange(10, dtype=torch.float32, device=dev) + 0.5, torch.zeros(10, dtype=torch.float32, device=dev)
            L.check(L.lib().nm_rccl_exchange_peers_f32(rc_comm, L.ptr(send), L.ptr(recv), 10, 1, L.stream_ptr(dev)), "nm_rccl_exchange_peers_f32")
            us = C.c_float(-1.0)
            L.check(L.lib().nm_rccl_time_exchange_peers(rc_comm, L.ptr(send), L.ptr(recv), 10, 1, 2, 5, C.byref(us), L.stream_ptr(dev)),
                    "nm_rccl_time_exchange_peers")
            torch.cuda.synchronize()
            res["self_swap_ok"] = bool(torch.equal(send, recv))
            res["exchange_us"] = float(us.value)
        res["allreduce_us"] = time_all_reduce_us(None, dev, count=1 << 16, rccl=rt.model.exchange.library_comm())
        # the stripe mode's collective: identity forward, all-reduce(sum) of the gradient backward
        g = torch.randn(1000, 3, device=dev)
        y = torch.randn(1000, 3, device=dev, requires_grad=True)
        (_AllReduceSum.apply(y, None) * g).sum().backward()
        res["allreduce_ok"] = bool(torch.equal(y.grad, g))
        q.put(res)
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})


def gpu_neighbourhood_miss(rank, world, port, q, cells=9.0, only_rank=None):
    """A body that crosses more than a block (4 cells) within one roll-out leaves the neighbourhood its rank announced at the
    first substep: every rank must see status bit 8 (exchange.check() raises), never a silently wrong sum."""
    try:
        dist = _init(rank, world, port)
        from neuma_amd import synth
        from neuma_amd.harness import SceneRuntime
        dev = torch.device("cuda", 0)
        rt = SceneRuntime(synth.make_scene("tiny"), dev, rank=rank, world=world, shard_sim=True, fused=True)
        rw = rt.rows
        speed = float(cells) / (rt.S * float(rt.scene.cfg["dt"]) * float(rt.scene.cfg["G"]))     # grid cells crossed per roll-out
        v = torch.zeros_like(rt.v0[rw])
        if only_rank is None or rank == only_rank:      # only_rank: ONE rank's particles leave - the others must hear of it
            v[:, 0] = speed
        with torch.no_grad():
            rt.rollout(rt.x0[rw], v, rt.C0[rw], rt.F0[rw])
        moved = float(speed) * rt.S * float(rt.scene.cfg["dt"]) * float(rt.scene.cfg["G"])
        res = {"rank": rank, "cells": moved}
        try:
            rt.model.exchange.check()
            res["raised"] = False
        except Exception as e:
            res.update(raised=True, text=str(e))
        # ... and every rank recovers together: the next roll-out re-probes the frame-level capacities (collectives that only
        # work if ALL ranks forgot them) and completes
        with torch.no_grad():
            rt.rollout(rt.x0[rw], torch.zeros_like(v), rt.C0[rw], rt.F0[rw])
        try:
            rt.model.exchange.check()
            res["rerun_ok"] = True
        except Exception as e:
            res.update(rerun_ok=False, rerun_text=str(e))
        q.put(res)
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put({"rank": rank, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()})
