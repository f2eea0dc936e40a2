"""Host side of the particle-sharded simulation on CPU: row ownership, and the two-rank gloo collectives around it
(gather_rows forward / backward, parameter-gradient reduction)."""
import socket

import torch.multiprocessing as mp

import shard_worker


def test_shard_ranges_partition_the_particle_list():
    from neuma_amd.sim.shard import shard_range, size_with_slack, explain_status
    for n, world in [(100000, 8), (11, 2), (7, 8), (0, 3), (1000003, 8)]:
        edges = [shard_range(n, world, r) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1
    assert size_with_slack(1000) == 1564
    assert "cap_shared" in explain_status(2) and "cache" in explain_status(4) and explain_status(0) == ""


def test_two_rank_gloo_gather_rows_and_param_grads():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [ctx.Process(target=shard_worker.cpu_rows, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in ps]
    [p.join(30) for p in ps]
    for r in res:
        assert "error" not in r, r
        assert all(r["ok"]), r


def _run(target, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ps = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=180) for _ in ps]
    [p.join(30) for p in ps]
    return res


def test_shared_block_rule_and_exchange_bookkeeping_over_gloo():
    """world_size 2 and 3: the ordering rule of nm_mpm_shared_blocks and the pack / all-reduce / unpack bookkeeping, on the host."""
    for world in (2, 3):
        res = _run(shard_worker.cpu_exchange, world)
        for r in res:
            assert "error" not in r, r
            assert all(r["ok"]), r
            assert r["n_shared"] >= 4


def test_comm_callbacks_of_the_library_level_sharded_rollout_over_gloo():
    """rollout._ShardLink: the two collectives nm_rollout_forward_sharded calls back for, driven through the C function pointers."""
    for world in (2, 3):
        res = _run(shard_worker.cpu_comm_link, world)
        for r in res:
            assert "error" not in r, r
            assert all(r["ok"]), r


def test_stripe_plan_covers_every_tile_row_once():
    from neuma_amd.harness import stripe_plan
    for V, rows, world in [(3, 68, 2), (3, 68, 8), (1, 16, 4), (3, 68, 3), (2, 5, 7)]:
        seen = {}
        for r in range(world):
            for (v, a, b) in stripe_plan(V, rows, world, r):
                assert 0 <= a < b <= rows
                for y in range(a, b):
                    assert (v, y) not in seen
                    seen[(v, y)] = r
        assert len(seen) == V * rows
