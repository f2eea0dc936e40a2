"""The multi-rank path of bench.py, end to end, on the one GPU of the test box: `python bench.py --gpus N` starts its own N ranks
(torch.distributed.run on 127.0.0.1), here with the gloo backend (RCCL refuses two ranks on one device) - the code the
driver's SCALE run executes (rendezvous, shard decision, sharded / replicated simulation, render stripes, the frame's
collectives, the per-rank report, the one JSON line), everything but the transport.  The reference has no counterpart
(experiments/configs/synthetic/finetune-bb.yaml:1 `gpu: 0`)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

from gpu_util import measured

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(gpus, workload="tiny", shard="auto", exchange=None, steps=3, warmup=1, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("NEUMA_SHARD_EXCHANGE", None)
    if gpus > 1:
        env["NEUMA_DIST_BACKEND"] = "gloo"
    if exchange:
        env["NEUMA_SHARD_EXCHANGE"] = exchange
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", str(warmup), "--workload", workload,
           "--no-cpu-baseline", "--epoch-frames", "0", "--shard-sim", shard]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"rc {p.returncode}\n{p.stderr[-3000:]}"
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {p.stdout[:500]}"
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def one_rank():
    return {w: _bench(1, workload=w) for w in ("tiny",)}


def _check_line(d, world, shard, steps=3):
    assert d["n_gpus"] == world and d["world"] == world and d["steps"] == steps
    assert d["metric"].startswith("sim+render frames/sec") and d["unit"] == "frames/s" and d["value"] > 0
    assert d["backend"] == "gloo" and d["scaling"] == "strong"
    assert isinstance(d["per_rank"], list) and len(d["per_rank"]) == world
    assert sorted(r["rank"] for r in d["per_rank"]) == list(range(world))
    for r in d["per_rank"]:
        assert r["ms_per_frame"] > 0 and r["shard_sim"] == shard and "all_reduce_K_x_3_us" in r["collectives"]
        if shard:
            assert "gather_rows_x_us" in r["collectives"]
    n_all = sum(r["particles"] for r in d["per_rank"])
    assert (n_all == d["per_rank"][0]["particles"] * world) or shard       # replicated: every rank holds all particles
    assert ("particle-sharded" in d["config"]["parallelism"]) == shard
    if shard:
        assert d["shard_collectives"]          # which transport the sharded substep's exchanges used
    assert d["roofline"] is not None and d["shard_cost_model"] is not None


@pytest.mark.parametrize("shard,exchange", [("on", "allreduce"), ("on", "peers"), ("off", None)])
def test_bench_two_ranks_gloo(one_rank, shard, exchange):
    d = _bench(2, shard=shard, exchange=exchange)
    _check_line(d, 2, shard == "on")
    if exchange == "peers":
        assert "swapped with ranks" in d["shard_collectives"]
    # the ranks' stripe losses add up to the one-rank loss of the same frame (fp32 partial sums in a different order, and the
    # sharded simulation sums the shared grid blocks in rank order): measured <= 3.8e-7 relative over the five cases
    ref = one_rank["tiny"]["loss"]
    assert measured(abs(d["loss"] - ref) / abs(ref), "loss, 2 ranks vs 1 rank (rel)") < 1e-5


@pytest.mark.parametrize("shard", ["on", "off"])
def test_bench_eight_ranks_gloo(one_rank, shard):
    """Eight ranks on the smallest workload (2000 particles: 250 per rank, 12 x 2 render stripes over 8 ranks)."""
    d = _bench(8, shard=shard, timeout=900)
    _check_line(d, 8, shard == "on")
    ref = one_rank["tiny"]["loss"]
    assert measured(abs(d["loss"] - ref) / abs(ref), "loss, 8 ranks vs 1 rank (rel)") < 1e-5


def test_bench_two_ranks_auto_decision_bb():
    """--shard-sim auto at the reference's own size (bb: 8k particles): the start-up calibration runs (roll-outs at N and N / 2,
    the all-reduce and the peer exchange timed) and every rank takes the same decision."""
    d = _bench(2, workload="bb", shard="auto")
    assert d["n_gpus"] == 2 and len(d["per_rank"]) == 2
    m = d["shard_cost_model"]
    assert m is not None and "shard" in m
    assert len({r["shard_sim"] for r in d["per_rank"]}) == 1
    assert ("particle-sharded" in d["config"]["parallelism"]) == bool(d["per_rank"][0]["shard_sim"])
