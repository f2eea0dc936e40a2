#!/usr/bin/env python3
"""Benchmark of the hot path: simulated + rendered frames/s, forward + backward.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or plainly
     as `python bench.py --gpus N`, which starts its own N ranks through torch.distributed.run on 127.0.0.1)

One "step" = one video frame of the BASELINE.json metric workload: 100k particles / 128^3 grid / 200k Gaussians /
1920x1080, S = 20 MPM substeps (each: elasticity net -> p2g/grid/g2p -> plasticity net) + V = 3 view renders +
pixel loss, then the full backward sweep (the settings NeuMA ships for its 1080p scenes:
experiments/configs/realworld/finetune-burger.yaml:109-113).  Synthetic data (neuma_amd/synth.py), real shipped
constitutive weights + LoRA r=16.  Inputs are resident in HBM before the timed region.

Multi-GPU: strong scaling of the same frame — the V x tile-row render stripes split across ranks, one RCCL all-reduce
of dL/dmeans3D per frame; the simulation replicated, or (--shard-sim, default from 100k particles per rank)
particle-sharded with two block all-reduces per substep (neuma_amd/harness.py, neuma_amd/sim/shard.py).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))


def prof_table(lib):
    import ctypes as C
    buf = C.create_string_buffer(1 << 16)
    lib.nm_prof_report(buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.rsplit(" ", 2)
        name = name.strip("()")          # (template instances with a comma are launched through a parenthesised macro argument)
        out[name] = (int(calls), float(ms))
    return out


# algorithmic flop per particle and launch of the constitutive kernels (see the roofline block in main)
MATERIAL_FLOPS = {"k_material_fwd": 11008.0, "k_material_fwd_pair": 2 * 11008.0, "k_material_bwd": 3 * 11008.0, "k_material_bwd_pair": 6 * 11008.0}


def algorithmic_bytes(kernel: str, rt, D) -> float:
    """Compulsory HBM bytes of ONE launch of `kernel` (formulas stated in DESIGN.md §Kernels).
    D: (Gaussian, tile) pairs of a view, or the dict of render.view_stats: the compositing kernels are then charged only the
    part of the lists they have to read - the entries up to where the tiles' pixels saturate (`walked`, from the forward
    pass's own walk record): 12 B per examined list entry (key + tile mask) and 36 B of Gaussian record per pair among them
    (pro rata: pairs x walked / list_entries), + the per-pixel state."""
    cfg = rt.scene.cfg
    N, K, W, H = rt.n_local, rt.K, cfg["W"], cfg["H"]   # simulation kernels see this rank's particles
    T = rt.touched_nodes
    base = kernel.split("<")[0]
    if isinstance(D, dict):
        st = D
        D = float(st["pairs"])
        walked = float(st["walked"]) if st.get("walked") else float(st["list_entries"])
        comp = 12.0 * walked + 36.0 * D * min(1.0, walked / max(float(st["list_entries"]), 1.0))
    else:
        comp = 40.0 * D
    table = {
        "k_render_bwd": comp + 20.0 * W * H + 36.0 * K,
        "k_render": comp + 20.0 * W * H,
        "k_render_fix": comp + 20.0 * W * H,      # second pass of the split compositing: at most the same stretch once more
        "k_preprocess": (4 * (3 + 6 + 1) + 12 * (cfg["sh"] + 1) ** 2) * K + 60.0 * K,
        "k_preprocess_bwd": (4 * (3 + 6) + 12 * (cfg["sh"] + 1) ** 2) * K + 36.0 * K + 12.0 * K,
        "k_material_fwd": 72.0 * N,
        "k_material_fwd_pair": (96.0 + 2 * 36.0) * N + 16.0 * T,     # g2p inputs + stencil nodes, F_{t+1} and stress_{t+1} out
        "k_material_bwd": 108.0 * N,
        "k_material_bwd_pair": 216.0 * N,
        "k_p2g": 124.0 * N + 16.0 * T,
        "k_g2p": 64.0 * N + 16.0 * T + 96.0 * N,
        "k_g2p_bwd": 192.0 * N + 32.0 * T,
        "k_p2g_bwd": 124.0 * N + 16.0 * T + 108.0 * N,
    }
    return table.get(base, 0.0)


def kernel_rooflines(full, rt, D, pmc=None, pmc_mfma=None):
    """Per-kernel roofline figures of the profiled frame (HIP-event durations of every launch; the views of the frame run
    on concurrent streams, so the render-side durations include some overlap): algorithmic bytes or flops per launch over
    the mean launch duration, against 8 TB/s HBM or the 157.3 TFLOP/s f32-MFMA peak.
    pmc / pmc_mfma (profiles/pmc_traffic.json, metric workload on one GPU only): the counter-measured HBM bytes per launch of
    the committed rocprofv3 passes next to the algorithmic figure (`traffic_over_algorithmic` well above 1 = wasted re-reads
    or scratch traffic), and for the constitutive kernels the fraction by the flops they execute - all marked static."""
    by = {}
    for name, (calls, ms) in full.items():
        b = name.split("<")[0]
        c0, m0 = by.get(b, (0, 0.0))
        by[b] = (c0 + calls, m0 + ms)
    out = []
    for b, (calls, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        avg_s = ms / calls / 1e3
        ab = algorithmic_bytes(b, rt, D)
        if b in MATERIAL_FLOPS:
            fl = MATERIAL_FLOPS[b] * rt.n_local
            row = {"kernel": b, "bound": "mfma", "achieved": round(fl / avg_s / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                   "frac": round(fl / avg_s / 1e12 / 157.3, 4), "avg_us": round(avg_s * 1e6, 1), "launches": calls}
            if pmc_mfma and b in pmc_mfma:
                row["frac_of_executed_flops"] = round(pmc_mfma[b]["mfma_flops"] / avg_s / 1e12 / 157.3, 4)
                row["mfma_busy_pct_of_simd_cycles_static"] = pmc_mfma[b]["mfma_busy_pct_of_simd_cycles"]
        else:
            if ab <= 0:
                continue
            row = {"kernel": b, "bound": "hbm", "achieved": round(ab / avg_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                   "frac": round(ab / avg_s / 1e9 / 8000.0, 4), "avg_us": round(avg_s * 1e6, 1), "launches": calls}
        if pmc and b in pmc and ab > 0:
            row["traffic_static"] = pmc[b]["hbm_bytes_per_launch"]
            row["algorithmic_bytes_per_launch"] = round(ab)
            row["traffic_over_algorithmic"] = round(pmc[b]["hbm_bytes_per_launch"] / ab, 2)
        out.append(row)
    return out


def frame_roofline(full, rt, D, frame_s):
    """The whole frame against both roofs (SURVEY §8d accounting): algorithmic HBM bytes and flops of every launch of one profiled
    frame, summed, over the measured frame time."""
    nbytes = flops = 0.0
    for name, (calls, _ms) in full.items():
        b = name.split("<")[0]
        nbytes += algorithmic_bytes(b, rt, D) * calls
        flops += MATERIAL_FLOPS.get(b, 0.0) * rt.n_local * calls
    return {"algorithmic_GB_per_frame": round(nbytes / 1e9, 3), "algorithmic_GFLOP_per_frame": round(flops / 1e9, 2),
            "achieved_TB_per_s": round(nbytes / frame_s / 1e12, 3), "frac_of_hbm_peak": round(nbytes / frame_s / 8e12, 4),
            "achieved_TFLOP_per_s": round(flops / frame_s / 1e12, 2), "frac_of_f32_mfma_peak": round(flops / frame_s / 157.3e12, 4),
            "note": "bytes = sum over the frame's launches of the per-kernel algorithmic figures of DESIGN.md section 4 (kernels without "
                    "one - the small binning / planning launches - count 0); flops = the constitutive nets' (SURVEY 8d: forward 11 008, "
                    "reverse 3 x that per particle and net); both over ms_per_step of the timed region"}


def measure_epoch(rt, frames: int, single_frame_fps: float, reps: int = 5, recompute: bool = False) -> dict:
    """Frames/s inside a native BPTT epoch of `frames` frames (SceneRuntime.epoch), peak device memory while it runs, and whether
    the roll-outs' activation caches stayed inside their budget (frames beyond it recompute)."""
    import torch
    from neuma_amd.train import simulate_video
    dev = rt.device
    # ground truth = the video of a PERTURBED start (initial velocities + 0.02 N(0,1), as SceneRuntime.make_ground_truth does for
    # the single frame): with the runtime's own roll-out as the target the loss is 1e-9 and every gradient of the reverse sweep
    # is ~0 - nothing in the kernels branches on that, but a measurement should not depend on it
    v_keep = rt.v0
    try:
        g = torch.Generator().manual_seed(2)
        rt.v0 = (rt.v0.detach() + (0.02 * torch.randn(rt.v0.shape, generator=g)).to(dev)).contiguous()
        with torch.no_grad():
            gt = simulate_video(rt, frames)
    finally:
        rt.v0 = v_keep
    weights = [1.0] * frames
    params = rt.parameters()

    from neuma_amd import rollout as _R
    keep_mode = _R._ACT_CACHE
    if recompute:
        _R._ACT_CACHE = '0'          # what NEUMA_ACT_CACHE=0 sets: no activation cache, the reverse sweep recomputes the MLPs

    def once():
        for p in params:
            p.grad = None
        return rt.epoch(gt, weights)

    try:
        return _measure_epoch_body(rt, frames, single_frame_fps, reps, once, recompute)
    finally:
        _R._ACT_CACHE = keep_mode


def _measure_epoch_body(rt, frames, single_frame_fps, reps, once, recompute):
    import torch
    dev = rt.device
    once(); once()
    torch.cuda.synchronize(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        loss = once()
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2] * reps        # median epoch (a fresh process now and then pays for device memory inside one)
    rt.flush()
    G = int(rt.scene.cfg["G"])
    nb = ((G + 2 + 3) // 4) ** 3
    lib_bytes = 3 * nb * 1024 + 4 * nb * 7 + 48 * rt.n_local      # the MPM handle: three node arrays, flags / lists, reverse-substep scratch
    fps = frames * reps / dt
    note = dict(getattr(rt, "last_epoch_note", {}) or {})
    return {"frames": frames, "substeps_per_frame": int(rt.S), "views_per_frame": int(rt.V), "epochs_timed": reps,
            "frames_per_s": round(fps, 2), "ms_per_frame": round(1e3 / fps, 4), "epoch_ms_each": [round(1e3 * t, 2) for t in times], "single_frame_frames_per_s": round(single_frame_fps, 2),
            "ratio_to_single_frame": round(fps / single_frame_fps, 3),
            "peak_hbm_GB": round((torch.cuda.max_memory_allocated(dev) + lib_bytes) / 2 ** 30, 2),
            "peak_hbm_note": "torch allocator peak (checkpoints 132 B / particle / substep, SVD + activation caches, grid cache records, "
                             "rasterizer state kept per frame and view, ground-truth frames) + the library's own allocations; the "
                             "reference needs an 80 GB A100 for this (README.md:202, SURVEY App. E)",
            "mode": "recompute (NEUMA_ACT_CACHE=0)" if recompute else "activation cache auto (budget = half of the free HBM unless NEUMA_ACT_CACHE_GB)",
            "activation_cache": note, "loss": float(loss)}


def measure_training(rt, frames: int, epoch_fps: float, epochs: int = 7, warm: int = 2) -> dict:
    """A TRAINING run, timed: train.finetune_constitutive (finetune.py:331-427) for `epochs` epochs of `frames` frames at the
    workload's S and V with everything the reference's loop does between two sweeps inside the timed region - per-net
    clip_grad_norm_ (error_if_nonfinite), the two RAdam steps, the LR schedules, the loss read-back for the log line, and the LoRA
    re-merge + weight re-permutation the changed factors force at the start of the next epoch.  The weights change between
    epochs, so the stateful shortcuts of a frame (hinted split plans from the camera's previous render, the grid cache capacity,
    prepared weights) go stale the way they do in training.  Settings: configs/realworld/finetune-burger.yaml:67-113 (lr 5e-3 /
    5e-4, clip 0.1, cos schedule, decay 0.5 -> 1 over 40 % of the epochs, decay_steps 5, l1 loss is the runtime's own choice).
    The first `warm` epochs are not timed (RAdam state allocation, pools)."""
    import torch
    from neuma_amd.train import finetune_constitutive, simulate_video
    dev = rt.device
    v_keep = rt.v0
    try:        # ground truth = the video of a perturbed start, as in measure_epoch
        g = torch.Generator().manual_seed(2)
        rt.v0 = (rt.v0.detach() + (0.02 * torch.randn(rt.v0.shape, generator=g)).to(dev)).contiguous()
        with torch.no_grad():
            gt = simulate_video(rt, frames)
    finally:
        rt.v0 = v_keep
    keep = [p.detach().clone() for p in rt.parameters()]
    for p in rt.parameters():
        p.grad = None
    stamps = []

    def log(_line):         # called once per epoch, behind float(loss) - the host has waited for the epoch's sweep, as the reference's log line does
        torch.cuda.synchronize(dev)
        stamps.append(time.perf_counter())

    sched = dict(type="cos", max_steps=1000, learning_rate_alpha=0.04)
    cfg = dict(elasticity_lr=0.005, elasticity_wd=0.0, elasticity_grad_max_norm=0.1, elasticity_scheduler=sched,
               plasticity_lr=0.0005, plasticity_wd=0.0, plasticity_grad_max_norm=0.1, plasticity_scheduler=sched,
               warmup_step=0, decay_init=0.5, decay_final=1.0, decay_steps=5, lambda_max_decay=0.4, num_epochs=epochs, num_frames=frames)
    torch.cuda.synchronize(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    try:
        losses = finetune_constitutive(rt, gt, cfg, tune_root=None, log=log)
        torch.cuda.synchronize(dev)
        moved = max(float((p.detach() - k).abs().max()) for p, k in zip(rt.parameters(), keep))
    finally:
        with torch.no_grad():       # the runtime goes back to the weights it was benchmarked with
            for p, k in zip(rt.parameters(), keep):
                p.copy_(k)
                p.grad = None
    timed = epochs - warm
    dt = stamps[-1] - stamps[warm - 1]
    fps = frames * timed / dt
    return {"what": "train.finetune_constitutive: epoch sweep + clip-norm x 2 + RAdam x 2 + LR schedules + loss read-back + LoRA re-merge, weights changing every epoch",
            "frames": frames, "substeps_per_frame": int(rt.S), "views_per_frame": int(rt.V), "epochs": epochs, "epochs_timed": timed,
            "frames_per_s": round(fps, 2), "ms_per_frame": round(1e3 / fps, 4),
            "epoch_ms_each": [round(1e3 * (b - a), 2) for a, b in zip(stamps[warm - 1:-1], stamps[warm:])],
            "ratio_to_epoch_metric": round(fps / epoch_fps, 3) if epoch_fps else None,
            "peak_hbm_GB_torch": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
            "losses": [float(f"{v:.6e}") for v in losses], "max_abs_change_of_a_lora_factor": moved,
            "settings": "experiments/configs/realworld/finetune-burger.yaml:67-113"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # >= 200 timed frames after 20 warm-up (~2 s)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="metric")
    ap.add_argument("--state", choices=("rest", "deformed", "impact"), default="rest",
                    help="state each timed frame starts from: rest (SURVEY §8d generator), deformed (F = I + 0.05 N(0,1)), "
                         "impact (post-floor-contact checkpoint) - see SceneRuntime.set_start_state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-op", action="store_true", help="use the per-operator drop-in path instead of the fused roll-out")
    ap.add_argument("--shard-sim", choices=("auto", "on", "off"), default="auto",
                    help="N > 1: particle-sharded simulation (neuma_amd/sim/shard.py) instead of a replicated one; "
                         "auto = whichever sim.shard.shard_cost_model estimates faster for this workload and world size "
                         "(measured per-substep latency at N / world particles + exchange machinery + assumed xGMI "
                         "all-reduce latency, DESIGN.md section 6)")
    ap.add_argument("--epoch-frames", type=int, default=19,
                    help="also measure a native multi-frame BPTT epoch (SceneRuntime.epoch; finetune.py:331-414 - the reference's "
                         "real unit of work) of this many frames at the workload's S and V, and a 400 x 1 x 1 epoch at bb size, "
                         "as a SECONDARY object `epoch` of the result line (one GPU only; 0 = skip).  19 x 20 x 3 = "
                         "configs/realworld/finetune-burger.yaml:107-113, 400 x 1 x 1 = configs/synthetic/finetune-bb.yaml:103-107")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))

    # stdout carries exactly ONE line (the JSON result of rank 0); everything else the libraries print goes to stderr
    # (at the file-descriptor level: gloo and the HIP runtime write from C)
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr

    # the shared library is a build artefact: build it if this checkout does not have it (rank 0 builds, the others wait)
    libpath = ROOT / "neuma_amd" / "lib" / "libneuma_hip.so"
    if not libpath.exists() and not os.environ.get("NEUMA_HIP_LIB"):
        if int(os.environ.get("RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            for _ in range(600):
                if libpath.exists():
                    break
                time.sleep(1.0)

    import torch
    import torch.distributed as dist
    from neuma_amd import synth, _lib
    from neuma_amd.harness import SceneRuntime

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)          # dry runs put several ranks on one GPU (gloo: RCCL refuses two ranks per device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("NEUMA_DIST_BACKEND", "nccl" if ndev >= world else "gloo")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    lib = _lib.lib()

    scene = synth.make_scene(args.workload)
    from neuma_amd.sim.shard import shard_cost_model, time_all_reduce_us
    measured = None
    if world > 1 and args.shard_sim == "auto" and os.environ.get("NEUMA_SHARD_CALIBRATE", "1") != "0":
        # start-up calibration: the cost model's inputs measured on THIS box at THIS world size - a few roll-outs at N and at
        # N / world particles, ~20 small all-reduces - instead of the one-GPU table and an assumed xGMI latency.  Every rank
        # measures; the maxima are used everywhere, so that all ranks take the same decision
        from neuma_amd.harness import measure_substep_us
        n_all = int(scene.x0.shape[0])
        # ... and both forms of the shared-block exchange: the all-reduce over the world and the buffers swapped with the
        # neighbour ranks only (nm_comm.exchange_peers_f32; needs the library's communicator) - the faster one is used
        from neuma_amd.sim.shard import create_library_comm, time_exchange_peers_us
        rc_comm = create_library_comm(dist.group.WORLD, dev) if backend == "nccl" else None
        t_ar = time_all_reduce_us(None, dev, count=1 << 16, rccl=rc_comm)
        t_px = time_exchange_peers_us(dev, rank, world, count=1 << 16, rccl=rc_comm)
        vals = torch.tensor([measure_substep_us(args.workload, n_all, dev), measure_substep_us(args.workload, -(-n_all // world), dev),
                             t_ar, -1.0 if t_px is None else t_px], dtype=torch.float64, device=dev)
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        measured = {"substep_us_full": round(float(vals[0]), 1), "substep_us_shard": round(float(vals[1]), 1),
                    "allreduce_us": round(float(vals[2]), 1), "allreduce_floats": 1 << 16}
        if float(vals[3]) > 0.0:
            measured["exchange_peers_us"] = round(float(vals[3]), 1)
            measured["allreduce_world_us"] = measured["allreduce_us"]
            if "NEUMA_SHARD_EXCHANGE" not in os.environ and float(vals[3]) < float(vals[2]):
                os.environ["NEUMA_SHARD_EXCHANGE"] = "peers"      # (read by GridExchange when the runtime shards its model)
                measured["allreduce_us"] = measured["exchange_peers_us"]      # the cost model's per-exchange latency
        measured["exchange"] = os.environ.get("NEUMA_SHARD_EXCHANGE", "allreduce")
    cost = shard_cost_model(int(scene.x0.shape[0]), world, int(scene.cfg["S"]), measured)
    shard_sim = world > 1 and (args.shard_sim == "on" or (args.shard_sim == "auto" and cost["shard"]))
    if world == 1 and args.shard_sim == "on" and os.environ.get("NEUMA_SHARD_FORCE") == "1":
        # overhead measurement on one GPU: a one-rank RCCL group, every collective of the sharded substep is issued
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        shard_sim = True
    rt = SceneRuntime(scene, dev, fused=not args.per_op, rank=rank, world=world, shard_sim=shard_sim,
                      group=dist.group.WORLD if world > 1 else None)
    if args.state != "rest":
        rt.set_start_state(args.state)
    rt.make_ground_truth()
    # touched grid nodes of the initial state (roofline accounting) and (Gaussian, tile) pairs per view
    with torch.no_grad():
        rt.rollout(*(t[rt.rows] for t in rt.start))
        _, rt.touched_nodes = rt.model.grid_stats()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    group = dist.group.WORLD if world > 1 else None
    rt.group = group

    params = rt.parameters()

    def zero_grads():
        for p in params:
            p.grad = None

    # ---- pick the dominant kernel from a fully profiled frame (outside the timed region)
    for _ in range(max(1, args.warmup)):
        zero_grads()
        rt.frame()
    sync()
    lib.nm_prof_reset()
    lib.nm_prof_enable(1, None)
    zero_grads()
    # per-kernel durations are taken with the views rendered one after another: on concurrent streams the compositing
    # kernels of different views stretch each other (450 -> 690 us) and the comparison between kernels would be skewed
    overlap, rt.overlap_views = rt.overlap_views, False
    rt.frame()
    sync()
    rt.overlap_views = overlap
    lib.nm_prof_enable(0, None)
    full = prof_table(lib)
    lib.nm_prof_reset()
    by_base = {}
    for name, (calls, ms) in full.items():      # template instances (k_material_bwd<...>) count as one kernel
        b = name.split("<")[0]
        by_base[b] = (by_base.get(b, (0, 0.0))[0] + calls, by_base.get(b, (0, 0.0))[1] + ms)
    dom_base = max(by_base.items(), key=lambda kv: kv[1][1])[0] if by_base else "k_render_bwd"

    # ---- timed region: K frames, HIP events on the dominant kernel only
    # (a pair of event records costs ~11 us of bubble around the launch - 19 x that per metric frame would be 3 % of the very
    #  rate being measured - so a kernel launched many times per frame is timed at every 8th launch: still hundreds of live
    #  samples over the timed region, spread over all substeps)
    dom_per_frame = by_base.get(dom_base, (1, 0.0))[0]
    prof_stride = 8 if dom_per_frame >= 8 else 1
    lib.nm_prof_enable(prof_stride, dom_base.encode())
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    import gc
    gc.collect()
    gc.disable()            # no collector pauses inside the timed region (a gen-2 pass costs milliseconds with this many tensors alive)
    t0 = time.perf_counter()
    last = None
    marks[0].record()
    for it in range(args.steps):
        zero_grads()
        last = rt.frame()
        marks[it + 1].record()          # per-frame GPU timeline (no host sync inside the timed region)
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    rt.flush()              # anything a frame reported late (exchange status, rasterizer overflow) surfaces here, loudly
    per_frame = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    lib.nm_prof_enable(0, None)
    dom = prof_table(lib)
    elapsed_local = elapsed
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    fps = args.steps / elapsed
    # every rank's loss is the partial sum over its own render stripes (harness.pixel_loss_rows): the ranks' parts add up to the
    # one-GPU loss of the same frame - summed here, outside the timed region, so that an N-rank line can be held against the 1-rank one
    loss_local = float(last.loss)
    loss_total = loss_local
    if world > 1:
        t = torch.tensor([loss_local], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        loss_total = float(t)

    # (Gaussian, tile) pairs, list entries and walked list entries of one full view, for the byte accounting
    D, vstats = 0.0, None
    try:
        from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F
        with torch.no_grad():
            m3 = compute_bindings_xyz(last.x, rt.start[0], rt.g_start, rt.bindings)
            dg = compute_bindings_F(last.F, rt.bindings)
            from neuma_amd.render import deform_cov_by_F, get_rasterizer, view_stats
            cov = deform_cov_by_F(rt._cov, dg)
            rast = get_rasterizer(rt.cameras[0], rt.gaussians.active_sh_degree, False, rt.background)
            vstats = view_stats(rast, m3.contiguous(), rt._opacity, shs=rt._shs, cov3D_precomp=cov)
            D = float(vstats["pairs"])
    except Exception as e:  # accounting only
        print(f"[bench] view statistics unavailable: {e}", file=sys.stderr)
    Dacc = vstats if (vstats is not None and world == 1) else D

    pmc, pmc_mfma, pmc_stale, pmc_digest = {}, {}, None, None
    try:        # HBM bytes per launch and matrix-pipe counters from the committed PMC passes (profiles/pmc_traffic.json, tools/make_profile_md.py)
        _p = json.loads((ROOT / "profiles" / "pmc_traffic.json").read_text())
        pmc_digest = _p.get("csrc_digest")
        # the counters are copies: valid only for the sources they were measured on.  A library built from other sources gets
        # no traffic figure at all (roofline.traffic null, no *_static fields) and the line says so
        pmc_stale = pmc_digest != _lib.csrc_digest()
        if not pmc_stale:
            pmc, pmc_mfma = _p.get("kernels", {}), _p.get("mfma", {})
        else:
            print(f"[bench] profiles/pmc_traffic.json was measured on sources {pmc_digest}, this checkout is {_lib.csrc_digest()}: "
                  "counter-derived fields dropped (pmc_stale)", file=sys.stderr)
    except Exception:
        pass
    roof = None
    if dom:
        calls = sum(v[0] for v in dom.values())
        ms = sum(v[1] for v in dom.values())
        name = dom_base + ("<*>" if any("<" in k for k in dom) else "")
        avg_s = ms / calls / 1e3
        base = dom_base
        if base in MATERIAL_FLOPS:
            # MFMA-bound: 2*(13*64 + 64*64 + 64*9) = 11 008 flop per particle per net forward; SURVEY §8d's figure for the
            # backward pass is 3x that (forward recompute + data-gradient + weight-gradient GEMMs); the pair kernel of the
            # reverse sweep runs two nets' backward passes per launch.  With the activation cache (default)
            # the kernel loads the second layer's activations instead of recomputing them and executes 2.15x: both fractions are
            # reported, `frac` uses SURVEY's figure as the contract asks
            flops = MATERIAL_FLOPS[base] * rt.n_local
            achieved = flops / avg_s / 1e12 if avg_s > 0 else 0.0
            roof = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 3), "peak": 157.3, "unit": "TFLOP/s",
                    "frac": round(achieved / 157.3, 5), "traffic": None, "launches": calls, "avg_us": round(avg_s * 1e6, 2),
                    "algorithmic_flops_per_launch": flops,
                    "note": "f32-input MFMA (v_mfma_f32_16x16x4_f32) peak = 157.3 TFLOP/s; the kernel also carries the SVD and GELU VALU work"}
            if not base.startswith("k_material_fwd"):
                cached = os.environ.get("NEUMA_ACT_CACHE", "auto") != "0" and not args.per_op
                roof["activation_cache"] = bool(cached)
                # with the cache the kernel runs 204 of the 284 MFMAs per tile (second and third layer of the forward pass loaded, first recomputed)
                roof["frac_of_executed_flops"] = round(achieved / 157.3 * (204.0 / 284.0 if cached else 1.0), 5)      # (replaced below by the counted flops when a PMC pass of this workload is on file)
        else:
            frac_view = 1.0
            if world > 1 and name.startswith(("k_render", "k_preprocess")):
                frac_view = 1.0 / world            # a rank composites 1/world of the tile rows per launch on average
            ab = algorithmic_bytes(name, rt, Dacc) * frac_view
            achieved = ab / avg_s / 1e9 if avg_s > 0 else 0.0
            roof = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 5), "traffic": None, "launches": calls, "avg_us": round(avg_s * 1e6, 2),
                    "algorithmic_bytes_per_launch": ab,
                    "note": "latency/VALU-bound kernel at this problem size (see DESIGN.md); HBM fraction reported as the contract asks"}

    if roof is not None:
        roof["launches"] = calls * prof_stride
        roof["launches_timed"] = calls
        roof["event_sampling"] = (f"HIP events around every {prof_stride}th launch of this kernel inside the timed region" if prof_stride > 1
                                  else "HIP events around every launch of this kernel inside the timed region")

    # ---- component rates, measured outside the timed region with HIP events on torch's stream
    rates = None
    if rank == 0 or world > 1:
        from neuma_amd.tune import compute_bindings_xyz, compute_bindings_F

        def gpu_ms(fn, reps=3):
            best = []
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); out = fn(); b.record(); torch.cuda.synchronize(dev)
                best.append(a.elapsed_time(b))
            return sorted(best)[len(best) // 2], out

        rw = rt.rows       # all particles, or this rank's range when the simulation is sharded

        def sim_fwd():
            with torch.no_grad():
                return rt.rollout(*(t[rw] for t in rt.start))

        def sim_fwdbwd():
            zero_grads()
            o = rt.rollout(*(t[rw] for t in rt.start))
            (o[0].sum() + o[3].sum()).backward()

        t_sf, o = gpu_ms(sim_fwd)
        t_sfb, _ = gpu_ms(sim_fwdbwd)
        with torch.no_grad():
            m3 = compute_bindings_xyz(rt.all_rows(o[0]), rt.start[0], rt.g_start, rt.bindings)
            dgr = compute_bindings_F(rt.all_rows(o[3]), rt.bindings)

        def ren_fwd():
            with torch.no_grad():
                return rt.pixel_loss(rt.render_view(m3, dgr, 0), rt.gt[0])

        def ren_fwdbwd():
            mm = m3.clone().requires_grad_(True)
            rt.pixel_loss(rt.render_view(mm, dgr, 0), rt.gt[0]).backward()

        t_rf, _ = gpu_ms(ren_fwd)
        t_rfb, _ = gpu_ms(ren_fwdbwd)
        S_ = rt.S
        rates = {"substeps_per_s_fwd": round(1e3 * S_ / t_sf, 1), "substeps_per_s_fwdbwd": round(1e3 * S_ / t_sfb, 1),
                 "renders_per_s_fwd": round(1e3 / t_rf, 1), "renders_per_s_fwdbwd": round(1e3 / t_rfb, 1),
                 "note": "full-image renders (1 view incl. loss) and whole-particle-set substeps on one GPU"}

    # ---- N > 1: what every rank spent where, so that a scaling run explains itself (one line per rank on stderr + `per_rank`)
    per_rank = None
    if world > 1:
        from neuma_amd.sim.shard import all_reduce_sum_, gather_rows

        def coll_us(fn, reps=20):
            for _ in range(3):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev); dist.barrier()
            a.record()
            for _ in range(reps):
                fn()
            b.record(); torch.cuda.synchronize(dev)
            return round(1e3 * a.elapsed_time(b) / reps, 1)

        colls = {}
        k3 = torch.zeros(rt.bindings.K, 3, dtype=torch.float32, device=dev)
        colls["all_reduce_K_x_3_us"] = coll_us(lambda: all_reduce_sum_(k3, group))           # the frame's gradient all-reduce
        lora = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
        colls["all_reduce_lora_grads_us"] = coll_us(lambda: all_reduce_sum_(lora, group))    # reduce_param_grads (sharded simulation)
        if rt.shard_sim:
            xl = rt.start[0][rt.rows].contiguous()
            Fl = rt.start[3][rt.rows].contiguous()
            with torch.no_grad():
                colls["gather_rows_x_us"] = coll_us(lambda: gather_rows(xl, rt.N, group))
                colls["gather_rows_F_us"] = coll_us(lambda: gather_rows(Fl, rt.N, group))
        jobs_now = getattr(rt, "_last_jobs", None)
        mine = {"rank": rank, "ms_per_frame": round(1e3 * elapsed_local / args.steps, 4), "particles": int(rt.n_local),
                "shard_sim": bool(rt.shard_sim), "rates": rates, "collectives": colls,
                "dominant_kernel_us": (round(1e3 * sum(v[1] for v in dom.values()) / max(1, sum(v[0] for v in dom.values())), 2) if dom else None),
                "render_jobs": jobs_now}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
        if rank == 0:
            for g in gathered:
                print(f"[bench] rank {g['rank']}: {g['ms_per_frame']} ms/frame, {g['particles']} particles, "
                      f"rates {g['rates']}, collectives {g['collectives']}", file=sys.stderr)

    epoch = None
    if rank == 0 and world == 1 and args.epoch_frames > 0 and not args.per_op:
        try:
            epoch = {"unit": "frames/s inside one epoch = F frames forward (state flowing from frame to frame, renders of frame f under "
                             "the simulation of frame f+1) + one reverse sweep; finetune.py:331-414",
                     args.workload: measure_epoch(rt, args.epoch_frames, fps)}
            from neuma_amd import rollout as _R
            _R.trim_pool()                   # (the pooled cache buffers of one measurement would count towards the next one's peak)
            # the same epoch without the activation cache: the speed / memory trade as a measured pair (the reference: "an 80 GB
            # A100", README.md:202; experiments/finetune.py:331-414)
            epoch[args.workload + " recompute"] = measure_epoch(rt, args.epoch_frames, fps, reps=3, recompute=True)
            # a training run on the same runtime: the epoch above re-runs identical weights; this one steps them
            try:
                _R.trim_pool()
                epoch["train"] = measure_training(rt, args.epoch_frames, epoch[args.workload]["frames_per_s"])
            except Exception as e:
                import traceback
                print(f"[bench] training leg failed: {e}\n{traceback.format_exc()}", file=sys.stderr)
                epoch["train"] = {"error": f"{type(e).__name__}: {e}"}
            if args.workload != "bb":
                _R.trim_pool()
                rt_bb = SceneRuntime(synth.make_scene("bb"), dev)
                rt_bb.make_ground_truth()
                for _ in range(20):
                    rt_bb.frame()
                torch.cuda.synchronize(dev)
                t0b = time.perf_counter()
                for _ in range(200):
                    rt_bb.frame()
                torch.cuda.synchronize(dev)
                epoch["bb"] = measure_epoch(rt_bb, 400, 200.0 / (time.perf_counter() - t0b))
                del rt_bb
        except Exception as e:  # secondary measurement: never takes the headline down
            import traceback
            print(f"[bench] epoch measurement failed: {e}\n{traceback.format_exc()}", file=sys.stderr)
            epoch = {"error": f"{type(e).__name__}: {e}"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # reported at N=1 only
        try:
            from oracle import cpu_baseline as cbaseline
            cpu = cbaseline.time_frame_sample(scene, rt)
        except Exception as e:
            print(f"[bench] cpu baseline unavailable: {e}", file=sys.stderr)

    # PMC-derived fields are NOT measurements of this run: they are copied from the committed rocprofv3 counter passes of the
    # same workload and build (profiles/pmc_traffic.json, made by tools/make_profile_md.py) and marked "static": true
    if roof is not None and roof["kernel"].split("<")[0] in pmc and args.workload == "metric" and world == 1:
        roof["traffic"] = pmc[roof["kernel"].split("<")[0]]["hbm_bytes_per_launch"]
        roof["traffic_static"] = {"static": True, "source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                                                            "this workload (2 x FETCH_SIZE + WRITE_SIZE per launch)"}
    if roof is not None and roof["kernel"].split("<")[0] in pmc_mfma and args.workload == "metric" and world == 1:
        m = pmc_mfma[roof["kernel"].split("<")[0]]
        roof["counters_static"] = {"static": True, "mfma_busy_pct_of_simd_cycles": m["mfma_busy_pct_of_simd_cycles"],
                                   "mfma_flops_counted": m["mfma_flops"],
                                   "source": "profiles/pmc_traffic.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 ... "
                                             "pass of this workload; MOPS x 512 = flops per launch)"}
        if roof.get("avg_us"):      # the flops the kernel executes (counted) over THIS run's measured duration
            roof["frac_of_executed_flops"] = round(m["mfma_flops"] / (roof["avg_us"] * 1e-6) / 1e12 / 157.3, 5)
    devices = [f"rank {rank}: cuda:{local} {torch.cuda.get_device_name(dev)}"]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
    if rank == 0:
        cfg = scene.cfg
        total_ms = sum(v[1] for v in full.values()) or 1.0
        out = {
            "metric": "sim+render frames/sec (fwd+bwd), 100k pts / 128^3 grid / 1080p",
            "value": round(fps, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['N']} particles / {cfg['G']}^3 grid / {cfg['K']} Gaussians / "
                                   f"{cfg['W']}x{cfg['H']}", "substeps_per_frame": cfg["S"], "views_per_frame": cfg["V"],
                       "sh_degree": cfg["sh"], "material": cfg["mat"] + "_0300 + LoRA r16", "path": "per-op" if args.per_op else ("fused-rollout (library-level sharded loop)" if shard_sim else "fused-rollout"),
                       "parallelism": (("particle-sharded sim" if shard_sim else "replicated sim") + " + render stripes") if world > 1 else "single GPU",
                       "start_state": rt.state_kind,
                       "touched_grid_nodes": int(rt.touched_nodes), "gaussian_tile_pairs_per_view": int(D)},
            "world": world, "backend": ({"nccl": "rccl"}.get(backend, backend) if world > 1 else None),
            "shard_collectives": getattr(getattr(rt.model, "exchange", None), "link_backend", None),
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
            "devices": devices,
            "roofline": roof,
            "pmc_stale": pmc_stale, "pmc_csrc_digest": pmc_digest, "csrc_digest": _lib.csrc_digest(),
            "cpu_baseline": cpu,
            "epoch": epoch,
            "kernel_breakdown_ms_per_frame": {k: round(v[1], 3) for k, v in sorted(full.items(), key=lambda kv: -kv[1][1])[:12]},
            "kernel_time_fraction_of_frame": round(total_ms / (1e3 * elapsed / args.steps), 3),
            "kernel_time_note": "sum of the HIP-event durations of every launch of ONE profiled frame (its views rendered one after "
                                "the other so that the events do not overlap) over the timed frame time; the timed frames run their V "
                                "views on V streams side by side, so the sum exceeds the frame (> 1)",
            "timed_region_s": round(elapsed, 4),
            "view_stats": vstats,
            "shard_cost_model": cost if world > 1 else None,
            "per_rank": per_rank,
            "kernel_rooflines": kernel_rooflines(full, rt, Dacc, *((pmc, pmc_mfma) if (args.workload == "metric" and world == 1) else (None, None))),
            "kernel_rooflines_static_note": "traffic_static / frac_of_executed_flops / mfma_busy_pct: copied from profiles/pmc_traffic.json "
                                            "(rocprofv3 --pmc passes of this workload and build), not measured by this run",
            "roofline_frame": frame_roofline(full, rt, Dacc, elapsed / args.steps),
            "frame_ms_gpu": {"median": round(per_frame[len(per_frame) // 2], 3), "p10": round(per_frame[len(per_frame) // 10], 3),
                             "p90": round(per_frame[(9 * len(per_frame)) // 10], 3)},
            "rates": rates,
            "loss": loss_total,
            "loss_rank0_part": loss_local,
        }
    else:
        out = None
    if dist.is_initialized():
        dist.barrier()
        from neuma_amd.sim.shard import close_library_comms
        close_library_comms()
        dist.destroy_process_group()
    if out is not None:
        # RCCL writes its version banner to the C stdout buffer; push it out first so that the JSON is the last line
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), file=result_out, flush=True)


if __name__ == "__main__":
    main()
