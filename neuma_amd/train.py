"""Training harness around the hot path: LoRA fine-tuning of the two constitutive nets by BPTT through
simulation + rendering.  Counterpart of /root/reference/experiments/finetune.py:234-488 (finetune_constitutive)
and modules/tune/scheduler/__init__.py:

  * two RAdam optimisers + LambdaLR schedules (cosine / exponential, scheduler/__init__.py:29-118), optional warm-up
    (finetune.py:345-351), schedulers stepped once per epoch after warm-up (482-484);
  * roll-out loss decay  rate(epoch) = decay_init + (decay_final - decay_init) * min(epoch / (lambda * num_epochs), 1),
    weight of frame f = rate ** ((f - 1) // decay_steps)                                   (finetune.py:353-358, 388);
  * previous particle / kernel positions are re-detached every rendered frame (391-392), frames listed in
    `exclude_steps` skip rendering AND the prev-state update (371-372);
  * one loss.backward() per epoch, clip_grad_norm_(error_if_nonfinite=True) per net, then the optimiser steps (413-427).
    The reference loop never zeroes the LoRA gradients between epochs (no zero_grad anywhere in 331-484; only stage A
    zeroes, :138), so `.grad` ACCUMULATES over the epochs before every clip + RAdam step.  That is reproduced by default
    (`accumulate_grads_like_reference=True`); set it to False for the conventional zero-per-epoch loop;
  * LoRA-only checkpoints `{epoch:04d}_lora.pt` = {'elasticity', 'plasticity', 'loss'} at epoch 1, every 10th and the
    last, newest `num_lora_ckpts` kept (470-480, natural sort); resume reloads the newest one with strict=False (299-309;
    the reference picks it with a lexicographic sort, which is the same file below 10 000 epochs - here the epoch number
    decides in both places).

Stage A, `optimize_init_velocity` (finetune.py:63-231): one global initial velocity (a 3-vector broadcast to every
particle, neuma_dataset.py:107-131) fitted by the same BPTT loop with RAdam + scheduler and the x-z prior
`lambda_reg * (mean|v_x| + mean|v_z|) / 2` switched on after 10 % of the epochs (207-214); result exported as `init.pt`
= {'init_x', 'init_v'} (neuma_dataset.py:115-118).

Everything numeric runs in the HIP kernels (neuma_amd.rollout / neuma_amd.tune / neuma_amd.render); this file is host
control flow only.
"""
import math
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.nn.utils import clip_grad_norm_
from torch.optim import RAdam, lr_scheduler

from .render import deform_cov_by_F
from .tune import compute_bindings_xyz, compute_bindings_F


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class CosineDecayScheduler(object):
    """scheduler/__init__.py:76-102: linear warm-up to `warm_up_end`, then cosine to learning_rate_alpha * lr."""

    def __init__(self, config) -> None:
        self.warm_up_end = _get(config, "warm_up_end") or 0
        self.max_steps = _get(config, "max_steps") or 1e5
        a = _get(config, "learning_rate_alpha")
        self.alpha = 0.05 if a is None else a

    def factor(self, step: int) -> float:
        if step < self.warm_up_end:
            return step / self.warm_up_end
        progress = (step - self.warm_up_end) / (self.max_steps - self.warm_up_end)
        return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - self.alpha) + self.alpha

    def get_scheduler(self, optimizer, lr_init: float):
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=self.factor)


class ExponentialDecayScheduler(object):
    """scheduler/__init__.py:29-73: linear / cosine ramp from lr_pre_warmup, then log-linear decay to lr_final."""

    def __init__(self, config) -> None:
        self.lr_pre_warmup = _get(config, "lr_pre_warmup") if _get(config, "lr_pre_warmup") is not None else 1e-8
        self.warmup_steps = _get(config, "warmup_steps") or 0
        self.max_steps = _get(config, "max_steps") or 1e5
        self.ramp = _get(config, "ramp") or "linear"
        self.lr_final = _get(config, "lr_final")

    def get_scheduler(self, optimizer, lr_init: float):
        lr_final = lr_init if self.lr_final is None else self.lr_final

        def func(step):
            if step < self.warmup_steps:
                if self.ramp == "cosine":
                    lr = self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * math.sin(
                        0.5 * math.pi * min(max(step / self.warmup_steps, 0), 1))
                else:
                    lr = self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * step / self.warmup_steps
            else:
                t = min(max((step - self.warmup_steps) / (self.max_steps - self.warmup_steps), 0), 1)
                lr = math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
            return lr / lr_init

        return lr_scheduler.LambdaLR(optimizer, lr_lambda=func)


def fetch_scheduler(config):
    """scheduler/__init__.py:105-118"""
    t = _get(config, "type")
    if t == "exp":
        return ExponentialDecayScheduler(config)
    if t == "cos":
        return CosineDecayScheduler(config)
    raise ValueError(f"Scheduler {t} not supported.")


def native_epoch_ok(rt) -> bool:
    """The runtime can run an epoch natively (SceneRuntime.epoch): one GPU, LoRA factors the only trainable tensors;
    NEUMA_NATIVE_EPOCH=0 keeps the composition of autograd nodes (video_loss)."""
    import os
    return (os.environ.get("NEUMA_NATIVE_EPOCH", "1") != "0" and hasattr(rt, "epoch") and getattr(rt, "world", 1) == 1
            and callable(getattr(rt, "_lean_ok", None)) and rt._lean_ok())


def epoch_weights(c, decay_rate: float):
    """(weights, frame_steps) of an epoch as video_loss applies them: weights[f] = decay_rate ** (f // decay_steps) for frame
    f = cur_step - 1, None for a frame whose dataset id is in exclude_steps (finetune.py:369-372, 386-389)."""
    nframes = int(c["num_frames"])
    frame_ids = list(c["steps"]) if c.get("steps") is not None else list(range(nframes + 1))
    weights = [None if frame_ids[cs] in c["exclude_steps"] else decay_rate ** ((cs - 1) // c["decay_steps"]) for cs in range(1, nframes + 1)]
    return weights, [frame_ids[cs] for cs in range(1, nframes + 1)]


def _flush(rt) -> None:
    """Before an optimizer step: wait for the runtime's deferred status words (a render whose lists overflowed composited the
    background only; a sharded substep whose exchange was incomplete summed too little) and raise instead of stepping on them."""
    flush = getattr(rt, "flush", None)
    if flush is not None:
        flush()


def rollout_decay_rate(cfg, epoch: int) -> float:
    """finetune.py:353-358"""
    lam = _get(cfg, "lambda_max_decay", 0)
    ratio = min((1.0 / lam) * epoch / _get(cfg, "num_epochs"), 1.0) if lam > 0 else 1.0
    return _get(cfg, "decay_init") + (_get(cfg, "decay_final") - _get(cfg, "decay_init")) * ratio


DEFAULT_CFG = dict(   # experiments/configs/synthetic/finetune-bb.yaml:61-107 (sizes reduced by the caller)
    elasticity_lr=0.008, elasticity_wd=0.0, elasticity_grad_max_norm=1.0,
    elasticity_scheduler=dict(type="cos", max_steps=1000, learning_rate_alpha=0.025),
    plasticity_lr=0.0008, plasticity_wd=0.0, plasticity_grad_max_norm=1.0,
    plasticity_scheduler=dict(type="cos", max_steps=1000, learning_rate_alpha=0.025),
    warmup_step=0, decay_init=0.5, decay_final=1.0, decay_steps=80, lambda_max_decay=0.33,
    num_epochs=1000, num_frames=400, exclude_steps=(), steps=None, num_lora_ckpts=3, resume=False,
    accumulate_grads_like_reference=True,   # finetune.py:331-484 has no zero_grad: gradients add up across epochs
    overlap_render=None,    # not in the reference: render frame f on a second stream while frame f+1 simulates.  None = the
                            # path's own default (native epoch: on; video_loss's autograd nodes: off); True / False force it on both
)


@torch.no_grad()
def simulate_video(rt, num_frames: int, views: Optional[Sequence[int]] = None, deform_cov: bool = True) -> List[List[torch.Tensor]]:
    """Forward-only roll-out (render.py:304-332 order) that renders every frame: used to synthesise ground truth.
    Returns frames[f][view] images for f = 1..num_frames."""
    views = list(range(rt.V)) if views is None else list(views)
    x, v, C, F = rt.x0, rt.v0, rt.C0, rt.F0
    de_prev = (x - rt.center) / rt.size
    g_prev = rt.gaussians.get_xyz
    out = []
    for _ in range(num_frames):
        x, v, C, F = rt.rollout(x, v, C, F)
        de_x = (x - rt.center) / rt.size
        means3D = compute_bindings_xyz(de_x, de_prev, g_prev, rt.bindings)
        dg = compute_bindings_F(F, rt.bindings) if deform_cov else None
        out.append([rt.render_view(means3D, dg, vi).clone() for vi in views])
        de_prev, g_prev = de_x.clone(), means3D.clone()
    return out


def video_loss(rt, gt_frames, c, decay_rate: float, views: Sequence[int], deform_cov: bool = True,
               overlap_render: bool = False) -> torch.Tensor:
    """One epoch's forward pass (finetune.py:334-392): roll the simulation out frame by frame from the initial state,
    bind + render the requested views of every frame, accumulate the decayed pixel loss.
    NB (reference semantics, tune/utils.py:353-373): the covariance push-forward by F is not differentiable, so the
    gradient of this loss deliberately omits the d(image)/dF path; deform_cov=False renders with the rest covariances
    (what the reference does for its first frame), which makes the returned gradient the exact one.
    overlap_render: binding + rendering + loss of frame f are enqueued on a second HIP stream, so they execute while the
    (latency-bound) simulation of frame f+1 runs on the main stream; autograd replays the same stream assignment in the
    backward pass.  Results are identical; only the schedule changes."""
    nframes = int(c["num_frames"])
    # dataset frame ids of the roll-out steps (dataset.steps, finetune.py:369): `exclude_steps` lists FRAME IDS
    frame_ids = list(c["steps"]) if c.get("steps") is not None else list(range(nframes + 1))
    x, v, C, F = rt.x0, rt.v0, rt.C0, rt.F0
    de_prev = ((x - rt.center) / rt.size).clone().detach()
    g_prev = rt.gaussians.get_xyz.clone().detach()
    main = torch.cuda.current_stream(rt.device) if overlap_render else None
    side = None
    if overlap_render:
        side = getattr(rt, "_render_stream", None)
        if side is None:
            side = rt._render_stream = torch.cuda.Stream(device=rt.device)
    terms = []
    for cur_step in range(1, nframes + 1):
        x, v, C, F = rt.rollout(x, v, C, F, step0=(cur_step - 1) * rt.S)      # `substeps` substeps (362-364)
        if frame_ids[cur_step] in c["exclude_steps"]:
            continue                                                          # finetune.py:369-372
        w = decay_rate ** ((cur_step - 1) // c["decay_steps"])
        if overlap_render:
            side.wait_stream(main)
            for t in (x, F):
                t.record_stream(side)                                         # read on `side` after `main` moves on
        with torch.cuda.stream(side) if overlap_render else _nullcontext():
            de_x = (x - rt.center) / rt.size
            means3D = compute_bindings_xyz(de_x, de_prev, g_prev, rt.bindings)
            dg = compute_bindings_F(F, rt.bindings) if deform_cov else None
            lf = torch.zeros((), device=rt.device)
            # covariance push-forward once per frame, not per view
            cov = deform_cov_by_F(rt._cov, dg) if dg is not None else rt._cov
            fid = frame_ids[cur_step]
            for i, vi in enumerate(views):
                render = rt.render_view(means3D, None, vi, cov=cov, step=fid)
                lf = lf + w * rt.pixel_loss(render, gt_frames[cur_step - 1][i])
            terms.append(lf)
            de_prev = de_x.clone().detach()
            g_prev = means3D.clone().detach()
    if overlap_render:
        main.wait_stream(side)
        for t in terms:
            t.record_stream(main)
    loss_rgb = torch.zeros((), device=rt.device)
    for t in terms:
        loss_rgb = loss_rgb + t
    return loss_rgb


class _nullcontext(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def lora_checkpoints(tune_root: Path) -> List[Path]:
    """`*_lora.pt` files of a run, oldest first by epoch number (natsorted in finetune.py:478)."""
    def epoch_of(p: Path):
        head = p.stem.split("_")[0]
        return (int(head) if head.isdigit() else -1, p.name)
    return sorted(Path(tune_root).glob("*_lora.pt"), key=epoch_of)


def finetune_constitutive(rt, gt_frames: List[List[torch.Tensor]], cfg: Optional[Dict] = None, tune_root: Optional[Path] = None,
                          views: Optional[Sequence[int]] = None, log=None) -> List[float]:
    """finetune.py:234-488 on a SceneRuntime.  gt_frames[f-1][i] = ground-truth image of frame f for views[i].
    Returns the per-epoch losses."""
    c = dict(DEFAULT_CFG)
    c.update(cfg or {})
    views = list(range(rt.V)) if views is None else list(views)
    E, P = rt.elasticity, rt.plasticity
    if tune_root is not None:
        tune_root = Path(tune_root)
        tune_root.mkdir(parents=True, exist_ok=True)
        if c["resume"]:                                                          # finetune.py:299-309
            prev = lora_checkpoints(tune_root)
            if prev:
                ck = torch.load(prev[-1], map_location=rt.device)
                E.load_state_dict(ck["elasticity"], strict=False)
                P.load_state_dict(ck["plasticity"], strict=False)
    E.freeze_all_except_lora(); P.freeze_all_except_lora()
    e_opt = RAdam([p for p in E.parameters() if p.requires_grad], lr=c["elasticity_lr"], weight_decay=c["elasticity_wd"])
    p_opt = RAdam([p for p in P.parameters() if p.requires_grad], lr=c["plasticity_lr"], weight_decay=c["plasticity_wd"])
    e_sch = fetch_scheduler(c["elasticity_scheduler"]).get_scheduler(e_opt, c["elasticity_lr"])
    p_sch = fetch_scheduler(c["plasticity_scheduler"]).get_scheduler(p_opt, c["plasticity_lr"])
    losses = []
    for epoch in range(1, int(c["num_epochs"]) + 1):
        if c["warmup_step"] != 0 and epoch <= c["warmup_step"]:                  # finetune.py:345-351
            for opt, lr in ((e_opt, c["elasticity_lr"]), (p_opt, c["plasticity_lr"])):
                for g in opt.param_groups:
                    g["lr"] = lr * float(epoch) / c["warmup_step"]
        decay_rate = rollout_decay_rate(c, epoch)
        if not c["accumulate_grads_like_reference"]:
            e_opt.zero_grad(set_to_none=True); p_opt.zero_grad(set_to_none=True)
        if native_epoch_ok(rt):
            # the whole epoch - F frames forward, one reverse sweep - as two plain calls into the library (harness._epoch_forward /
            # _epoch_backward): what video_loss + loss.backward() compute through one autograd node per frame
            ew, esteps = epoch_weights(c, decay_rate)
            # (renders on a second stream under the next frame's simulation unless the configuration says otherwise: same
            #  results, tests/test_gpu_train.py::test_native_epoch...[False / True])
            loss_rgb = rt.epoch(gt_frames, ew, views=views, frame_steps=esteps, overlap=True if c.get("overlap_render") is None else bool(c["overlap_render"]))
        else:
            loss_rgb = video_loss(rt, gt_frames, c, decay_rate, views, overlap_render=bool(c.get("overlap_render") or False))
            loss_rgb.backward()
        _flush(rt)      # deferred reports (rasterizer capacity, sharded exchanges) raise HERE, before the gradients are used
        e_gn = clip_grad_norm_(E.parameters(), max_norm=c["elasticity_grad_max_norm"], error_if_nonfinite=True)
        e_opt.step()
        p_gn = clip_grad_norm_(P.parameters(), max_norm=c["plasticity_grad_max_norm"], error_if_nonfinite=True)
        p_opt.step()
        losses.append(float(loss_rgb))
        if log is not None:
            log(f"[Epoch {epoch}/{c['num_epochs']} | L rgb: {losses[-1]:.4e} | e-lr: {e_opt.param_groups[0]['lr']:.2e} | "
                f"e-gd: {float(e_gn):.2e} | p-lr: {p_opt.param_groups[0]['lr']:.2e} | p-gd: {float(p_gn):.2e} | decay: {decay_rate:.2f}]")
        if tune_root is not None and (epoch == 1 or epoch % 10 == 0 or epoch == c["num_epochs"]):   # finetune.py:470-480
            torch.save({"elasticity": E.lora_state_dict(), "plasticity": P.lora_state_dict(), "loss": losses[-1]},
                       tune_root / f"{epoch:04d}_lora.pt")
            files = lora_checkpoints(tune_root)
            if len(files) > c["num_lora_ckpts"]:
                files[0].unlink()
        if c["warmup_step"] == 0 or epoch > c["warmup_step"]:                   # finetune.py:482-484
            e_sch.step(); p_sch.step()
    return losses


VELOCITY_CFG = dict(   # experiments/configs/*/finetune-*.yaml `velocity:` block (sizes reduced by the caller)
    num_epochs=100, num_frames=5, lr=0.1, scheduler=dict(type="cos", max_steps=100, learning_rate_alpha=0.05),
    lambda_reg=None, reg_all=False, pixel_loss="l2", steps=None,
)


def optimize_init_velocity(rt, gt_frames: List[List[torch.Tensor]], cfg: Optional[Dict] = None, tune_root: Optional[Path] = None,
                           views: Optional[Sequence[int]] = None, log=None):
    """finetune.py:63-231 on a SceneRuntime.  gt_frames[f-1][i] = ground-truth image of frame f for views[i].
    Returns (init_v (3,) tensor, per-epoch rgb losses); rt.v0 is set to the broadcast result."""
    from .tune import l1_loss, l2_loss
    c = dict(VELOCITY_CFG)
    c.update(cfg or {})
    views = list(range(rt.V)) if views is None else list(views)
    if tune_root is not None and (Path(tune_root) / "init.pt").exists():          # finetune.py:76-85
        d = torch.load(Path(tune_root) / "init.pt", map_location="cpu")
        rt.x0 = d["init_x"].to(rt.device).float().contiguous()
        rt.v0 = d["init_v"].to(rt.device).float().contiguous()
        return rt.v0.mean(0), []
    pixel_loss = {"l1": l1_loss, "l2": l2_loss}[c["pixel_loss"]]
    init_v = torch.nn.Parameter(torch.zeros(3, device=rt.device))                 # neuma_dataset.py:128-131
    opt = RAdam([init_v], lr=c["lr"])
    sch = fetch_scheduler(c["scheduler"]).get_scheduler(opt, c["lr"])
    losses = []
    nframes = int(c["num_frames"])
    frame_ids = list(c["steps"]) if c.get("steps") is not None else list(range(nframes + 1))     # dataset.steps, finetune.py:155
    for epoch in range(1, int(c["num_epochs"]) + 1):
        opt.zero_grad(set_to_none=True)
        x, C, F = rt.x0, rt.C0, rt.F0
        v = init_v.unsqueeze(0).expand(rt.N, -1) + 0.0                            # finetune.py:148
        de_prev = ((x - rt.center) / rt.size).clone().detach()
        g_prev = rt.gaussians.get_xyz.clone().detach()
        loss_rgb = torch.zeros((), device=rt.device)
        for cur_step in range(1, nframes + 1):
            x, v, C, F = rt.rollout(x, v, C, F, step0=(cur_step - 1) * rt.S)
            de_x = (x - rt.center) / rt.size
            means3D = compute_bindings_xyz(de_x, de_prev, g_prev, rt.bindings)
            dg = compute_bindings_F(F, rt.bindings)
            for i, vi in enumerate(views):
                loss_rgb = loss_rgb + pixel_loss(rt.render_view(means3D, dg, vi, step=frame_ids[cur_step]), gt_frames[cur_step - 1][i])
            de_prev, g_prev = de_x.clone().detach(), means3D.clone().detach()
        if c["lambda_reg"] is not None and epoch > int(0.1 * c["num_epochs"]):    # finetune.py:207-214
            if c["reg_all"]:
                loss_reg = c["lambda_reg"] * init_v.abs().mean()
            else:
                loss_reg = c["lambda_reg"] * (init_v[0].abs() + init_v[2].abs()) / 2.0
        else:
            loss_reg = torch.zeros_like(loss_rgb)
        (loss_rgb + loss_reg).backward()
        _flush(rt)      # (as in stage B: no optimizer step on a frame whose render overflowed)
        opt.step()
        losses.append(float(loss_rgb))
        if log is not None:
            log(f"[Epoch {epoch}/{c['num_epochs']} | L rgb: {losses[-1]:.4e}, reg: {float(loss_reg):.4e} | "
                f"lr: {opt.param_groups[0]['lr']:.4f} | init_v: {init_v.detach().cpu().tolist()}]")
        sch.step()
    init_v.requires_grad_(False)
    rt.v0 = init_v.detach().unsqueeze(0).expand(rt.N, -1).contiguous()
    if tune_root is not None:                                                     # neuma_dataset.py:115-118
        Path(tune_root).mkdir(parents=True, exist_ok=True)
        torch.save({"init_x": rt.x0.cpu(), "init_v": rt.v0.cpu()}, Path(tune_root) / "init.pt")
    return init_v.detach(), losses
