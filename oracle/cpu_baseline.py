"""Oracle (TEST INFRASTRUCTURE ONLY) — CPU baseline timing for bench.py's `cpu_baseline` leg.

The reference's own CPU path cannot be timed (warp-lang is absent; diff_gaussian_rasterization has no CPU
implementation), so the reported baseline is this oracle — a PyTorch-CPU *port* of the reference
algorithm — timed on the GPU box's host cores on a BOUNDED SAMPLE of the bench workload:

  sim     one full substep (elasticity net -> dense-grid p2g / grid_op / g2p -> plasticity net), forward and
          backward, on ALL particles of the workload                                  -> t_substep
  render  per-Gaussian preprocess of ALL Gaussians for one view, then front-to-back compositing + backward for
          three sampled 16-pixel tile rows (1/4, 1/2, 3/4 of the image height), each tile evaluated densely over
          the Gaussians whose tile rectangle covers it                                -> t_view ~ t_pre + gy * mean(t_row)

  frames/s (estimated) = 1 / (S * t_substep + V * t_view)

kind = "port", cores = torch.get_num_threads().  Never used as a fallback by the product.
"""
import math
import time

import torch

from . import material as omat
from . import mpm as om
from . import raster as orr


def _composite_tile(pp, idx_sorted, tx, ty, W, H, bg):
    """Dense front-to-back composite of one 16x16 tile over the (depth-sorted) Gaussians covering it."""
    rminx, rminy, rmaxx, rmaxy = pp["rect"]
    sel = idx_sorted[(rminx[idx_sorted] <= tx) & (rmaxx[idx_sorted] > tx) & (rminy[idx_sorted] <= ty) & (rmaxy[idx_sorted] > ty)]
    y0, x0 = ty * 16, tx * 16
    py, px = torch.meshgrid(torch.arange(y0, min(H, y0 + 16)), torch.arange(x0, min(W, x0 + 16)), indexing="ij")
    px, py = px.reshape(-1), py.reshape(-1)
    if sel.numel() == 0:
        return bg[None].expand(px.numel(), 3), 0
    xy, con, op, rgb = pp["xy"][sel], pp["conic"][sel], pp["opacity"][sel], pp["rgb"][sel]
    dt = xy.dtype
    dx = xy[None, :, 0] - px[:, None].to(dt)
    dy = xy[None, :, 1] - py[:, None].to(dt)
    power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
    a_raw = op[None] * torch.exp(torch.clamp_max(power, 0.0))
    alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()
    contrib = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a_eff = torch.where(contrib, alpha, torch.zeros_like(alpha))
    T_after = torch.cumprod(1.0 - a_eff, dim=1)
    stop = contrib & (T_after.detach() < 1e-4)
    keep = contrib & (torch.cumsum(stop.to(torch.int32), dim=1) == 0)
    T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], 1)
    wgt = torch.where(keep, alpha * T_before, torch.zeros_like(alpha))
    T_fin = torch.prod(torch.where(keep, 1.0 - alpha, torch.ones_like(alpha)), dim=1)
    return wgt @ rgb + T_fin[:, None] * bg[None], int(sel.numel())


def time_frame_sample(scene, rt=None, max_seconds: float = 60.0):
    """Returns the `cpu_baseline` JSON object."""
    cfg = scene.cfg
    torch.manual_seed(0)
    # many tiny ops: more than ~16 intra-op threads only adds overhead on a 128/256-core host
    torch.set_num_threads(min(16, torch.get_num_threads()))
    from neuma_amd import synth          # data generator only (host numpy) — not a compute path
    w = synth.load_base_weights(cfg["mat"])
    We = [torch.tensor(a) for a in w["e"]]
    Wp = [torch.tensor(a) for a in w["p"]]
    N, G = scene.x0.shape[0], cfg["G"]
    const = om.MPMConstant(G, cfg["dt"], 1, (0.0, -9.8, 0.0), 6e-7, "noslip")
    x = torch.tensor(scene.x0).requires_grad_(True)
    v = torch.tensor(scene.v0).requires_grad_(True)
    C = torch.zeros(N, 3, 3, requires_grad=True)
    F = (torch.eye(3).repeat(N, 1, 1) + 0.01 * torch.randn(N, 3, 3)).requires_grad_(True)
    vol = torch.full((N,), scene.vol); rho = torch.full((N,), 1000.0); clip = torch.full((N,), 0.1)
    en = torch.ones(N, dtype=torch.int32)
    Weg = [t.clone().requires_grad_(True) for t in We]
    Wpg = [t.clone().requires_grad_(True) for t in Wp]

    def substep():
        stress = omat.elasticity(F, Weg)
        xn, vn, Cn, Fn = om.step(const, vol, rho, clip, en, x, v, C, F, stress)
        Fn = omat.plasticity(Fn, Wpg, 1e-3)
        loss = xn.sum() + vn.sum() + Cn.sum() + Fn.sum()
        torch.autograd.grad(loss, [x, v, C, F] + Weg + Wpg)

    substep()  # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 and time.perf_counter() - t0 < max_seconds / 3:
        substep()
        reps += 1
    t_sub = (time.perf_counter() - t0) / max(reps, 1)

    # ---- render sample
    W, H = cfg["W"], cfg["H"]
    cam = synth.ring_cameras(cfg["V"], W, H)[0]
    s = orr.Settings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.ones(3), 1.0, cam.world_view_transform,
                     cam.full_proj_transform, cfg["sh"], cam.camera_center)
    means = torch.tensor(scene.g_xyz).requires_grad_(True)
    cov = orr.build_cov3D(torch.exp(torch.tensor(scene.g_logscale)), torch.tensor(scene.g_rot))
    op = torch.sigmoid(torch.tensor(scene.g_opacity_logit))
    shs = torch.tensor(scene.g_sh)
    t0 = time.perf_counter()
    pp = orr.preprocess(s, means, cov, op, shs=shs)
    vis = torch.nonzero(pp["visible"]).reshape(-1)
    order = torch.argsort(pp["depth"].detach()[vis], stable=True)
    idx_sorted = vis[order]
    t_pre = time.perf_counter() - t0
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rows = sorted(set([gy // 4, gy // 2, (3 * gy) // 4]))
    t_rows, pairs = [], 0
    bg = torch.ones(3)
    budget = time.perf_counter() + max_seconds / 2
    for ty in rows:
        t0 = time.perf_counter()
        loss = torch.zeros(())
        for tx in range(gx):
            img, npair = _composite_tile(pp, idx_sorted, tx, ty, W, H, bg)
            pairs += npair
            loss = loss + (img * img).sum()
        (g,) = torch.autograd.grad(loss, means, retain_graph=True)
        t_rows.append(time.perf_counter() - t0)
        if time.perf_counter() > budget:
            break
    t_row = sum(t_rows) / len(t_rows)
    t_view = 2.0 * t_pre + gy * t_row            # preprocess forward + (about as much) backward
    S, V = cfg["S"], cfg["V"]
    fps = 1.0 / (S * t_sub + V * t_view)
    return {"value": round(fps, 6), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"torch-CPU oracle: {reps} full substep(s) fwd+bwd on all {N} particles (t={t_sub:.3f}s each, dense {G}^3 grid) + "
                      f"preprocess of all {scene.g_xyz.shape[0]} Gaussians (t={t_pre:.3f}s) + composite fwd+bwd of {len(t_rows)} of {gy} tile rows "
                      f"(mean t={t_row:.3f}s/row, {pairs} pairs); frame estimated as S*t_sub + V*(2*t_pre + {gy}*t_row) with S={S}, V={V}",
            "t_substep_s": round(t_sub, 4), "t_view_s": round(t_view, 4)}
