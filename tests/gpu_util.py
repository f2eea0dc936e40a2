"""Shared helpers for the GPU parity tests (oracle = checker, HIP path = thing under test)."""
import numpy as np
import torch

from oracle import mpm as om


def dev():
    return torch.device("cuda", 0)


class Measured(float):
    """A measured error that publishes itself when it is compared with its bound: `assert rel_max(a, b) < 3e-6` prints
    `PARITY {case, tensor, measured, bound}` and appends it to gpurun_out/parity_measured.jsonl exactly as `parity()` does
    (case = the running pytest item, tensor = kind + source line of the comparison), so every comparison of every GPU test is
    in the log without a second spelling of it.  Arithmetic on the value gives a plain float (nothing is logged for it)."""

    def __new__(cls, value, kind="err"):
        obj = float.__new__(cls, value)
        obj.kind = kind
        return obj

    def _publish(self, bound):
        import inspect
        import os
        try:
            fr = inspect.stack()[2]
            where = f"{os.path.basename(fr.filename)}:{fr.lineno}"
            src = (fr.code_context[0].strip() if fr.code_context else "")[:110]
        except Exception:       # noqa: BLE001 - logging must never fail a test
            where, src = "?", ""
        case = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]
        _log({"case": case, "tensor": f"{self.kind} @ {where}: {src}", "measured": float(self), "bound": float(bound)})

    def __truediv__(self, scale):         # (an error over a scale is still the measured quantity)
        return Measured(float(self) / float(scale), self.kind + " / scale")

    def __lt__(self, bound):
        self._publish(bound)
        return float(self) < float(bound)

    def __le__(self, bound):
        self._publish(bound)
        return float(self) <= float(bound)


def rel_max(a: torch.Tensor, b: torch.Tensor) -> float:
    """max-norm relative error of a (GPU fp32) against b (oracle)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return Measured(float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)), "rel_max")


def abs_max(a, b) -> float:
    return Measured(float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max()), "abs_max")


def measured(value, kind="err") -> float:
    """wrap any error figure a test computes by hand so that its comparison is logged like the others"""
    return Measured(float(value), kind)


def far_image(img: torch.Tensor) -> torch.Tensor:
    """A target image FAR from anything the scene renders (a smooth pattern, same shape / device): equivalence tests of two
    schedules or code paths take their loss against it.  Two runs of one frame differ in the last bits of the particle states
    (order of the scatters' float atomics), and a last-bit difference now and then flips the rasterizer's alpha >= 1/255 cut-off
    for a pixel.  Against a near ground truth (loss ~ 1e-6, residuals ~ 1e-3) such a flip is 1e-5 .. 1e-3 of the loss and of
    every gradient - which is why these tests once had to retry; against residuals of ~ 0.5 per pixel it is ~ 1e-8 of the loss
    and below the atomics' rounding noise in the gradients, so the tight bound can be demanded of EVERY run."""
    c, h, w = img.shape[-3:]
    yy = torch.linspace(0.0, 1.0, h, device=img.device).view(1, h, 1)
    xx = torch.linspace(0.0, 1.0, w, device=img.device).view(1, 1, w)
    ch = torch.arange(c, device=img.device, dtype=torch.float32).view(c, 1, 1)
    pat = 0.5 + 0.4 * torch.sin(6.0 * xx + 2.0 * ch) * torch.cos(5.0 * yy - ch)
    return pat.expand_as(img).contiguous().to(img.dtype)


def far_ground_truth(rt) -> None:
    """Replace a SceneRuntime's ground-truth views by far_image targets (after make_ground_truth)."""
    rt.gt = [far_image(g) for g in rt.gt]


def mpm_case(N=4096, G=32, seed=0, bc="noslip", near_wall=True, disabled=True, dt=1e-3):
    g = torch.Generator().manual_seed(seed)
    const = om.MPMConstant(num_grids=G, dt=dt, bound=1, gravity=(0.0, -9.8, 0.0), eps=6e-7, bc=bc)
    dx = const.dx
    x = 0.3 + 0.4 * torch.rand(N, 3, generator=g, dtype=torch.float64)
    if near_wall:
        k = N // 16
        x[:k] = 0.1 * dx + 0.9 * dx * torch.rand(k, 3, generator=g, dtype=torch.float64)          # low wall, < 1 cell
        x[k:2 * k] = 1.0 - 1.6 * dx - 0.9 * dx * torch.rand(k, 3, generator=g, dtype=torch.float64)  # high wall
        x[2 * k:3 * k, 1] = 0.1 * dx + 0.5 * dx * torch.rand(k, generator=g, dtype=torch.float64)    # floor contact
    v = torch.randn(N, 3, generator=g, dtype=torch.float64)
    C = 2.0 * torch.randn(N, 3, 3, generator=g, dtype=torch.float64)
    F = torch.eye(3, dtype=torch.float64)[None] + 0.1 * torch.randn(N, 3, 3, generator=g, dtype=torch.float64)
    S = 50.0 * torch.randn(N, 3, 3, generator=g, dtype=torch.float64)
    vol = torch.full((N,), (dx / 2) ** 3, dtype=torch.float64)
    rho = torch.full((N,), 1000.0, dtype=torch.float64)
    clip = torch.full((N,), 0.1, dtype=torch.float64)
    en = torch.ones(N, dtype=torch.int32)
    if disabled:
        en[N // 2: N // 2 + N // 10] = 0
    # sort by cell so that workgroups are spatially coherent (LDS-tile path); tests also run unsorted
    cell = torch.floor(x * G).long()
    key = (cell[:, 0] * G + cell[:, 1]) * G + cell[:, 2]
    o = torch.argsort(key)
    return const, vol, rho, clip, en[o].contiguous(), x[o].contiguous(), v[o].contiguous(), C[o].contiguous(), \
        F[o].contiguous(), S[o].contiguous()


def build_model(const, device):
    from neuma_amd.sim import MPMModelBuilder
    cfg = dict(gravity=list(const.gravity), bc=const.bc, num_grids=const.num_grids, dt=const.dt, bound=const.bound, eps=const.eps)
    return MPMModelBuilder().parse_cfg(cfg).finalize(device, requires_grad=True)


def build_statics(model, vol, rho, clip, en, device):
    st = model.statics(vol.shape[0])
    st.vol.copy_(vol.float()); st.rho.copy_(rho.float()); st.clip_bound.copy_(clip.float()); st.enabled.copy_(en)
    return st


def parity(case: str, name: str, measured: float, bound: float, noise=None) -> None:
    """Record one measured parity error next to the bound it is held to (and, where a fixture carries the reference's own fp32
    run, the distance of that run from its fp64 run): printed (`pytest -s`, or the captured log) and appended to
    gpurun_out/parity_measured.jsonl, from which tools/parity_table.py makes DESIGN.md's table.  Then the assertion."""
    rec = {"case": case, "tensor": name, "measured": float(measured), "bound": float(bound)}
    if noise is not None:
        rec["reference_fp32_vs_fp64"] = float(noise)
    _log(rec)
    assert float(measured) <= bound, rec


def _log(rec: dict) -> None:
    import json
    import os
    from pathlib import Path
    print("PARITY " + json.dumps(rec))
    out = Path(os.environ.get("NEUMA_PARITY_LOG", Path(__file__).resolve().parent.parent / "gpurun_out" / "parity_measured.jsonl"))
    try:
        out.parent.mkdir(parents=True, exist_ok=True)
        with open(out, "a") as fh:
            fh.write(json.dumps(rec) + "\n")
    except OSError:
        pass
