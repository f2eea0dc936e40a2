"""Particle-GS render operator: drop-in for the `diff_gaussian_rasterization` extension plus the glue of
/root/reference/modules/d3gs/gaussian_renderer/__init__.py:92-119 (get_rasterizer) and
/root/reference/modules/d3gs/utils/simulation_utils.py:25-48 (deform_cov_by_F).

    GaussianRasterizationSettings(NamedTuple)   12 fields, same order as the reference fills them
    GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                        scales=None, rotations=None, cov3D_precomp=None) -> (color, radii)
"""
import ctypes as C
import math
import os
from typing import NamedTuple, Optional, Tuple

import torch
import torch.autograd as autograd
import torch.nn as nn
from torch import Tensor

from .. import _lib as L


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


def _c_cfg(s: GaussianRasterizationSettings, tile_rows: Optional[Tuple[int, int]] = None) -> L.nm_raster_cfg:
    """Host copy of the settings (three tiny D2H reads per camera; cache the result per camera in hot loops)."""
    bg = [float(v) for v in s.bg.detach().float().cpu().reshape(-1)]
    vm = [float(v) for v in s.viewmatrix.detach().float().cpu().reshape(-1)]
    pm = [float(v) for v in s.projmatrix.detach().float().cpu().reshape(-1)]
    cp = [float(v) for v in s.campos.detach().float().cpu().reshape(-1)]
    t0, t1 = tile_rows if tile_rows is not None else (0, 0)
    return L.nm_raster_cfg(int(s.image_height), int(s.image_width), float(s.tanfovx), float(s.tanfovy), (C.c_float * 3)(*bg),
                           float(s.scale_modifier), (C.c_float * 16)(*vm), (C.c_float * 16)(*pm), int(s.sh_degree),
                           (C.c_float * 3)(*cp), int(bool(s.prefiltered)), int(bool(s.debug)), int(t0), int(t1))


class RasterCamera(object):
    """Pre-marshalled camera: build once per (view, frame) and reuse, so the hot loop performs no host reads."""

    def __init__(self, settings: GaussianRasterizationSettings, tile_rows: Optional[Tuple[int, int]] = None, walk_store=None):
        self.settings = settings
        self.tile_rows = tile_rows
        self.cfg = _c_cfg(settings, tile_rows)
        self.bins = BinCapacity()
        # device -> per-tile walk record of the previous render (nm_raster_forward_ex).  walk_store: a dict shared by all the
        # RasterCameras of one viewpoint (its tile-row stripes on several GPUs render into the same full-image record)
        self._walk = walk_store if walk_store is not None else {}

    def tile_walk(self, device) -> Optional[Tensor]:
        """Per-tile walk record of this camera on `device` (int32 storage of the C ABI's uint32 array), created zeroed on
        first use.  NEUMA_RASTER_HINT=0 switches the hinted split compositing off (every render planned from scratch)."""
        if os.environ.get("NEUMA_RASTER_HINT", "1") == "0":
            return None
        key = torch.device(device)
        w = self._walk.get(key)
        if w is None:
            gx, gy = (self.cfg.image_width + 15) // 16, (self.cfg.image_height + 15) // 16
            w = self._walk[key] = torch.zeros(gx * gy, dtype=torch.int32, device=key)
        return w


def build_cov3D(scales: Tensor, rotations: Tensor, scale_modifier: float = 1.0) -> Tensor:
    """general_utils.py:93-139 + gaussian_model.py:27-31 in torch (differentiable; off the hot path:
    NeuMA always passes cov3D_precomp)."""
    q = rotations / rotations.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1)], -2)
    Lm = R * (scale_modifier * scales)[:, None, :]
    S = Lm @ Lm.transpose(-1, -2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


def _raster_inputs(means3D, shs, colors_precomp, opacities, cov3D):
    m3 = means3D.detach().float().contiguous()
    op = opacities.detach().float().contiguous()
    cv = cov3D.detach().float().contiguous()
    sh = None if shs is None else shs.detach().float().contiguous()
    cp = None if colors_precomp is None else colors_precomp.detach().float().contiguous()
    return m3, sh, cp, op, cv


class BinCapacity(object):
    """Capacity policy of one camera's rasterizer state: `cap` = (Gaussian, bin) pairs of the depth-sorted bin lists
    (nm_raster_forward's cap_pairs), `items` = work items of the split compositing (nm_raster_cfg.split_items).  The first
    render with a camera is checked synchronously (and repeated with a larger state buffer if it overflowed); afterwards the
    pair capacity is twice the last observed number of pairs and the status of every render is read back asynchronously.
    An overflow that slipped through (the scene suddenly needs > 2x the pairs) is never silent and never turns into garbage
    gradients: such a render composites the background only, its backward pass contributes exactly zero (device-side guard in
    k_render_bwd) and raises if the status has already arrived; the next forward with this camera, any later drain and
    `flush_pending()` (what evaluate / the trainers call at the end of a frame) raise."""

    def __init__(self):
        self.cap = 0              # 0: not sized yet (first guess 8 K + 4096)
        self.items = 0            # 0: the library's default (8192)
        self.verified = False     # a synchronously checked render has completed within `cap`

    def observe(self, pairs: int, items_wanted: int = 0):
        self.cap = max(self.cap, 2 * pairs + 4096)
        # work items: 1.5 x what the plan asked for, in steps of 1024 (too few only costs time: the other tiles go whole)
        want = min(32768, max(1024, (3 * items_wanted // 2 + 1023) // 1024 * 1024))
        if self.items == 0 or want > self.items or 2 * want < self.items:
            self.items = want


class _Pending(object):
    """Status words of renders that have been enqueued but not looked at yet (pinned memory + an event each)."""

    def __init__(self):
        self.entries = []

    def add(self, bins: BinCapacity, status: Tensor, ev) -> "list":
        entry = [bins, status, ev, False]
        self.entries.append(entry)
        return entry

    @staticmethod
    def examine(entry, wait: bool) -> bool:
        """True once the entry has been looked at; raises if that render overflowed its lists."""
        bins, status, ev, seen = entry
        if seen:
            return True
        if not wait and not ev.query():
            return False
        ev.synchronize()
        entry[3] = True
        pairs, overflow, items = int(status[0]), int(status[1]) & 0xFFFFFFFF, int(status[2])     # (high half: NM_RASTER_DEBUG)
        _STATUS_POOL.append((status, ev))          # pinned words + event go round (allocating them was ~20 us per render)
        entry[1] = entry[2] = None
        bins.observe(pairs, items)
        if overflow:
            raise L.NeumaHipError(f"rasterizer bin lists overflowed ({pairs} pairs > capacity): that render's image is incomplete; "
                                  "the capacity has been raised, repeat the step")
        return True

    def drain(self, wait: bool, bins: Optional[BinCapacity] = None):
        """Look at every entry that has finished (wait: at all of them; bins: and wait for that camera's)."""
        keep, err = [], None
        for e in self.entries:
            try:
                if not self.examine(e, wait or e[0] is bins):
                    keep.append(e)
            except L.NeumaHipError as ex:
                err = err or ex
        self.entries = keep
        if err is not None:
            raise err


_PENDING = _Pending()
_STATUS_POOL = []       # (pinned int64[3], event) pairs whose render has been looked at
# NEUMA_RASTER_BACKWARD_WAIT=1: the backward pass waits for its forward pass's status before it produces gradients (an
# overflow then raises before any gradient exists, at the price of a host stall per render)
_BACKWARD_WAITS = os.environ.get("NEUMA_RASTER_BACKWARD_WAIT", "0") == "1"


def flush_pending():
    """Wait for every render enqueued so far and raise if one of them overflowed its bin lists (its image is incomplete).
    Call it where an image leaves the GPU without a backward pass behind it: at the end of an evaluation frame / epoch."""
    _PENDING.drain(wait=True)


class RenderRecord(object):
    """What one render keeps for its backward pass (raster_forward_raw -> raster_backward_raw)."""
    __slots__ = ("cfg", "K", "M", "cap", "has_sh", "m3", "shcol", "op", "cv", "state", "pending")


_HINT_ENV_DONE = [False]


def _apply_hint_env(lib):
    """NEUMA_HINT_FWD_LEN / NEUMA_HINT_MINSEG: process-wide tuning of the hinted plan (nm_raster_set_hinted), read once."""
    if not _HINT_ENV_DONE[0]:
        _HINT_ENV_DONE[0] = True
        fl, ms = os.environ.get("NEUMA_HINT_FWD_LEN"), os.environ.get("NEUMA_HINT_MINSEG")
        if fl is not None or ms is not None:
            L.check(lib.nm_raster_set_hinted(int(fl) if fl is not None else 0, int(ms) if ms is not None else 256), "nm_raster_set_hinted")


def raster_forward_raw(cam: RasterCamera, m3: Tensor, sh: Optional[Tensor], cp: Optional[Tensor], op: Tensor, cv: Tensor):
    """GaussianRasterizer.forward on detached contiguous fp32 inputs, outside autograd: (color, radii, RenderRecord).  Shared by
    the autograd.Function below and by the frame driver's fused node (harness._FrameTail)."""
    lib = L.lib()
    _apply_hint_env(lib)
    dev = m3.device
    stream = L.stream_ptr(dev)
    K = m3.size(0)
    M = 0 if sh is None else sh.size(1)
    H, W = cam.cfg.image_height, cam.cfg.image_width
    bins = cam.bins
    # renders that have finished meanwhile, and this camera's previous one: sizes observed, overflows raised
    _PENDING.drain(wait=False, bins=bins)
    first = not bins.verified
    if bins.cap == 0:
        bins.cap = 8 * K + 4096
    radii = torch.empty(K, dtype=torch.int32, device=dev)            # k_preprocess writes every entry
    full = cam.cfg.tile_y1 <= cam.cfg.tile_y0 or (cam.cfg.tile_y0 == 0 and cam.cfg.tile_y1 * 16 >= H)
    # a full-image render writes every pixel; a stripe leaves the rows outside it untouched (zero)
    color = (torch.empty if full else torch.zeros)(3, H, W, dtype=torch.float32, device=dev)
    rec = RenderRecord()
    while True:
        cap = int(bins.cap)
        # this render's own copy of the settings (the item capacity may change later) and the buffer sizes that go with it:
        # kept per (K, cap, items) - a steady frame loop asks for the same ones every frame
        sized = cam.__dict__.get("_sized")
        if sized is None or sized[0] != (K, cap, int(bins.items)):
            cfg = L.nm_raster_cfg.from_buffer_copy(cam.cfg)
            cfg.split_items = int(bins.items)
            nstate, nscratch = C.c_size_t(0), C.c_size_t(0)
            L.check(lib.nm_raster_state_bytes_ex(C.byref(cfg), K, cap, C.byref(nstate), C.byref(nscratch)), "nm_raster_state_bytes_ex")
            sized = cam._sized = ((K, cap, int(bins.items)), cfg, int(nstate.value), int(nscratch.value))
        _, cfg, nstate, nscratch = sized
        # kept for the backward pass: records, lists, checkpoints.  The forward-only part (pair log, counters, per-segment
        # scratch: the larger half) goes back to the caching allocator when this function returns - stream-ordered, so
        # the next render on this stream reuses it
        state = torch.empty(nstate, dtype=torch.uint8, device=dev)
        scratch = torch.empty(nscratch, dtype=torch.uint8, device=dev)
        if _STATUS_POOL:
            status, ev = _STATUS_POOL.pop()
            status.zero_()
        else:
            status, ev = torch.zeros(3, dtype=torch.int64, pin_memory=True), torch.cuda.Event()
        L.check(lib.nm_raster_forward_ex(C.byref(cfg), K, M, L.ptr(m3), L.ptr(sh), L.ptr(cp), L.ptr(op), L.ptr(cv), L.ptr(radii),
                                         L.ptr(state), nstate, L.ptr(scratch), nscratch, cap, L.ptr(color),
                                         C.c_void_p(status.data_ptr()), L.ptr(cam.tile_walk(dev)), stream),
                "nm_raster_forward_ex")
        ev.record(torch.cuda.current_stream(dev))
        if not first:
            rec.pending = _PENDING.add(bins, status, ev)
            break
        ev.synchronize()                      # first render with this camera: size the lists from what the view needs
        bins.observe(int(status[0]), int(status[2]))
        if not (int(status[1]) & 0xFFFFFFFF):
            bins.verified = True
            rec.pending = None
            break
    rec.cfg, rec.K, rec.M, rec.cap, rec.has_sh = cfg, K, M, cap, sh is not None
    rec.m3, rec.shcol, rec.op, rec.cv, rec.state = m3, (sh if sh is not None else cp), op, cv, state
    return color, radii, rec


def raster_backward_raw(rec: RenderRecord, grad_color: Tensor, need_means2D=False, need_cov=False, need_opacity=False,
                        need_color=False):
    """GaussianRasterizer.backward for a RenderRecord: (dmeans3D, dmeans2D, dcov, dopacity, dsh, dcolors_precomp)."""
    lib = L.lib()
    dev = rec.m3.device
    if rec.pending is not None:
        # no gradients of an incomplete image: raises if the forward pass is known to have overflowed its lists.  If it has not
        # finished yet (a frame loop whose host runs ahead of the device) nothing waits here: k_render_bwd reads the overflow
        # flag on the device and contributes zero gradient in that case, and the status stays registered - the next render with
        # this camera, any later drain, and flush_pending() raise
        _Pending.examine(rec.pending, wait=_BACKWARD_WAITS)
    K, M = rec.K, rec.M
    g = grad_color.float().contiguous()
    dmeans3D = torch.empty(K, 3, dtype=torch.float32, device=dev)
    dmeans2D = torch.empty(K, 3, dtype=torch.float32, device=dev) if need_means2D else None
    dcov = torch.empty(K, 6, dtype=torch.float32, device=dev) if need_cov else None
    dop = torch.empty(K, 1, dtype=torch.float32, device=dev) if need_opacity else None
    dsh = torch.empty(K, M, 3, dtype=torch.float32, device=dev) if (rec.has_sh and need_color) else None
    dcol = torch.empty(K, 3, dtype=torch.float32, device=dev) if ((not rec.has_sh) and need_color) else None
    ws_bytes = int(lib.nm_raster_bwd_workspace(K))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    sh = rec.shcol if rec.has_sh else None
    cp = None if rec.has_sh else rec.shcol
    L.check(lib.nm_raster_backward(C.byref(rec.cfg), K, M, L.ptr(rec.m3), L.ptr(sh), L.ptr(cp), L.ptr(rec.op), L.ptr(rec.cv),
                                   L.ptr(rec.state), rec.cap, L.ptr(g), L.ptr(dmeans3D), L.ptr(dmeans2D), L.ptr(dcov), L.ptr(dop),
                                   L.ptr(dsh), L.ptr(dcol), L.ptr(ws), ws_bytes, L.stream_ptr(dev)), "nm_raster_backward")
    return dmeans3D, dmeans2D, dcov, dop, dsh, dcol


class _RasterizeGaussians(autograd.Function):

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, cov3D, cam: RasterCamera):
        m3, sh, cp, op, cv = _raster_inputs(means3D, shs, colors_precomp, opacities, cov3D)
        color, radii, rec = raster_forward_raw(cam, m3, sh, cp, op, cv)
        state = rec.state
        rec.m3 = rec.shcol = rec.op = rec.cv = rec.state = None          # (the tensors travel through save_for_backward)
        ctx.rec = rec
        ctx.save_for_backward(m3, sh if sh is not None else cp, op, cv, state)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _grad_radii):
        rec = ctx.rec
        rec.m3, rec.shcol, rec.op, rec.cv, rec.state = ctx.saved_tensors
        need = ctx.needs_input_grad  # means3D, means2D, shs, colors, opac, cov3D, cam
        dmeans3D, dmeans2D, dcov, dop, dsh, dcol = raster_backward_raw(
            rec, grad_color, need_means2D=need[1], need_cov=need[5], need_opacity=need[4],
            need_color=(need[2] if rec.has_sh else need[3]))
        rec.m3 = rec.shcol = rec.op = rec.cv = rec.state = None
        return dmeans3D, dmeans2D, dsh, dcol, dop, dcov, None


def count_tile_pairs(rasterizer, means3D, opacities, shs=None, colors_precomp=None, cov3D_precomp=None) -> int:
    """Exact number of (Gaussian, 16x16 tile) pairs of a view - the reference extension's `num_rendered` (statistics)."""
    lib = L.lib()
    cam = rasterizer._cam
    m3, sh, cp, op, cv = _raster_inputs(means3D, shs, colors_precomp, opacities, cov3D_precomp)
    dev, K = m3.device, m3.size(0)
    M = 0 if sh is None else sh.size(1)
    cap = 8 * K + 4096
    state_bytes = int(lib.nm_raster_state_bytes(C.byref(cam.cfg), K, cap))
    state = torch.empty(state_bytes, dtype=torch.uint8, device=dev)
    radii = torch.empty(K, dtype=torch.int32, device=dev)
    color = torch.empty(3, cam.cfg.image_height, cam.cfg.image_width, dtype=torch.float32, device=dev)
    L.check(lib.nm_raster_forward(C.byref(cam.cfg), K, M, L.ptr(m3), L.ptr(sh), L.ptr(cp), L.ptr(op), L.ptr(cv), L.ptr(radii),
                                  L.ptr(state), state_bytes, cap, L.ptr(color), None, L.stream_ptr(dev)), "nm_raster_forward")
    out = C.c_int64(0)
    L.check(lib.nm_raster_count_pairs(C.byref(cam.cfg), K, L.ptr(state), cap, C.byref(out), L.stream_ptr(dev)), "nm_raster_count_pairs")
    return int(out.value)


def view_stats(rasterizer, means3D, opacities, shs=None, colors_precomp=None, cov3D_precomp=None) -> dict:
    """Statistics of one view for the byte accounting (bench.py): pairs = exact (Gaussian, tile) pairs (count_tile_pairs),
    list_entries = sum over the tiles of the length of their bin's depth-sorted list (what a tile-by-tile walk of the whole
    lists would examine), walked = the same up to where the tiles' pixels saturate (the camera's walk record of its last
    render; 0 if it has none)."""
    lib = L.lib()
    cam = rasterizer._cam
    m3, sh, cp, op, cv = _raster_inputs(means3D, shs, colors_precomp, opacities, cov3D_precomp)
    dev, K = m3.device, m3.size(0)
    M = 0 if sh is None else sh.size(1)
    cap = 8 * K + 4096
    state_bytes = int(lib.nm_raster_state_bytes(C.byref(cam.cfg), K, cap))
    state = torch.empty(state_bytes, dtype=torch.uint8, device=dev)
    radii = torch.empty(K, dtype=torch.int32, device=dev)
    color = torch.empty(3, cam.cfg.image_height, cam.cfg.image_width, dtype=torch.float32, device=dev)
    L.check(lib.nm_raster_forward(C.byref(cam.cfg), K, M, L.ptr(m3), L.ptr(sh), L.ptr(cp), L.ptr(op), L.ptr(cv), L.ptr(radii),
                                  L.ptr(state), state_bytes, cap, L.ptr(color), None, L.stream_ptr(dev)), "nm_raster_forward")
    out = C.c_int64(0)
    L.check(lib.nm_raster_count_pairs(C.byref(cam.cfg), K, L.ptr(state), cap, C.byref(out), L.stream_ptr(dev)), "nm_raster_count_pairs")
    hdr = state[:64].view(torch.int32).cpu().numpy().astype("int64") & 0xFFFFFFFF
    walk = cam._walk.get(torch.device(dev))
    return {"pairs": int(out.value), "list_entries": int(hdr[14] + (hdr[15] << 32)),
            "walked": int(walk.long().sum()) if walk is not None else 0}


def split_plan(rasterizer, means3D, opacities, shs=None, colors_precomp=None, cov3D_precomp=None, hinted: bool = False) -> Tuple[int, int]:
    """(work items, segment length) the split compositing chose for this view (nm_raster_set_split; 0 work items = every
    tile composited by one workgroup).  hinted: plan from a copy of the camera's walk record, as its next render will
    (nm_raster_forward_ex).  Diagnostics: runs a forward pass into a scratch state and reads its header."""
    lib = L.lib()
    cam = rasterizer._cam
    m3, sh, cp, op, cv = _raster_inputs(means3D, shs, colors_precomp, opacities, cov3D_precomp)
    dev, K = m3.device, m3.size(0)
    M = 0 if sh is None else sh.size(1)
    cap = 8 * K + 4096
    state_bytes = int(lib.nm_raster_state_bytes(C.byref(cam.cfg), K, cap))
    state = torch.empty(state_bytes, dtype=torch.uint8, device=dev)
    radii = torch.empty(K, dtype=torch.int32, device=dev)
    color = torch.empty(3, cam.cfg.image_height, cam.cfg.image_width, dtype=torch.float32, device=dev)
    walk = cam.tile_walk(dev) if hinted else None
    walk = None if walk is None else walk.clone()
    L.check(lib.nm_raster_forward_ex(C.byref(cam.cfg), K, M, L.ptr(m3), L.ptr(sh), L.ptr(cp), L.ptr(op), L.ptr(cv), L.ptr(radii),
                                     L.ptr(state), state_bytes, None, 0, cap, L.ptr(color), None, L.ptr(walk), L.stream_ptr(dev)),
            "nm_raster_forward_ex")
    hdr = state[:64].view(torch.int32).cpu()
    return int(hdr[8]), int(hdr[9])


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, tile_rows: Optional[Tuple[int, int]] = None, walk_store=None):
        super().__init__()
        self.raster_settings = raster_settings
        self._cam = (RasterCamera(raster_settings, tile_rows, walk_store) if not isinstance(raster_settings, RasterCamera)
                     else raster_settings)

    def markVisible(self, positions: Tensor) -> Tensor:
        s = self._cam.settings
        hom = torch.cat([positions, torch.ones_like(positions[:, :1])], 1)
        return (hom @ s.viewmatrix.to(positions))[:, 2] > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        if cov3D_precomp is None:
            cov3D_precomp = build_cov3D(scales, rotations, self._cam.settings.scale_modifier)
        if means2D is None:
            means2D = torch.zeros_like(means3D)
        return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, cov3D_precomp, self._cam)


def get_rasterizer(viewpoint_camera, active_sh_degree: int, debug, bg_color: Tensor, scaling_modifier=1.0,
                   tile_rows: Optional[Tuple[int, int]] = None) -> GaussianRasterizer:
    """gaussian_renderer/__init__.py:92-119 (viewpoint_camera: anything with FoVx, FoVy, image_height, image_width,
    world_view_transform, full_proj_transform, camera_center — Camera / PhysCamera / MiniCam of cameras.py)."""
    cache = getattr(viewpoint_camera, "_nm_raster_cache", None)
    # the marshalled camera is reused only while nothing it was built from has changed: tensors are identified by storage
    # AND version counter, so an in-place update of a transform / the background (viewer-style camera reuse) rebuilds it
    tensors = (viewpoint_camera.world_view_transform, viewpoint_camera.full_proj_transform, viewpoint_camera.camera_center, bg_color)
    key = (int(active_sh_degree), float(scaling_modifier), tile_rows, bool(debug), float(viewpoint_camera.FoVx),
           float(viewpoint_camera.FoVy), int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)) + \
        tuple((t.data_ptr(), t._version) for t in tensors)
    # (key, rasterizer) of the last use first; up to 8 stripe / setting variants of a viewpoint stay marshalled (a stripe plan
    # that is re-balanced now and then must not pay the first-render synchronisation again when it returns to an old cut)
    entries = cache if isinstance(cache, list) else ([cache] if cache else [])
    for i, (k, r) in enumerate(entries):
        if k == key:
            if i:
                entries.insert(0, entries.pop(i))
                viewpoint_camera._nm_raster_cache = entries
            return r
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=debug,
    )
    walk_key = key[:2] + key[4:]              # everything but the stripe and the debug flag: one walk record per viewpoint
    stores = getattr(viewpoint_camera, "_nm_walk_stores", None)
    if stores is None or stores[0] != walk_key:
        stores = (walk_key, {})
    rast = GaussianRasterizer(raster_settings=raster_settings, tile_rows=tile_rows, walk_store=stores[1])
    try:
        viewpoint_camera._nm_walk_stores = stores
        viewpoint_camera._nm_raster_cache = [(key, rast)] + entries[:7]
    except Exception:
        pass
    return rast


def deform_cov_by_F(cov6: Tensor, F: Tensor) -> Tensor:
    """simulation_utils.py:25-48: Sigma' = F Sigma F^T on packed (K,6); not differentiable, as in the reference
    (tune/utils.py:365-373 launches it outside any tape)."""
    c = cov6.detach().float().contiguous()
    Fc = F.detach().float().reshape(-1, 3, 3).contiguous()
    out = torch.empty_like(c)
    L.check(L.lib().nm_cov_deform(c.size(0), L.ptr(c), L.ptr(Fc), L.ptr(out), L.stream_ptr(c.device)), "nm_cov_deform")
    return out
