"""Camera / video datasets of a NeuMA experiment.  Mirrors /root/reference/modules/tune/dataset/neuma_dataset.py:
CameraDataset 16-74 (cameras[view][step], views, steps, getCameras) and VideoDataset 76-153 (initial particle state,
the global initial velocity as a Parameter with its optimiser / scheduler, init.pt export).  Readers: neuma_amd.io
(NeuMASynthetic `data_dynamic.json`, RealCapture `cameras_calib.json` + COLMAP intrinsics)."""
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import io as nio

_READERS = {"NeuMASynthetic": nio.read_neuma_synthetic_cameras, "RealCapture": nio.read_realcapture_cameras}


def _get(cfg, key, default=None):
    return cfg.get(key, default) if hasattr(cfg, "get") else getattr(cfg, key, default)


class CameraDataset(object):
    def __init__(self, cfg, readCameras: bool = True):
        self.eval = bool(_get(cfg, "eval", False))
        self.cameras = {}
        if readCameras:
            self.readCameras(cfg)

    def readCameras(self, cfg) -> None:
        camera_type = _get(cfg, "camera_type")
        if camera_type not in _READERS:
            raise ValueError(f"unknown camera_type {camera_type!r} (NeuMASynthetic | RealCapture)")
        data = dict(_get(cfg, "data"))
        if camera_type != "RealCapture":
            data.pop("read_mask_only", None)
        info = _READERS[camera_type](load_images=True, **data)
        self.views, self.steps = info["views"], info["steps"]  # both sorted
        self.length = len(self.views) * len(self.steps)
        cam_cfg = _get(cfg, "camera") or {}
        device = _get(cam_cfg, "data_device") or _get(cfg, "device") or "cpu"
        self.cameras = {}
        for ci in info["cam_infos"]:
            self.cameras.setdefault(ci.view, {})[ci.step] = nio.DiskCamera(ci, device=device)

    def getCameras(self, view, step):
        if isinstance(view, int):
            view = self.views[view]
        elif not isinstance(view, str):
            raise ValueError(f"view must be an integer or a string, but got {view} ({type(view)})")
        return self.cameras[view][step]

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        idx = idx % self.length
        return self.cameras[self.views[idx // len(self.steps)]][self.steps[idx % len(self.steps)]]


class VideoDataset(CameraDataset):
    def __init__(self, cfg, readCameras: bool = True):
        super().__init__(cfg, readCameras=False)
        self.device = _get(cfg, "device") or "cpu"
        self._init_x = None
        self._init_v = None
        self._velocity_opt = None
        self._velocity_sch = None
        if readCameras:
            self.readCameras(cfg)

    def get_init_material_data(self):
        n, dev = self._init_x.shape[0], self._init_x.device
        init_C = torch.zeros(n, 3, 3, device=dev)
        init_F = torch.eye(3, device=dev).unsqueeze(0).expand(n, 3, 3)
        init_S = torch.zeros(n, 3, 3, device=dev)
        return self.get_init_x, self.get_init_v, init_C, init_F, init_S

    @property
    def getVelocityOptimizer(self):
        return self._velocity_opt

    @property
    def getVelocityScheduler(self):
        return self._velocity_sch

    @property
    def get_init_x(self):
        return self._init_x

    @property
    def get_init_v(self):
        if self._init_v.ndim == 1:                                 # one global velocity, broadcast to every particle
            return self._init_v.unsqueeze(0).expand(self._init_x.shape[0], -1)
        return self._init_v

    def export_init_x_and_v(self, path) -> None:
        torch.save({"init_x": self.get_init_x.detach().cpu(), "init_v": self.get_init_v.detach().cpu().contiguous()}, path)

    def set_init_x_and_v(self, init_x, init_v=None) -> None:
        self._init_x = torch.as_tensor(np.asarray(init_x) if not isinstance(init_x, torch.Tensor) else init_x).to(self.device).float()
        if init_v is None:
            self._init_v = nn.Parameter(torch.zeros(3, device=self.device), requires_grad=True)
        else:
            t = torch.as_tensor(np.asarray(init_v) if not isinstance(init_v, torch.Tensor) else init_v)
            self._init_v = t.detach().to(self.device).float()

    def init_velocity_optimizer(self, optimizer, lr: float) -> None:
        self._velocity_opt = optimizer([self._init_v], lr=lr)

    def init_velocity_scheduler(self, scheduler_config, init_lr: float) -> None:
        from .train import fetch_scheduler
        self._velocity_sch = fetch_scheduler(scheduler_config).get_scheduler(self._velocity_opt, init_lr)

    def free_velocity_optimizer(self) -> None:
        self._velocity_opt = None

    def free_velocity_scheduler(self) -> None:
        self._velocity_sch = None

    def freeze_velocity(self) -> None:
        self._init_v.requires_grad = False
