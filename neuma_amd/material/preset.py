"""ComposeMaterial — per-object split / apply / concatenate.
Mirrors /root/reference/modules/nclaw/material/preset.py:12-27 (the classical presets of that file are never
instantiated by any NeuMA driver and are out of scope)."""
from typing import Sequence

import torch
import torch.nn as nn
from torch import Tensor


class ComposeMaterial(nn.Module):
    def __init__(self, materials, sections: Sequence[int]) -> None:
        super().__init__()
        self.dim = 3
        self.materials = nn.ModuleList(materials)
        self.sections = sections

    def update_sections(self, sections: Sequence[int]) -> None:
        self.sections = sections

    def forward(self, F: Tensor) -> Tensor:
        outs = []
        for m, f in zip(self.materials, torch.split(F, list(self.sections), dim=0)):
            if f.numel() == 0:
                continue
            outs.append(m(f))
        return torch.cat(outs, dim=0)
