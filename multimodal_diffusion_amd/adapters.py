"""Timestep embedder — mirror of ``avdiff/models/adapters.py:130-158`` (``TimestepEmbedder``), SURVEY row a10.

The sampler never instantiates it (it concatenates the parameter-free sinusoid, sample_clip.py:59-70); it is here
because it is the reference's only anchor for the "timestep-embedding MLP" of the task description:
sinusoid(dim) → Linear(dim, 2·dim) → SiLU → Linear(2·dim, dim), ``state_dict`` keys ``mlp.0.*`` / ``mlp.2.*``.
The other classes of that file (positional / modality embeddings) are never used by sampler or trainer.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn


@dataclass
class TimestepCfg:
    dim: int = 256
    mode: str = "sin"      # "sin" | "mlp"


class TimestepEmbedder(nn.Module):
    def __init__(self, cfg: TimestepCfg):
        super().__init__()
        self.cfg = cfg
        if cfg.mode == "mlp":
            # containers only (torch default Linear init, as the reference); SiLU slot kept for key parity
            self.mlp = nn.Sequential(nn.Linear(cfg.dim, cfg.dim * 2), nn.SiLU(), nn.Linear(cfg.dim * 2, cfg.dim))
        elif cfg.mode == "sin":
            self.mlp = None
        else:
            raise ValueError(f"unknown TimestepEmbedder mode: {cfg.mode}")

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        base = Fn.timestep_embedding(t, self.cfg.dim)
        if self.mlp is None:
            return base
        h = Fn.linear(base, self.mlp[0].weight, self.mlp[0].bias, act=L.ACT_SILU)
        return Fn.linear(h, self.mlp[2].weight, self.mlp[2].bias)
