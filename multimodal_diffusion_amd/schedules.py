"""Per-modality schedule object — mirror of ``avdiff/models/schedules.py:27-109`` (``q_sample`` is training-only)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import torch

from . import schedule_utils as su


@dataclass
class ModalitySchedule:
    kind: str
    steps: int
    betas: torch.Tensor
    alphas: torch.Tensor
    alphas_cumprod: torch.Tensor

    @classmethod
    def make(cls, *, kind: str = "cosine", steps: int = 1000, min_beta: float = 1e-4,
             max_beta: float = 2e-2) -> "ModalitySchedule":
        betas = su.make_beta_schedule(steps=steps, kind=kind, min_beta=min_beta, max_beta=max_beta)
        alphas, abar = su.alphas_cumprod_from_betas(betas)
        return cls(kind=kind, steps=int(steps), betas=betas, alphas=alphas, alphas_cumprod=abar)

    def to(self, device) -> "ModalitySchedule":
        self.betas = self.betas.to(device)
        self.alphas = self.alphas.to(device)
        self.alphas_cumprod = self.alphas_cumprod.to(device)
        return self

    def ddim_step(self, z_t, t, t_prev, eps_hat, eta: float = 0.0):
        return su.ddim_step(z_t, t, t_prev, eps_hat, self.alphas_cumprod, eta=eta)

    def make_sampling_schedule(self, steps_sample: int) -> torch.Tensor:
        return su.make_sampling_schedule(self.steps, steps_sample)

    def timestep_embedding(self, t, dim: int, max_period: int = 10_000):
        return su.timestep_embedding(t, dim=dim, max_period=max_period)


def build_schedules_from_config(cfg: Dict) -> Dict[str, ModalitySchedule]:
    out = {}
    for m in ("video", "audio"):
        c = cfg["diffusion"][m]
        out[m] = ModalitySchedule.make(kind=c.get("schedule", "cosine"), steps=int(c.get("steps", 1000)),
                                       min_beta=float(c.get("min_beta", 1e-4)), max_beta=float(c.get("max_beta", 2e-2)))
    return out
