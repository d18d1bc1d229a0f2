"""Diffusion schedule utilities — host-side mirror of ``avdiff/utils/schedule_utils.py`` over the HIP C ABI.

Schedule tables are a once-per-run host computation: they are built on the CPU with the same
fp32 operation sequence as the reference so the tables are bit-identical, then uploaded.  The per-step ops
(``timestep_embedding``, ``ddim_step``) run as HIP kernels.  ``q_sample`` is training-only and not provided.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

from . import functional as Fn

_KINDS = ("linear", "sigmoid", "cosine")


def make_beta_schedule(steps: int, kind: str = "cosine", min_beta: float = 1e-4, max_beta: float = 2e-2) -> torch.Tensor:
    """betas[t], t = 0..steps-1 (schedule_utils.py:14-49).  fp32 CPU tensor."""
    kind = kind.lower()
    if kind not in _KINDS:
        raise ValueError(f"Unknown schedule kind: {kind}")
    f32 = torch.float32
    if kind == "cosine":
        u = torch.linspace(0, steps, steps + 1, dtype=f32) / steps
        curve = torch.cos((u + 0.008) / (1 + 0.008) * math.pi / 2) ** 2
        curve = curve / curve[0]
        betas = 1 - curve[1:] / curve[:-1]
    elif kind == "linear":
        betas = torch.linspace(min_beta, max_beta, steps, dtype=f32)
    else:
        betas = min_beta + (max_beta - min_beta) * torch.sigmoid(torch.linspace(-6, 6, steps, dtype=f32))
    return betas.clamp(1e-8, 0.999)


def alphas_cumprod_from_betas(betas: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(alphas, alpha_bar) — schedule_utils.py:52-57."""
    alphas = 1.0 - betas.to(torch.float32)
    return alphas, alphas.cumprod(0)


def make_sampling_schedule(T_train: int, T_sample: int) -> torch.Tensor:
    """T_sample+1 decreasing int64 timesteps from T_train-1 to -1 (schedule_utils.py:132-143)."""
    return torch.linspace(T_train - 1, -1, T_sample + 1).round().to(torch.long)


def timestep_embedding(timesteps: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """[cos | sin] sinusoidal embedding [B, dim] on the device (schedule_utils.py:64-86)."""
    return Fn.timestep_embedding(timesteps, dim, max_period)


def ddim_step(x_t: torch.Tensor, t_now: torch.Tensor, t_prev: torch.Tensor, eps_hat: torch.Tensor,
              alpha_bar: torch.Tensor, eta: float = 0.0, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One DDIM update x_{t_prev} from x_t (schedule_utils.py:146-200); t_prev = -1 means alpha_bar = 1.

    ``alpha_bar`` may live on the CPU (as it does in the reference sampler, sample_clip.py:274-275): it is
    uploaded.  ``noise`` (extension) lets the caller supply z for eta > 0; otherwise it is drawn like the reference.
    """
    return Fn.ddim_step(x_t, t_now, t_prev, eps_hat, alpha_bar, eta, noise)
