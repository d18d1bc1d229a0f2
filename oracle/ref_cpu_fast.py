"""CPU baseline port that runs at the reference's own CPU speed — TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

``oracle/ref_cpu.py`` is written for clarity and dtype-genericity (explicit q k^T / softmax / p v, ``x @ w.T + b``, a GELU
spelled out with ``erf``); timed on the same host it is 1.7x slower than the reference's ``MMDiT.forward`` because the reference's
``nn.Module`` calls land on fused ATen kernels.  This file restates the SAME functions on those fused kernels — the ones the
reference itself dispatches to in ``eval()`` under ``no_grad`` — so that ``bench.py``'s ``cpu_baseline`` is the reference's CPU path
at the reference's CPU speed:

  nn.Linear (mmdt.py:77-83, noise_heads.py:141-147, sample_clip.py:54-56) .......... ``F.linear`` (one addmm, bias fused)
  nn.MultiheadAttention, batch_first, need_weights=False, eval (mmdt.py:51-61) ..... packed in_proj ``F.linear`` +
                                                                                    ``F.scaled_dot_product_attention`` + out_proj
  F.gelu (mmdt.py:79) / nn.GELU (noise_heads.py:28-36) ............................. ``F.gelu`` (erf form)
  nn.LayerNorm (noise_heads.py:141-147) ............................................ ``F.layer_norm``
  RMSNorm (mmdt.py:39-42) .......................................................... ``x.norm(dim=-1)`` exactly as written there

Pinned: ``tests/test_oracle_golden.py::test_fast_port_matches_oracle`` holds every function here to 1e-5 of ``ref_cpu`` (itself
pinned by the golden fixtures), and ``tools/cpu_port_speed.py`` times it beside the imported reference modules in the build
container (``profiles/r03_cpu_port_speed.json``).  fp32 only: it is the timed baseline, not the adjudicator.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn.functional as F

from . import ref_cpu as R

Tensor = torch.Tensor
Weights = R.Weights


def rmsnorm(x: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    # mmdt.py:39-42, operation for operation: norm over the last dim, / sqrt(d), eps outside the sqrt
    norm_x = x.norm(dim=-1, keepdim=True) / math.sqrt(x.shape[-1])
    return scale * x / (norm_x + eps)


def self_attention(x: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor, n_heads: int) -> Tensor:
    """nn.MultiheadAttention(batch_first=True)(x, x, x, need_weights=False) in eval mode, no masks (mmdt.py:51-61)."""
    B, N, d = x.shape
    qkv = F.linear(x, w_in, b_in).view(B, N, 3, n_heads, d // n_heads)
    q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)                         # [B,H,N,dh] each
    o = F.scaled_dot_product_attention(q, k, v)                             # scale 1/sqrt(dh), no dropout
    return F.linear(o.transpose(1, 2).reshape(B, N, d), w_out, b_out)


def mmdit_forward(x: Tensor, W: Weights, n_layers: int, n_heads: int) -> Tensor:
    for i in range(n_layers):
        p = f"blocks.{i}."
        x = x + self_attention(rmsnorm(x, W[p + "norm1.scale"]), W[p + "attn.mha.in_proj_weight"], W[p + "attn.mha.in_proj_bias"],
                               W[p + "attn.mha.out_proj.weight"], W[p + "attn.mha.out_proj.bias"], n_heads)
        h = F.gelu(F.linear(rmsnorm(x, W[p + "norm2.scale"]), W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])
    return rmsnorm(x, W["final_norm.scale"])


def noise_head(h: Tensor, W: Weights, modality: str, n_shared: int = 2) -> Tensor:
    """MultiModalNoiseHead for one modality, GELU trunk (noise_heads.py:185-229)."""
    y = F.linear(h, W[f"input_proj.{modality}.weight"], W[f"input_proj.{modality}.bias"])
    for j in range(n_shared):
        y = F.linear(y, W[f"shared.{j}.0.weight"], W[f"shared.{j}.0.bias"])
        y = F.gelu(F.layer_norm(y, (y.shape[-1],), W[f"shared.{j}.1.weight"], W[f"shared.{j}.1.bias"], 1e-5))
    return F.linear(y, W[f"out_proj.{modality}.weight"], W[f"out_proj.{modality}.bias"])


def eps_pair(Xt: Tensor, Xp: Tensor, core: Weights, head: Weights, n_layers: int, n_heads: int) -> Tuple[Tensor, Tensor]:
    """cond / null ε̂ of the video target, sequence [video ; audio]; the reference runs two forwards (sample_clip.py:374,378)."""
    nt = Xt.shape[1]

    def run(prompt_rows: Tensor) -> Tensor:
        hfull = mmdit_forward(torch.cat([Xt, prompt_rows], 1), core, n_layers, n_heads)
        return noise_head(hfull[:, :nt], head, "video")

    return run(Xp), run(torch.zeros_like(Xp))


def denoise_step_a2v(z_v: Tensor, z_a0: Tensor, t_now: Tensor, t_prev: Tensor, alpha_bar: Tensor, *,
                     adapt_v: Weights, adapt_a: Weights, core: Weights, head: Weights, n_layers: int, n_heads: int,
                     tdim: int = 256, tube=(2, 4, 4), chunk=(4, 4), guidance: float = 3.5) -> Tensor:
    """One audio->video CFG step (sample_clip.py:359-389), eta = 0, concat timestep embedding — the bench workload."""
    B, C, T, H, Wd = z_v.shape

    def embed(tok, ad, t):
        x = F.linear(tok, ad["proj.weight"], ad["proj.bias"])
        e = R.timestep_embedding(t, tdim, dtype=x.dtype)[:, None, :].expand(-1, x.shape[1], -1)
        return torch.cat([x, e], dim=-1)

    Xv = embed(R.tube_patch(z_v, *tube), adapt_v, t_now)
    Xa = embed(R.audio_tokens(z_a0, *chunk), adapt_a, torch.zeros_like(t_now))
    e_c, e_n = eps_pair(Xv, Xa, core, head, n_layers, n_heads)
    eps_lat = R.tube_unpatch(e_n + guidance * (e_c - e_n), C, T, H, Wd, *tube)
    return R.ddim_update(z_v, t_now, t_prev, eps_lat, alpha_bar, 0.0)
