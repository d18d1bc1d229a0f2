"""Token layout ops — host-side mirror of ``avdiff/utils/ops.py`` over the HIP C ABI.

``tube_patch_video`` / ``tube_unpatch_video`` (ops.py:100-144) and ``overlap_add_1d`` (ops.py:48-93) are HIP
kernels; ``chunk_1d`` (ops.py:17-45) is a zero-copy strided view in the reference (``Tensor.unfold``) and is one
here too.  ``pad_to_multiple`` is unused by the sampler and not provided.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import functional as Fn


def tube_patch_video(z: torch.Tensor, t: int, h: int, w: int) -> torch.Tensor:
    """[B,C,T,H,W] -> [B,N,C*t*h*w]; token order (T',H',W'), feature order (C,t,h,w)."""
    return Fn.tube_patch(z, t, h, w)


def tube_unpatch_video(tokens: torch.Tensor, C: int, T: int, H: int, W: int, t: int, h: int, w: int) -> torch.Tensor:
    return Fn.tube_unpatch(tokens, C, T, H, W, t, h, w)


def chunk_1d(x: torch.Tensor, length: int, stride: int, dim: int = -1) -> torch.Tensor:
    """[..., L] -> view [..., N, length], N = (L-length)//stride + 1 (no data movement)."""
    if dim not in (-1, x.dim() - 1):
        raise NotImplementedError("chunk_1d: only the last dimension is chunked on the sampler path")
    L_ = x.size(-1)
    if length <= 0 or stride <= 0 or L_ < length:
        return x[..., :max(0, min(L_, length))].unsqueeze(-2)
    return x.unfold(-1, length, stride)


def overlap_add_1d(windows: torch.Tensor, stride: int, length: Optional[int] = None, dim_windows: int = -2,
                   apply_hann: bool = False) -> torch.Tensor:
    """[..., N, W] -> [..., (N-1)*stride + W]: overlap-add divided by the summed window weights (rectangular window, or
    torch.hann_window(W) with apply_hann as ops.py:76-93)."""
    if dim_windows not in (-2, windows.dim() - 2):
        raise NotImplementedError("overlap_add_1d: windows must be indexed by dim -2")
    W = windows.size(-1)
    if length is not None and length != W:
        raise ValueError("length must equal the window size")
    prefix, N = windows.shape[:-2], windows.size(-2)
    L_out = (N - 1) * stride + W
    flat = windows.reshape(-1, N, W)                 # one 'channel' per prefix row
    win = torch.hann_window(W, dtype=torch.float32).to(windows.device) if apply_hann else None      # host-built table, as the reference
    y = Fn.audio_untokens(flat, 1, W, L_out, stride, window=win)  # [P,1,L]
    return y.view(*prefix, L_out)
