"""Tensor-level wrappers over the C ABI (one function per ``avd_*`` kernel entry point).

Each wrapper validates shapes the way the reference op it stands under does (AssertionError / ValueError),
allocates the output with torch (plumbing) and enqueues the HIP kernel on torch's current stream.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _lib as L

Tensor = torch.Tensor


def _st(t: Tensor) -> int:
    return L.stream_ptr(t.device)


def rmsnorm(x: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    """RMSNorm with eps outside the sqrt — avdiff/models/mmdt.py:39-42."""
    x = L.dev_f32(x, "x")
    scale = L.dev_f32(scale, "scale")
    d = x.shape[-1]
    y = torch.empty_like(x)
    L.check(L.lib().avd_rmsnorm_f32(x.data_ptr(), scale.data_ptr(), y.data_ptr(), x.numel() // d, d, eps, _st(x)))
    return y


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, act: int = L.ACT_NONE,
           residual: Optional[Tensor] = None) -> Tensor:
    """act(x @ weight.T + bias) + residual on the fp32 matrix cores (torch.nn.Linear semantics)."""
    x = L.dev_f32(x, "x")
    weight = L.dev_f32(weight, "weight")
    n, k = weight.shape
    if x.shape[-1] != k:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} x {k}->{n})")
    m = x.numel() // k
    out = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    b = None if bias is None else L.dev_f32(bias, "bias")
    r = None
    if residual is not None:
        r = L.dev_f32(residual, "residual")
        if r.shape != out.shape:
            raise RuntimeError("residual shape mismatch")
    L.check(L.lib().avd_gemm_bias_act_f32(x.data_ptr(), k, weight.data_ptr(), L.ptr(b), L.ptr(r), n, out.data_ptr(), n,
                                          m, n, k, act, _st(x)))
    return out


def linear_rmsfold(x: Tensor, weight_n: Tensor, bias: Optional[Tensor] = None, act: int = L.ACT_NONE,
                   residual: Optional[Tensor] = None, ss_in: Optional[Tensor] = None, eps: float = 1e-6, want_ss: bool = False):
    """The folded-RMSNorm Linear of the MMDiT composite (avd_gemm_rmsfold_f32).  ``weight_n`` = weight * norm_scale[None, :]
    when ``ss_in`` (the per-row sums of squares of x, [M, cols]) is given.  Returns (y, ss_out or None)."""
    x = L.dev_f32(x, "x")
    weight_n = L.dev_f32(weight_n, "weight")
    n, k = weight_n.shape
    m = x.numel() // k
    y = torch.empty(*x.shape[:-1], n, device=x.device, dtype=torch.float32)
    ss_out = torch.empty(m, n // 32, device=x.device, dtype=torch.float32) if want_ss else None
    cols = 0 if ss_in is None else ss_in.shape[-1]
    L.check(L.lib().avd_gemm_rmsfold_f32(x.data_ptr(), weight_n.data_ptr(), L.ptr(bias), L.ptr(residual), y.data_ptr(), m, n, k,
                                         act, L.ptr(ss_in), cols, eps, L.ptr(ss_out), _st(x)))
    return y, ss_out


def key_padding_bytes(mask: Optional[Tensor], B: int, N: int, device: torch.device) -> Optional[Tensor]:
    """nn.MultiheadAttention's key_padding_mask ([B,N], True / non-zero = ignore this key) as the uint8 table the kernels read.
    Float masks (additive) are not supported: the reference documents the boolean form (mmdt.py:120-123)."""
    if mask is None:
        return None
    if mask.is_floating_point():
        raise NotImplementedError("key_padding_mask must be boolean / integer (True = padding), as MMDiT.forward documents")
    if tuple(mask.shape) != (B, N):
        raise RuntimeError(f"key_padding_mask shape {tuple(mask.shape)} != {(B, N)}")
    if not mask.is_cuda:
        raise L.AvdError("key_padding_mask must be on the ROCm device (no CPU fallback)")
    return (mask != 0).to(torch.uint8).contiguous()


def attention(qkv: Tensor, n_heads: int, n_query: Optional[int] = None, key_padding_mask: Optional[Tensor] = None) -> Tensor:
    """softmax(q k^T / sqrt(Dh)) v over packed qkv [B,N,3d] -> [B,N,d] (head_dim must be 64)."""
    qkv = L.dev_f32(qkv, "qkv")
    B, N, d3 = qkv.shape
    kpm = key_padding_bytes(key_padding_mask, B, N, qkv.device)
    d = d3 // 3
    dh = d // n_heads
    out = torch.empty(B, N, d, device=qkv.device, dtype=torch.float32) if n_query in (None, N) else \
        torch.zeros(B, N, d, device=qkv.device, dtype=torch.float32)
    L.check(L.lib().avd_attn_fwd_f32(qkv.data_ptr(), out.data_ptr(), B, N, n_heads, dh, 1.0 / math.sqrt(dh),
                                     N if n_query is None else n_query, L.ptr(kpm), _st(qkv)))
    return out


def layernorm_act(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5, act: int = L.ACT_NONE) -> Tensor:
    x = L.dev_f32(x, "x")
    d = x.shape[-1]
    y = torch.empty_like(x)
    L.check(L.lib().avd_layernorm_act_f32(x.data_ptr(), L.dev_f32(weight).data_ptr(), L.dev_f32(bias).data_ptr(),
                                          y.data_ptr(), x.numel() // d, d, eps, act, _st(x)))
    return y


_FREQS = {}


def temb_freqs(dim: int, max_period: int, device: torch.device) -> Tensor:
    """f_i = exp(-ln(max_period) * i / half), built on the host with the reference's fp32 op sequence
    (schedule_utils.py:80) and uploaded once per (dim, max_period, device)."""
    key = (dim, max_period, str(device))
    if key not in _FREQS:
        half = dim // 2
        f = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
        _FREQS[key] = f.to(device)
    return _FREQS[key]


def timestep_embedding(t: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    if not t.is_cuda:
        raise L.AvdError("timesteps must be on the ROCm device (no CPU fallback)")
    if t.is_floating_point():
        # the reference accepts float timesteps; the kernel takes int64 like every caller on the hot path
        if not torch.equal(t, t.round()):
            raise L.AvdError("fractional timesteps are not supported by the HIP path")
    t = L.dev_i64(t, t.device)
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=torch.float32)
    fr = temb_freqs(dim, max_period, t.device) if dim >= 2 else None
    L.check(L.lib().avd_timestep_embedding_f32(t.data_ptr(), L.ptr(fr), out.data_ptr(), t.shape[0], dim,
                                               float(max_period), _st(out)))
    return out


def tube_patch(z: Tensor, t: int, h: int, w: int) -> Tensor:
    z = L.dev_f32(z, "z")
    B, C_, T, H, W = z.shape
    assert T % t == 0 and H % h == 0 and W % w == 0, "tube sizes must divide latent dims"
    n = (T // t) * (H // h) * (W // w)
    tok = torch.empty(B, n, C_ * t * h * w, device=z.device, dtype=torch.float32)
    L.check(L.lib().avd_tube_patch_f32(z.data_ptr(), tok.data_ptr(), B, C_, T, H, W, t, h, w, _st(z)))
    return tok


def tube_unpatch(tokens: Tensor, C_: int, T: int, H: int, W: int, t: int, h: int, w: int) -> Tensor:
    tokens = L.dev_f32(tokens, "tokens")
    B, N, D = tokens.shape
    assert D == C_ * t * h * w, "token width mismatch"
    assert N == (T // t) * (H // h) * (W // w), "token count mismatch"
    z = torch.empty(B, C_, T, H, W, device=tokens.device, dtype=torch.float32)
    L.check(L.lib().avd_tube_unpatch_f32(tokens.data_ptr(), z.data_ptr(), B, C_, T, H, W, t, h, w, _st(z)))
    return z


def audio_tokens(z_a: Tensor, length: int, stride: int) -> Tensor:
    z_a = L.dev_f32(z_a, "z_a")
    B, Ca, F = z_a.shape
    na = (F - length) // stride + 1
    tok = torch.empty(B, na, Ca * length, device=z_a.device, dtype=torch.float32)
    L.check(L.lib().avd_audio_tokens_f32(z_a.data_ptr(), tok.data_ptr(), B, Ca, F, length, stride, _st(z_a)))
    return tok


def audio_untokens(tokens: Tensor, Ca: int, length: int, frames: int, stride: int, window: Optional[Tensor] = None) -> Tensor:
    tokens = L.dev_f32(tokens, "tokens")
    if window is not None:
        window = L.dev_f32(window, "window")
        assert window.numel() == length
    B, na, D = tokens.shape
    assert D == Ca * length
    if na != (frames - length) // stride + 1:
        raise L.AvdError("audio_untokens: token count does not match (frames, length, stride)")
    z = torch.empty(B, Ca, frames, device=tokens.device, dtype=torch.float32)
    L.check(L.lib().avd_audio_untokens_f32(tokens.data_ptr(), L.ptr(window), z.data_ptr(), B, Ca, frames, length, stride, _st(z)))
    return z


def ddim_step(x_t: Tensor, t_now: Tensor, t_prev: Tensor, eps_hat: Tensor, alpha_bar: Tensor, eta: float = 0.0,
              noise: Optional[Tensor] = None) -> Tensor:
    x_t = L.dev_f32(x_t, "x_t")
    eps_hat = L.dev_f32(eps_hat, "eps_hat")
    if eps_hat.shape != x_t.shape:
        raise RuntimeError("eps_hat must have the shape of x_t")
    dev = x_t.device
    ab = alpha_bar if (alpha_bar.is_cuda and alpha_bar.dtype == torch.float32) else alpha_bar.to(dev, torch.float32)
    ab = ab.contiguous()
    tn, tp = L.dev_i64(t_now, dev), L.dev_i64(t_prev, dev)
    B = x_t.shape[0]
    if tn.numel() != B or tp.numel() != B:
        raise RuntimeError("t_now / t_prev must have one entry per sample")
    if eta > 0.0 and noise is None:
        noise = torch.randn_like(x_t)          # the reference draws randn_like here (schedule_utils.py:197)
    nz = None if noise is None else L.dev_f32(noise, "noise")
    out = torch.empty_like(x_t)
    L.check(L.lib().avd_ddim_step_f32(x_t.data_ptr(), eps_hat.data_ptr(), tn.data_ptr(), tp.data_ptr(), ab.data_ptr(),
                                      ab.numel(), float(eta), L.ptr(nz), out.data_ptr(), B, x_t.numel() // B, _st(x_t)))
    return out


# ---- "bf16x3": fp32-accurate Linear on the bf16 matrix pipe (csrc/gemm_bf16x3.hip) ----
def split3(x: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """fp32 [rows, K] -> its split3 image (uint8; three bf16 planes, tiled).  K must be a multiple of 16.
    ``out``: an existing image of the same size to overwrite (keeps its address)."""
    x = L.dev_f32(x, "x")
    k = x.shape[-1]
    rows = x.numel() // k
    nbytes = L.lib().avd_split3_bytes(rows, k)
    if nbytes < 0:
        raise L.AvdError(f"split3: K={k} must be a multiple of 16")
    if out is None or out.numel() != nbytes or out.device != x.device:
        out = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    L.check(L.lib().avd_split3_f32(x.data_ptr(), out.data_ptr(), rows, k, _st(x)))
    return out


def rmsnorm_split3(x: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    """RMSNorm(x) written as a split3 image (the A operand of the next bf16x3 Linear)."""
    x = L.dev_f32(x, "x")
    scale = L.dev_f32(scale, "scale")
    d = x.shape[-1]
    rows = x.numel() // d
    out = torch.empty(L.lib().avd_split3_bytes(rows, d), dtype=torch.uint8, device=x.device)
    L.check(L.lib().avd_rmsnorm_split3_f32(x.data_ptr(), scale.data_ptr(), out.data_ptr(), rows, d, eps, _st(x)))
    return out


def linear_bf16x3(x3: Tensor, rows: int, w3: Tensor, n: int, k: int, bias: Optional[Tensor] = None,
                  residual: Optional[Tensor] = None, act: int = L.ACT_NONE, out_split3: bool = False, terms: int = 6) -> Tensor:
    """act(x @ W.T + bias) + residual with both operands given as split3 images; fp32 [rows, n] result, or its split3
    image when out_split3 (bias + GELU only).  terms: 6 default, 9 strict (nothing dropped), 1 plain bf16 operands."""
    b = None if bias is None else L.dev_f32(bias, "bias")
    r = None if residual is None else L.dev_f32(residual, "residual")
    if out_split3:
        out = torch.empty(L.lib().avd_split3_bytes(rows, n), dtype=torch.uint8, device=x3.device)
        c, c3 = None, out.data_ptr()
    else:
        out = torch.empty(rows, n, dtype=torch.float32, device=x3.device)
        c, c3 = out.data_ptr(), None
    L.check(L.lib().avd_gemm_bf16x3_f32(x3.data_ptr(), w3.data_ptr(), L.ptr(b), L.ptr(r), c, c3, rows, n, k, act, terms, _st(x3)))
    return out


# ---- "f16x2": two scaled fp16 planes per operand, three product terms (include/avdiff_hip.h) ----
def f16x2_scale(bound: float) -> float:
    """Largest power of two s with s * bound <= 2^15 (fp16 tops out at 65504), for a bound on |x| over the image.
    The bound is widened by 1 % first so that fp32 rounding of the values it covers cannot cross it."""
    import math
    bound = float(bound) * 1.01
    if not math.isfinite(bound):
        raise L.AvdError("f16x2: the bound on the operand's magnitude is not finite")
    if bound <= 0.0:
        return 1.0
    e = 15 - math.ceil(math.log2(bound))
    return float(2.0 ** max(-100, min(100, e)))


def weight_bounds(tensors) -> list:
    """[(max|w|, max row 2-norm), ...] for a list of fp32 device tensors (vectors count as one row), computed by the library
    (``avd_weight_bounds_f32``); one device-to-host copy for the whole list."""
    tensors = [L.dev_f32(t.detach(), "weight") for t in tensors]
    if not tensors:
        return []
    out = torch.empty(len(tensors), 2, dtype=torch.float32, device=tensors[0].device)
    for i, t in enumerate(tensors):
        cols = t.shape[-1]
        rows = t.numel() // cols
        L.check(L.lib().avd_weight_bounds_f32(t.data_ptr(), rows, cols, out[i].data_ptr(), _st(t)))
    return [(float(a), float(b)) for a, b in out.cpu().tolist()]


def split_f16x2(x: Tensor, scale: Optional[float] = None, out: Optional[Tensor] = None):
    """fp32 [rows, K] -> (f16x2 image, scale).  scale None: derived from max|x| (one device sync)."""
    x = L.dev_f32(x, "x")
    k = x.shape[-1]
    rows = x.numel() // k
    nbytes = L.lib().avd_split3_bytes(rows, k)
    if nbytes < 0:
        raise L.AvdError(f"split_f16x2: K={k} must be a multiple of 16")
    if scale is None:
        scale = f16x2_scale(weight_bounds([x.reshape(rows, k)])[0][0])
    if out is None or out.numel() != nbytes or out.device != x.device:
        out = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    L.check(L.lib().avd_split_f16x2_f32(x.data_ptr(), out.data_ptr(), rows, k, scale, _st(x)))
    return out, scale


def rmsnorm_split_f16x2(x: Tensor, gamma: Tensor, eps: float = 1e-6, scale: Optional[float] = None):
    """RMSNorm(x) written as an f16x2 image; the default scale uses |y_i| <= sqrt(d) max|gamma|."""
    x = L.dev_f32(x, "x")
    gamma = L.dev_f32(gamma, "scale")
    d = x.shape[-1]
    rows = x.numel() // d
    if scale is None:
        scale = f16x2_scale(weight_bounds([gamma])[0][0] * d ** 0.5)
    out = torch.empty(L.lib().avd_split3_bytes(rows, d), dtype=torch.uint8, device=x.device)
    L.check(L.lib().avd_rmsnorm_split_f16x2_f32(x.data_ptr(), gamma.data_ptr(), out.data_ptr(), rows, d, eps, scale, _st(x)))
    return out, scale


def linear_f16x2(x2: Tensor, rows: int, w2: Tensor, n: int, k: int, ab_scale: float, bias: Optional[Tensor] = None,
                 residual: Optional[Tensor] = None, act: int = L.ACT_NONE, out_scale: Optional[float] = None) -> Tensor:
    """act(x @ W.T + bias) + residual with both operands given as f16x2 images of scales s_x, s_w (ab_scale = s_x * s_w);
    fp32 [rows, n] result, or — out_scale given — its f16x2 image at that scale (bias + GELU only)."""
    b = None if bias is None else L.dev_f32(bias, "bias")
    r = None if residual is None else L.dev_f32(residual, "residual")
    if out_scale is not None:
        out = torch.empty(L.lib().avd_split3_bytes(rows, n), dtype=torch.uint8, device=x2.device)
        c, c2 = None, out.data_ptr()
    else:
        out = torch.empty(rows, n, dtype=torch.float32, device=x2.device)
        c, c2 = out.data_ptr(), None
    L.check(L.lib().avd_gemm_f16x2_f32(x2.data_ptr(), w2.data_ptr(), L.ptr(b), L.ptr(r), c, c2, rows, n, k, act, ab_scale,
                                       1.0 if out_scale is None else out_scale, _st(x2)))
    return out
