"""CPU oracle: a from-scratch functional restatement of the reference's denoising hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Parity status: **pinned** —
``tests/test_oracle_golden.py`` checks every function here against golden vectors that
``tools/make_golden.py`` produced by importing the reference itself in the build container
(fixtures committed under ``tests/golden/``), plus the one value-level test the reference
ships (patch/unpatch round trip, ``tests/test_shapes.py:26-36`` in the reference).

Everything is written as plain functions over a flat ``{name: tensor}`` weight dict (the
reference modules' ``state_dict`` key names), in whatever float dtype the inputs carry, so the
same code serves as the fp32 oracle, as an fp64 adjudicator and as the timed ``cpu_baseline``.

Reference anchors (paths relative to the reference repo root):
  schedules/ddim .......... avdiff/utils/schedule_utils.py:14-49, 52-57, 64-86, 132-143, 146-200
  patch / chunk ........... avdiff/utils/ops.py:17-45, 48-93, 100-119, 122-144
  audio token helpers ..... avdiff/models/infer/sample_clip.py:184-188, 191-215
  RMSNorm/MHA/MLP/Block ... avdiff/models/mmdt.py:33-42, 51-61, 66-83, 88-99, 134-149
  noise head .............. avdiff/models/heads/noise_heads.py:185-229
  adapters + t-emb concat . avdiff/models/infer/sample_clip.py:48-56, 59-70
  sampler loop body ....... avdiff/models/infer/sample_clip.py:318-348 (V->A), 359-389 (A->V); chained: sample_a2v / sample_v2a (:313-394)
  TimestepEmbedder(mlp) ... avdiff/models/adapters.py:137-158
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


# --------------------------------------------------------------------------------------
# schedule tables (schedule_utils.py:14-57, 132-143) — always fp32 like the reference
# --------------------------------------------------------------------------------------

def beta_table(steps: int, kind: str = "cosine", min_beta: float = 1e-4, max_beta: float = 2e-2) -> Tensor:
    k = kind.lower()
    if k == "linear":
        b = torch.linspace(min_beta, max_beta, steps, dtype=torch.float32)
    elif k == "sigmoid":
        ramp = torch.sigmoid(torch.linspace(-6, 6, steps, dtype=torch.float32))
        b = min_beta + (max_beta - min_beta) * ramp
    elif k == "cosine":
        grid = torch.linspace(0, steps, steps + 1, dtype=torch.float32)
        f = torch.cos(((grid / steps + 0.008) / 1.008) * math.pi / 2) ** 2
        f = f / f[0]
        b = 1 - f[1:] / f[:-1]
    else:
        raise ValueError(f"Unknown schedule kind: {kind}")
    return b.clamp(1e-8, 0.999)


def alpha_bar_table(betas: Tensor) -> Tensor:
    return torch.cumprod(1.0 - betas.to(torch.float32), dim=0)


def sampling_schedule(t_train: int, t_sample: int) -> Tensor:
    # linspace from T-1 down to -1 inclusive, round-half-even, int64 (schedule_utils.py:132-143)
    return torch.round(torch.linspace(t_train - 1, -1, t_sample + 1)).to(torch.long)


# --------------------------------------------------------------------------------------
# sinusoidal timestep embedding (schedule_utils.py:64-86): [cos | sin], cos first
# --------------------------------------------------------------------------------------

def timestep_embedding(t: Tensor, dim: int, max_period: int = 10000, dtype=torch.float32) -> Tensor:
    tf = t if t.is_floating_point() else t.to(torch.float32)
    half = dim // 2
    # the reference builds freqs in fp32 regardless of anything else
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    ang = tf[:, None] * freqs[None, :]
    out = torch.cat([ang.cos(), ang.sin()], dim=1)
    if dim % 2:
        out = torch.nn.functional.pad(out, (0, 1))
    return out.to(dtype)


# --------------------------------------------------------------------------------------
# DDIM update (schedule_utils.py:146-200).  eta>0 needs externally supplied noise.
# --------------------------------------------------------------------------------------

def ddim_update(x_t: Tensor, t_now: Tensor, t_prev: Tensor, eps_hat: Tensor, alpha_bar: Tensor,
                eta: float = 0.0, noise: Optional[Tensor] = None) -> Tensor:
    ab = alpha_bar.to(x_t.dtype)
    a_t = ab[t_now.clamp(min=0).long()]
    a_p = torch.where(t_prev >= 0, ab[t_prev.clamp(min=0).long()], torch.ones_like(a_t))
    shape = [-1] + [1] * (x_t.dim() - 1)
    a_t = a_t.view(shape)
    a_p = a_p.view(shape)
    x0 = (x_t - (1.0 - a_t).clamp(min=0).sqrt() * eps_hat) / a_t.sqrt().clamp(min=1e-8)
    if eta > 0.0:
        frac = ((1.0 - a_p) / (1.0 - a_t).clamp(min=1e-8)).clamp(min=0)
        omr = (1.0 - a_t / a_p.clamp(min=1e-8)).clamp(min=0)
        sigma = eta * (frac * omr).sqrt()
        if noise is None:
            raise ValueError("oracle ddim_update with eta>0 needs explicit noise")
    else:
        sigma = torch.zeros_like(a_t)
        noise = torch.zeros_like(x_t)
    c_eps = (1.0 - a_p - sigma ** 2).clamp(min=0).sqrt()
    return a_p.sqrt() * x0 + c_eps * eps_hat + sigma * noise


# --------------------------------------------------------------------------------------
# tokenisers (ops.py:100-144; sample_clip.py:184-215)
# --------------------------------------------------------------------------------------

def video_patch_index(C: int, T: int, H: int, W: int, t: int, h: int, w: int) -> Tensor:
    """Flat gather map: tokens.flatten()[i] = latent.flatten()[idx[i]] for one sample.

    Token order (T',H',W') row-major; feature order (C,t,h,w) row-major — ops.py:114-116.
    """
    if T % t or H % h or W % w:
        raise AssertionError("tube sizes must divide latent dims")
    base = torch.arange(C * T * H * W).view(C, T // t, t, H // h, h, W // w, w)
    return base.permute(1, 3, 5, 0, 2, 4, 6).reshape(-1)


def tube_patch(z: Tensor, t: int, h: int, w: int) -> Tensor:
    B, C, T, H, W = z.shape
    idx = video_patch_index(C, T, H, W, t, h, w)
    n = (T // t) * (H // h) * (W // w)
    return z.reshape(B, -1)[:, idx].reshape(B, n, C * t * h * w)


def tube_unpatch(tok: Tensor, C: int, T: int, H: int, W: int, t: int, h: int, w: int) -> Tensor:
    B, n, D = tok.shape
    if D != C * t * h * w:
        raise AssertionError("token width mismatch")
    if n != (T // t) * (H // h) * (W // w):
        raise AssertionError("token count mismatch")
    idx = video_patch_index(C, T, H, W, t, h, w)
    out = torch.empty(B, C * T * H * W, dtype=tok.dtype)
    out[:, idx] = tok.reshape(B, -1)
    return out.view(B, C, T, H, W)


def audio_tokens(z_a: Tensor, length: int, stride: int) -> Tensor:
    """[B,Ca,F] -> [B,Na,Ca*length]; Na = (F-length)//stride + 1; trailing frames dropped."""
    B, Ca, F = z_a.shape
    na = (F - length) // stride + 1
    starts = torch.arange(na) * stride
    win = z_a[:, :, starts[:, None] + torch.arange(length)[None, :]]   # [B,Ca,Na,l]
    return win.permute(0, 2, 1, 3).reshape(B, na, Ca * length)


def audio_untokens(tok: Tensor, Ca: int, length: int, frames: int, stride: int, hann: bool = False) -> Tensor:
    """Overlap-add with a rectangular (or, ops.py's apply_hann, Hann) window and summed-weight normalisation, then crop /
    zero-pad to ``frames`` (sample_clip.py:191-215 → ops.py:48-93)."""
    B, na, D = tok.shape
    assert D == Ca * length
    win = tok.view(B, na, Ca, length).permute(0, 2, 1, 3)              # [B,Ca,Na,l]
    L = (na - 1) * stride + length
    acc = torch.zeros(B, Ca, L, dtype=tok.dtype)
    cnt = torch.zeros(L, dtype=tok.dtype)
    w = torch.hann_window(length, dtype=tok.dtype) if hann else torch.ones(length, dtype=tok.dtype)
    for i in range(na):
        acc[..., i * stride:i * stride + length] += win[:, :, i] * w
        cnt[i * stride:i * stride + length] += w
    acc = acc / cnt.clamp(min=1e-8)
    if L >= frames:
        return acc[..., :frames].contiguous()
    return torch.nn.functional.pad(acc, (0, frames - L))


# --------------------------------------------------------------------------------------
# MMDiT core (mmdt.py)
# --------------------------------------------------------------------------------------

def gelu_erf(x: Tensor) -> Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def rmsnorm(x: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    # eps is added to the RMS (outside the sqrt) — mmdt.py:39-42
    rms = x.pow(2).sum(-1, keepdim=True).sqrt() / math.sqrt(x.shape[-1])
    return scale * x / (rms + eps)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor] = None) -> Tensor:
    y = x @ w.transpose(-1, -2)
    return y if b is None else y + b


def self_attention(x: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor, n_heads: int,
                   key_padding_mask: Optional[Tensor] = None) -> Tensor:
    """nn.MultiheadAttention(batch_first=True), q=k=v=x, eval (mmdt.py:51-61); key_padding_mask [B,N] True = ignore the key."""
    B, N, d = x.shape
    dh = d // n_heads
    qkv = linear(x, w_in, b_in).view(B, N, 3, n_heads, dh)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))          # [B,H,N,dh]
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(dh))
    if key_padding_mask is not None:
        s = s.masked_fill(key_padding_mask.bool()[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, d)
    return linear(o, w_out, b_out)


def _norm(x: Tensor, W: Weights, pre: str) -> Tensor:
    """build_norm (mmdt.py:44-45): RMSNorm keys `scale`, nn.LayerNorm keys `weight` / `bias` (eps 1e-5)."""
    if pre + "scale" in W:
        return rmsnorm(x, W[pre + "scale"])
    return layernorm(x, W[pre + "weight"], W[pre + "bias"])


def mmdit_block(x: Tensor, W: Weights, pre: str, n_heads: int, key_padding_mask: Optional[Tensor] = None) -> Tensor:
    h = _norm(x, W, pre + "norm1.")
    x = x + self_attention(h, W[pre + "attn.mha.in_proj_weight"], W[pre + "attn.mha.in_proj_bias"],
                           W[pre + "attn.mha.out_proj.weight"], W[pre + "attn.mha.out_proj.bias"], n_heads, key_padding_mask)
    h = _norm(x, W, pre + "norm2.")
    h = gelu_erf(linear(h, W[pre + "mlp.fc1.weight"], W[pre + "mlp.fc1.bias"]))
    return x + linear(h, W[pre + "mlp.fc2.weight"], W[pre + "mlp.fc2.bias"])


def mmdit_forward(x: Tensor, W: Weights, n_layers: int, n_heads: int, key_padding_mask: Optional[Tensor] = None) -> Tensor:
    for i in range(n_layers):
        x = mmdit_block(x, W, f"blocks.{i}.", n_heads, key_padding_mask)
    return _norm(x, W, "final_norm.")


# --------------------------------------------------------------------------------------
# MultiModalNoiseHead (noise_heads.py:185-229) for num_modality_specific_layers == 1
# --------------------------------------------------------------------------------------

def layernorm(x: Tensor, g: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) / (var + eps).sqrt() * g + b


_HEAD_ACTS = {"gelu": gelu_erf, "relu": lambda t: t.clamp(min=0), "leaky_relu": lambda t: torch.where(t > 0, t, 0.1 * t)}


def noise_head(h: Tensor, W: Weights, modality: str, n_shared: int = 2, activation: str = "gelu") -> Tensor:
    shp = h.shape
    act = _HEAD_ACTS[activation]                          # noise_heads.py:28-36
    y = linear(h.reshape(-1, shp[-1]), W[f"input_proj.{modality}.weight"], W[f"input_proj.{modality}.bias"])
    for j in range(n_shared):
        y = linear(y, W[f"shared.{j}.0.weight"], W[f"shared.{j}.0.bias"])
        y = act(layernorm(y, W[f"shared.{j}.1.weight"], W[f"shared.{j}.1.bias"]))
    y = linear(y, W[f"out_proj.{modality}.weight"], W[f"out_proj.{modality}.bias"])
    return y.view(*shp[:-1], y.shape[-1])


# --------------------------------------------------------------------------------------
# TimestepEmbedder(mode="mlp") (adapters.py:137-158): sinusoid -> Linear -> SiLU -> Linear
# --------------------------------------------------------------------------------------

def timestep_mlp(t: Tensor, W: Weights, dim: int) -> Tensor:
    e = timestep_embedding(t, dim, dtype=W["mlp.0.weight"].dtype)
    y = linear(e, W["mlp.0.weight"], W["mlp.0.bias"])
    y = y * torch.sigmoid(y)
    return linear(y, W["mlp.2.weight"], W["mlp.2.bias"])


# --------------------------------------------------------------------------------------
# one CFG denoising step, batched (sample_clip.py loop bodies)
# --------------------------------------------------------------------------------------

def embed_with_time(tok: Tensor, w: Tensor, b: Tensor, t: Tensor, tdim: int, mode: str = "concat") -> Tensor:
    """mode "concat": sampler (sample_clip.py:59-70); "add": trainer (train/trainer.py:45-49, embedding at token width)."""
    x = linear(tok, w, b)
    if mode == "add":
        return x + timestep_embedding(t, x.shape[-1], dtype=x.dtype)[:, None, :]
    e = timestep_embedding(t, tdim, dtype=x.dtype)[:, None, :].expand(-1, x.shape[1], -1)
    return torch.cat([x, e], dim=-1)


def eps_pair(Xt: Tensor, Xp: Tensor, target_first: bool, core: Weights, head: Weights, target: str,
             n_layers: int, n_heads: int) -> Tuple[Tensor, Tensor]:
    """cond / null ε̂ for the target modality.  Sequence order is always [video ; audio]."""
    nt = Xt.shape[1]

    def run(prompt_rows: Tensor) -> Tensor:
        seq = torch.cat([Xt, prompt_rows], 1) if target_first else torch.cat([prompt_rows, Xt], 1)
        hfull = mmdit_forward(seq, core, n_layers, n_heads)
        ht = hfull[:, :nt] if target_first else hfull[:, -nt:]
        return noise_head(ht, head, target)

    return run(Xp), run(torch.zeros_like(Xp))


def denoise_step_a2v(z_v: Tensor, z_a0: Tensor, t_now: Tensor, t_prev: Tensor, alpha_bar: Tensor, *,
                     adapt_v: Weights, adapt_a: Weights, core: Weights, head: Weights,
                     n_layers: int, n_heads: int, tdim: int = 256, tube=(2, 4, 4), chunk=(4, 4),
                     guidance: float = 3.5, eta: float = 0.0, return_eps: bool = False, temb_mode: str = "concat"):
    """Audio prompt -> video target (sample_clip.py:359-389), any batch size."""
    B, C, T, H, Wd = z_v.shape
    tok_v = tube_patch(z_v, *tube)
    tok_a = audio_tokens(z_a0, *chunk)
    Xv = embed_with_time(tok_v, adapt_v["proj.weight"], adapt_v["proj.bias"], t_now, tdim, temb_mode)
    Xa = embed_with_time(tok_a, adapt_a["proj.weight"], adapt_a["proj.bias"], torch.zeros_like(t_now), tdim, temb_mode)
    e_c, e_n = eps_pair(Xv, Xa, True, core, head, "video", n_layers, n_heads)
    eps_tok = e_n + guidance * (e_c - e_n)
    eps_lat = tube_unpatch(eps_tok, C, T, H, Wd, *tube)
    z_next = ddim_update(z_v, t_now, t_prev, eps_lat, alpha_bar, eta)
    return (z_next, eps_tok) if return_eps else z_next


def denoise_step_v2a(z_a: Tensor, z_v0: Tensor, t_now: Tensor, t_prev: Tensor, alpha_bar: Tensor, *,
                     adapt_v: Weights, adapt_a: Weights, core: Weights, head: Weights,
                     n_layers: int, n_heads: int, tdim: int = 256, tube=(2, 4, 4), chunk=(4, 4),
                     guidance: float = 3.0, eta: float = 0.0, return_eps: bool = False):
    """Video prompt -> audio target (sample_clip.py:318-348), any batch size."""
    B, Ca, F = z_a.shape
    tok_v = tube_patch(z_v0, *tube)
    tok_a = audio_tokens(z_a, *chunk)
    Xv = embed_with_time(tok_v, adapt_v["proj.weight"], adapt_v["proj.bias"], torch.zeros_like(t_now), tdim)
    Xa = embed_with_time(tok_a, adapt_a["proj.weight"], adapt_a["proj.bias"], t_now, tdim)
    e_c, e_n = eps_pair(Xa, Xv, False, core, head, "audio", n_layers, n_heads)
    eps_tok = e_n + guidance * (e_c - e_n)
    eps_lat = audio_untokens(eps_tok, Ca, chunk[0], F, chunk[1])
    z_next = ddim_update(z_a, t_now, t_prev, eps_lat, alpha_bar, eta)
    return (z_next, eps_tok) if return_eps else z_next


def sample_a2v(z_v: Tensor, z_a0: Tensor, sched: Tensor, alpha_bar: Tensor, **kw) -> Tensor:
    """Chained A->V loop over a sampling schedule (no codec/VAE): returns the final latent."""
    B = z_v.shape[0]
    for i in range(len(sched) - 1):
        z_v = denoise_step_a2v(z_v, z_a0, sched[i].repeat(B), sched[i + 1].repeat(B), alpha_bar, **kw)
    return z_v


def sample_v2a(z_a: Tensor, z_v0: Tensor, sched: Tensor, alpha_bar: Tensor, **kw) -> Tensor:
    """Chained V->A loop over a sampling schedule (sample_clip.py:313-352: the per-step statements :318-348 in the loop the
    function's A->V branch uses; the reference's own V->A branch dies on a permute before it gets here, :286-289): final audio latent."""
    B = z_a.shape[0]
    for i in range(len(sched) - 1):
        z_a = denoise_step_v2a(z_a, z_v0, sched[i].repeat(B), sched[i + 1].repeat(B), alpha_bar, **kw)
    return z_a


# --------------------------------------------------------------------------------------
# synthetic weights with the reference's shapes and init families (no reference code involved)
# --------------------------------------------------------------------------------------

def synth_like(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Weights:
    """Seeded stand-in weights for ANY module given its ``state_dict`` key -> shape table (keys walked in sorted order, one CPU
    generator): ``*.weight`` with two or more dims ~ U(-a, a), a = 1 / sqrt(fan_in); one-dim ``*.weight`` (norm scales) ~ 1 + 0.05 N(0,1);
    ``*.bias`` ~ 0.05 N(0,1); anything else ~ 0.05 N(0,1).  Used where a fixture would otherwise have to store a codec's / VAE's weights:
    the generating script loads these into the reference modules (``strict=True``), the tests load the same into the mirrors."""
    g = torch.Generator().manual_seed(seed)
    out: Weights = {}
    for name in sorted(shapes):
        shp = tuple(int(v) for v in shapes[name])
        if name.endswith("weight") and len(shp) >= 2:
            fan_in = 1
            for v in shp[1:]:
                fan_in *= v
            out[name] = (torch.rand(*shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
        elif name.endswith("weight"):
            out[name] = 1.0 + 0.05 * torch.randn(*shp, generator=g)
        else:
            out[name] = 0.05 * torch.randn(*shp, generator=g)
    return out


def _xavier(gen: torch.Generator, out_f: int, in_f: int) -> Tensor:
    a = math.sqrt(6.0 / (in_f + out_f))
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * a


def _kaiming_default(gen: torch.Generator, out_f: int, in_f: int) -> Tuple[Tensor, Tensor]:
    a = 1.0 / math.sqrt(in_f)
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * a, (torch.rand(out_f, generator=gen) * 2 - 1) * a


def synth_weights(seed: int = 0, d: int = 512, n_layers: int = 8, mlp_ratio: float = 4.0,
                  tok_v: int = 256, tok_a: int = 32, tdim: int = 256, head_hidden: int = 512,
                  n_shared: int = 2, bias_jitter: float = 0.02) -> Dict[str, Weights]:
    """Seeded random weights shaped like ``build_components`` output (sample_clip.py:75-109).

    Biases/norm scales get a small jitter so bias/scale plumbing bugs cannot hide behind zeros/ones.
    """
    g = torch.Generator().manual_seed(seed)
    hid = int(d * mlp_ratio)

    def jit(n, base=0.0):
        return base + bias_jitter * torch.randn(n, generator=g)

    core: Weights = {}
    for i in range(n_layers):
        p = f"blocks.{i}."
        core[p + "norm1.scale"] = jit(d, 1.0)
        core[p + "attn.mha.in_proj_weight"] = _xavier(g, 3 * d, d)
        core[p + "attn.mha.in_proj_bias"] = jit(3 * d)
        w, b = _kaiming_default(g, d, d)
        core[p + "attn.mha.out_proj.weight"] = w
        core[p + "attn.mha.out_proj.bias"] = jit(d)
        core[p + "norm2.scale"] = jit(d, 1.0)
        core[p + "mlp.fc1.weight"] = _xavier(g, hid, d)
        core[p + "mlp.fc1.bias"] = jit(hid)
        core[p + "mlp.fc2.weight"] = _xavier(g, d, hid)
        core[p + "mlp.fc2.bias"] = jit(d)
    core["final_norm.scale"] = jit(d, 1.0)

    head: Weights = {}
    for m, od in (("video", tok_v), ("audio", tok_a)):
        head[f"input_proj.{m}.weight"] = _xavier(g, head_hidden, d)
        head[f"input_proj.{m}.bias"] = jit(head_hidden)
        head[f"out_proj.{m}.weight"] = _xavier(g, od, head_hidden)
        head[f"out_proj.{m}.bias"] = jit(od)
    for j in range(n_shared):
        head[f"shared.{j}.0.weight"] = _xavier(g, head_hidden, head_hidden)
        head[f"shared.{j}.0.bias"] = jit(head_hidden)
        head[f"shared.{j}.1.weight"] = jit(head_hidden, 1.0)
        head[f"shared.{j}.1.bias"] = jit(head_hidden)

    wv, bv = _kaiming_default(g, d - tdim, tok_v)
    wa, ba = _kaiming_default(g, d - tdim, tok_a)
    return {
        "core": core,
        "head": head,
        "adapt_v": {"proj.weight": wv, "proj.bias": bv},
        "adapt_a": {"proj.weight": wa, "proj.bias": ba},
    }


def cast_weights(ws: Dict[str, Weights], dtype) -> Dict[str, Weights]:
    return {k: {n: t.to(dtype) for n, t in v.items()} for k, v in ws.items()}


# --------------------------------------------------------------------------------------
# VideoVAE.decode (vae_video3d.py:195-214; _conv_block_3d :79-84) — loop boundary, SURVEY a9 / next-1
# --------------------------------------------------------------------------------------

def _lin_src(out_size: int, in_size: int, dtype):
    """torch's align_corners=False source index rule: src = max(scale*(dst+0.5)-0.5, 0), i1 = min(i0+1, in-1)."""
    scale = in_size / out_size
    src = (scale * (torch.arange(out_size, dtype=torch.float32) + 0.5) - 0.5).clamp(min=0.0)
    i0 = src.floor().long().clamp(max=in_size - 1)
    i1 = (i0 + 1).clamp(max=in_size - 1)
    l1 = (src - i0.float()).to(dtype)
    return i0, i1, 1.0 - l1, l1


def trilinear_upsample(x: Tensor, size) -> Tensor:
    """[B,C,t,h,w] -> [B,C,T,H,W], trilinear, align_corners=False (F.interpolate semantics)."""
    T, H, W = size
    t0, t1, tl0, tl1 = _lin_src(T, x.shape[2], x.dtype)
    h0, h1, hl0, hl1 = _lin_src(H, x.shape[3], x.dtype)
    w0, w1, wl0, wl1 = _lin_src(W, x.shape[4], x.dtype)

    def at(ti, hi, wi):
        return x[:, :, ti][:, :, :, hi][:, :, :, :, wi]

    tl0, tl1 = tl0.view(1, 1, -1, 1, 1), tl1.view(1, 1, -1, 1, 1)
    hl0, hl1 = hl0.view(1, 1, 1, -1, 1), hl1.view(1, 1, 1, -1, 1)
    return (tl0 * (hl0 * (wl0 * at(t0, h0, w0) + wl1 * at(t0, h0, w1)) + hl1 * (wl0 * at(t0, h1, w0) + wl1 * at(t0, h1, w1))) +
            tl1 * (hl0 * (wl0 * at(t1, h0, w0) + wl1 * at(t1, h0, w1)) + hl1 * (wl0 * at(t1, h1, w0) + wl1 * at(t1, h1, w1))))


def group_norm(x: Tensor, groups: int, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    B, C = x.shape[:2]
    g = x.reshape(B, groups, -1)
    mu = g.mean(-1, keepdim=True)
    var = (g - mu).pow(2).mean(-1, keepdim=True)
    y = ((g - mu) / (var + eps).sqrt()).reshape(x.shape)
    shp = [1, C] + [1] * (x.dim() - 2)
    return y * w.view(shp) + b.view(shp)


def vae_decode(z: Tensor, W: Weights, t_down: int = 4, s_down: int = 8, n_blocks: int = 2, out_act: str = "sigmoid",
               out_size=None) -> Tensor:
    conv3d = torch.nn.functional.conv3d
    h = conv3d(z, W["from_lat.weight"], W["from_lat.bias"])
    size = out_size or (z.shape[2] * t_down, z.shape[3] * s_down, z.shape[4] * s_down)
    h = trilinear_upsample(h, size)
    for i in range(n_blocks):
        cw = W[f"dec_net.{i}.0.weight"]
        h = conv3d(h, cw, W[f"dec_net.{i}.0.bias"], padding=1)
        h = gelu_erf(h)
        h = group_norm(h, min(8, cw.shape[0]), W[f"dec_net.{i}.2.weight"], W[f"dec_net.{i}.2.bias"])
    x = conv3d(h, W["to_img.weight"], W["to_img.bias"])
    return torch.sigmoid(x) if out_act == "sigmoid" else torch.tanh(x)


def synth_vae_decoder(seed: int = 0, cv: int = 8, base: int = 64, n_blocks: int = 2, out_ch: int = 3) -> Weights:
    g = torch.Generator().manual_seed(seed)

    def u(*shape, fan_in):
        a = 1.0 / math.sqrt(fan_in)
        return (torch.rand(*shape, generator=g) * 2 - 1) * a

    W: Weights = {"from_lat.weight": u(base, cv, 1, 1, 1, fan_in=cv), "from_lat.bias": u(base, fan_in=cv)}
    for i in range(n_blocks):
        W[f"dec_net.{i}.0.weight"] = u(base, base, 3, 3, 3, fan_in=27 * base)
        W[f"dec_net.{i}.0.bias"] = u(base, fan_in=27 * base)
        W[f"dec_net.{i}.2.weight"] = 1.0 + 0.1 * torch.randn(base, generator=g)
        W[f"dec_net.{i}.2.bias"] = 0.1 * torch.randn(base, generator=g)
    W["to_img.weight"] = u(out_ch, base, 1, 1, 1, fan_in=base)
    W["to_img.bias"] = u(out_ch, fan_in=base)
    return W


def vae_encode(x: Tensor, W: Weights, t_down: int = 4, s_down: int = 8, n_blocks: int = 2, variational: bool = False):
    """VideoVAE.encode (vae_video3d.py:164-189); x already divisible by the down factors.  variational (eval): returns
    (z = to_mu(h), kld) as :175-185."""
    conv3d = torch.nn.functional.conv3d
    h = x
    for i in range(n_blocks):
        cw = W[f"enc_net.{i}.0.weight"]
        h = gelu_erf(conv3d(h, cw, W[f"enc_net.{i}.0.bias"], padding=1))
        h = group_norm(h, min(8, cw.shape[0]), W[f"enc_net.{i}.2.weight"], W[f"enc_net.{i}.2.bias"])
    B, Cc, T, H, Wd = h.shape
    h = h.view(B, Cc, T // t_down, t_down, H // s_down, s_down, Wd // s_down, s_down).mean(dim=(3, 5, 7))
    if variational:
        mu, logv = conv3d(h, W["to_mu.weight"], W["to_mu.bias"]), conv3d(h, W["to_logv.weight"], W["to_logv.bias"])
        return mu, 0.5 * torch.mean(-1 - logv + mu.pow(2) + logv.exp())
    return conv3d(h, W["to_lat.weight"], W["to_lat.bias"])


# --------------------------------------------------------------------------------------
# AudioCodec (audio_codec.py:184-214) — loop boundary, next-2
# --------------------------------------------------------------------------------------

def _exact_pool(Lw: int, Fa: int):
    hop = max(1, int(round(Lw / Fa)))
    if Fa * hop < Lw:
        hop += 1
    return hop, Fa * hop


def codec_encode(wav: Tensor, W: Weights, frames_per_clip: Optional[int] = 150, hop: int = 320) -> Tensor:
    conv1d = torch.nn.functional.conv1d
    h = gelu_erf(conv1d(wav, W["pre.0.0.weight"], W["pre.0.0.bias"], padding=4))
    h = gelu_erf(conv1d(h, W["pre.1.0.weight"], W["pre.1.0.bias"], padding=4))
    Lw = h.shape[-1]
    if frames_per_clip is None:
        Fa, hp = math.ceil(Lw / hop), hop
    else:
        Fa = frames_per_clip
        hp, _ = _exact_pool(Lw, Fa)
    total = Fa * hp
    h = torch.nn.functional.pad(h, (0, total - Lw)) if total > Lw else h[..., :total]
    h = h.view(h.shape[0], h.shape[1], Fa, hp).mean(-1)
    return conv1d(h, W["to_lat.weight"], W["to_lat.bias"])


def codec_decode(z: Tensor, W: Weights, hop: int = 320) -> Tensor:
    conv1d = torch.nn.functional.conv1d
    h = conv1d(z, W["from_lat.weight"], W["from_lat.bias"])
    h = h.repeat_interleave(hop, dim=-1)                      # nearest-neighbour x hop
    pad = W["smooth.0.weight"].shape[-1] // 2
    h = gelu_erf(conv1d(h, W["smooth.0.weight"], W["smooth.0.bias"], padding=pad))
    h = gelu_erf(conv1d(h, W["smooth.2.weight"], W["smooth.2.bias"], padding=pad))
    return torch.tanh(conv1d(h, W["smooth.4.weight"], W["smooth.4.bias"], padding=pad))


# --------------------------------------------------------------------------------------
# stream_infer stitching (stream_infer.py:85-143) — next-3.  numpy, fp32, same accumulation order.
# --------------------------------------------------------------------------------------

def crossfade(chunks, w, hop: int):
    """chunks [N, L, ...] float32, w [L] -> weighted overlap-add / summed weights (clamped 1e-6)."""
    import numpy as np
    N, Lw = chunks.shape[:2]
    shape = (Lw,) + (1,) * (chunks.ndim - 2)
    y = np.zeros(((N - 1) * hop + Lw,) + chunks.shape[2:], dtype=np.float32)
    nrm = np.zeros(((N - 1) * hop + Lw,) + (1,) * (chunks.ndim - 2), dtype=np.float32)
    ww = w.astype(np.float32).reshape(shape)
    for i in range(N):
        y[i * hop:i * hop + Lw] += chunks[i] * ww
        nrm[i * hop:i * hop + Lw] += ww
    return (y / np.maximum(nrm, 1e-6)).astype(np.float32)
