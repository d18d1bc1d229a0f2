"""MMDiT denoiser core — host-side mirror of ``avdiff/models/mmdt.py`` over the HIP C ABI.

Same class names, constructor kwargs, ``forward`` signatures and ``state_dict`` keys as the reference
(``blocks.{i}.norm1.scale``, ``blocks.{i}.attn.mha.in_proj_weight`` … ``final_norm.scale``), so a reference
``core`` state dict loads with ``strict=True``.  The modules own parameters only; every FLOP runs in
``libavdiff_hip.so``.  Inference only (the benchmarked path is ``@torch.no_grad`` + ``.eval()``,
sample_clip.py:84-107,220): dropout / token-dropout in training mode and float masks raise; ``rope`` is accepted and
ignored as the reference does (mmdt.py:125-127 never reads it).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn


class RMSNorm(nn.Module):
    """mmdt.py:33-42 — y = scale * x / (||x||/sqrt(d) + eps)."""

    def __init__(self, d: int, eps: float = 1e-6):
        super().__init__()
        self.scale = nn.Parameter(torch.ones(d))
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.rmsnorm(x, self.scale, self.eps)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm as a parameter container (state_dict keys ``weight`` / ``bias``, eps 1e-5) with the HIP kernel underneath —
    what ``build_norm`` returns for anything but "rmsnorm" (mmdt.py:44-45)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.layernorm_act(x, self.weight, self.bias, eps=self.eps, act=L.ACT_NONE)


def build_norm(kind: str, d: int) -> nn.Module:
    return RMSNorm(d) if kind.lower() == "rmsnorm" else LayerNorm(d)


class _OutProj(nn.Module):
    """Parameter holder matching nn.MultiheadAttention.out_proj (NonDynamicallyQuantizableLinear)."""

    def __init__(self, d: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d, d))
        self.bias = nn.Parameter(torch.zeros(d))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))


class _PackedMHA(nn.Module):
    """Parameter holder with nn.MultiheadAttention's key names and default init (packed q|k|v projection)."""

    def __init__(self, d: int, n_heads: int):
        super().__init__()
        if d % n_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads = d, n_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = _OutProj(d)
        nn.init.xavier_uniform_(self.in_proj_weight)


class MHA(nn.Module):
    """mmdt.py:51-61 — joint self-attention over [target ; prompt] tokens."""

    def __init__(self, d_model: int, n_heads: int, attn_dropout: float = 0.0, resid_dropout: float = 0.0):
        super().__init__()
        self.mha = _PackedMHA(d_model, n_heads)
        self.attn_dropout, self.resid_dropout = attn_dropout, resid_dropout

    def forward(self, x, attn_mask=None, key_padding_mask=None, residual=None):
        if attn_mask is not None:
            raise NotImplementedError("attn_mask is always None in the reference (MMDiT.forward passes attn_mask=None, mmdt.py:147)")
        if self.training and (self.attn_dropout > 0 or self.resid_dropout > 0):
            raise NotImplementedError("HIP path is inference-only; call .eval()")
        m = self.mha
        qkv = Fn.linear(x, m.in_proj_weight, m.in_proj_bias)
        o = Fn.attention(qkv, m.num_heads, key_padding_mask=key_padding_mask)
        return Fn.linear(o, m.out_proj.weight, m.out_proj.bias, residual=residual)


class _Lin(nn.Module):
    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(d_out, d_in))
        self.bias = nn.Parameter(torch.zeros(d_out))
        nn.init.xavier_uniform_(self.weight)


class MLP(nn.Module):
    """mmdt.py:66-83 — fc1 -> GELU(erf) -> fc2, xavier weights / zero biases."""

    def __init__(self, d_model: int, mlp_ratio: float = 4.0, dropout: float = 0.0):
        super().__init__()
        hidden = int(d_model * mlp_ratio)
        self.fc1 = _Lin(d_model, hidden)
        self.fc2 = _Lin(hidden, d_model)
        self.dropout = dropout

    def forward(self, x, residual=None):
        if self.training and self.dropout > 0:
            raise NotImplementedError("HIP path is inference-only; call .eval()")
        h = Fn.linear(x, self.fc1.weight, self.fc1.bias, act=L.ACT_GELU)
        return Fn.linear(h, self.fc2.weight, self.fc2.bias, residual=residual)


class Block(nn.Module):
    """mmdt.py:88-99 — pre-norm attention + MLP with residuals (residual adds fused into the GEMM epilogues)."""

    def __init__(self, d_model, n_heads, mlp_ratio, dropout, attn_dropout, norm):
        super().__init__()
        self.norm1 = build_norm(norm, d_model)
        self.attn = MHA(d_model, n_heads, attn_dropout=attn_dropout, resid_dropout=dropout)
        self.norm2 = build_norm(norm, d_model)
        self.mlp = MLP(d_model, mlp_ratio=mlp_ratio, dropout=dropout)

    def forward(self, x, attn_mask=None, key_padding_mask=None):
        x = self.attn(self.norm1(x), attn_mask=attn_mask, key_padding_mask=key_padding_mask, residual=x)
        return self.mlp(self.norm2(x), residual=x)


@dataclass
class MMDiTCfg:
    d_model: int = 1024
    n_layers: int = 16
    n_heads: int = 16
    mlp_ratio: float = 4.0
    dropout: float = 0.1
    attn_dropout: float = 0.0
    norm: str = "rmsnorm"
    rope: bool = False
    token_dropout: float = 0.0


class MMDiT(nn.Module):
    """mmdt.py:116-149.  ``forward`` is ONE C-ABI call (``avd_core_forward_f32``) enqueuing all layer kernels."""

    def __init__(self, d_model=1024, n_layers=16, n_heads=16, mlp_ratio=4.0,
                 dropout=0.1, attn_dropout=0.0, norm="rmsnorm", rope=False, token_dropout=0.0):
        super().__init__()
        # `rope` is stored and never read, exactly as in the reference (mmdt.py:125-127 keeps it in cfg only): a config that sets
        # it builds the same model there and here
        self.cfg = MMDiTCfg(d_model, n_layers, n_heads, mlp_ratio, dropout, attn_dropout, norm, rope, token_dropout)
        self.blocks = nn.ModuleList([Block(d_model, n_heads, mlp_ratio, dropout, attn_dropout, norm)
                                     for _ in range(n_layers)])
        self.final_norm = build_norm(norm, d_model)
        self._ws: Optional[torch.Tensor] = None
        # "auto" (default): "bf16x3" where the split kernels engage, the norm-folded fp32 MFMA kernels where they do not (small
        # batches, key-padding masks, LayerNorm, widths off the 256-column tiles) — the fastest fp32-level path at every shape.
        # "f32": fp32 MFMA everywhere.  "bf16x3": the four projections of every block and the attention run on the bf16 matrix
        # pipe with exactly split operands (fp32-level error, csrc/gemm_bf16x3.hip) whenever the batch is large enough;
        # "bf16x3_strict" keeps all nine product terms; "bf16" keeps one (plain bf16 operands — reduced precision, config C2);
        # "f16x2" holds every operand as two scaled fp16 planes (22 significant bits) and keeps three terms: half the matrix-pipe
        # work of bf16x3, scales derived from the weights (_f16x2_scales).
        self.matmul = "auto"
        # "default": the attention follows `matmul`.  "fp8": QK^T and PV take e4m3 operands (csrc/attn_fp8.hip; needs a split matmul
        # mode) — reduced precision, BASELINE config C5, never the default.
        self.attn = "default"
        self.fold_norms = True        # fp32 path: fold norm1 / norm2 into the neighbouring Linear epilogues (same math, one pass less)
        self._split3: dict = {}
        self._folded: dict = {}

    def _folded_weight(self, name: str, w: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        key = (w.data_ptr(), w._version, scale.data_ptr(), scale._version)
        hit = self._folded.get(name)
        if hit is None or hit[0] != key:
            new = (w.detach() * scale.detach()[None, :]).contiguous()
            if hit is not None and hit[1].shape == new.shape and hit[1].device == new.device:
                hit[1].copy_(new)          # refresh in place: pointer tables (and captured graphs) that hold it stay valid
                new = hit[1]
            hit = (key, new)
            self._folded[name] = hit
        return hit[1]

    def _split3_image(self, name: str, p: torch.Tensor, h2_scale: float = 0.0) -> torch.Tensor:
        """Operand image of a weight: three bf16 planes, or (h2_scale > 0) the two fp16 planes of the f16x2 mode."""
        key = (p.data_ptr(), p._version, tuple(p.shape), float(h2_scale))
        hit = self._split3.get(name)
        if hit is None or hit[0] != key:
            from . import functional as Fn
            old = hit[1] if hit is not None and hit[0][2] == key[2] and hit[1].device == p.device else None
            # same storage when the shape is unchanged
            img = Fn.split_f16x2(p.detach(), h2_scale, out=old)[0] if h2_scale > 0 else Fn.split3(p.detach(), out=old)
            hit = (key, img)
            self._split3[name] = hit
        return hit[1]

    def _f16x2_scales(self, i: int, ps: dict):
        """Power-of-two scales of block i's f16x2 images, from bounds that hold for EVERY input (so fp16 can never overflow):
        RMSNorm output  |y_j| <= sqrt(d) max|gamma|  and  ||y||_2 <= sqrt(d) max|gamma|   (|x_j| <= ||x||_2 = sqrt(d) rms);
        a Linear output |o_n| <= ||a||_2 ||w_n||_2 + |b_n|  (Cauchy-Schwarz);  attention output rows are convex combinations of
        V rows;  |GELU(x)| <= |x|.  Cached per parameter version (one device sync when a parameter changes)."""
        names = ("norm1_scale", "in_proj_weight", "in_proj_bias", "out_proj_weight", "norm2_scale", "fc1_weight", "fc1_bias",
                 "fc2_weight")
        key = tuple((ps[k].data_ptr(), ps[k]._version) for k in names)
        hit = self._split3.get(f"{i}.f16x2")
        if hit is None or hit[0] != key:
            from . import functional as Fn
            d = self.cfg.d_model
            bd = dict(zip(names, Fn.weight_bounds([ps[k] for k in names])))      # (max |w|, max row norm) per tensor, one sync
            g1, g2 = bd["norm1_scale"][0], bd["norm2_scale"][0]
            win, wfc1 = bd["in_proj_weight"][1], bd["fc1_weight"][1]
            bin_, bfc1 = bd["in_proj_bias"][0], bd["fc1_bias"][0]
            m_in, m_out, m_fc1, m_fc2 = (bd[k][0] for k in ("in_proj_weight", "out_proj_weight", "fc1_weight", "fc2_weight"))
            n1 = d ** 0.5 * g1
            n2 = d ** 0.5 * g2
            sc = [Fn.f16x2_scale(m_in), Fn.f16x2_scale(m_out), Fn.f16x2_scale(m_fc1), Fn.f16x2_scale(m_fc2),
                  Fn.f16x2_scale(n1), Fn.f16x2_scale(n1 * win + bin_), Fn.f16x2_scale(n2), Fn.f16x2_scale(n2 * wfc1 + bfc1)]
            hit = (key, sc)
            self._split3[f"{i}.f16x2"] = hit
        return hit[1]

    def _attn_code(self) -> int:
        if self.attn not in ("default", "fp8"):
            raise ValueError(f"attn must be 'default' or 'fp8', got {self.attn!r}")
        if self.attn == "fp8" and self.matmul == "f32":
            raise ValueError("attn='fp8' reads the qkv3 image of the split matmul modes: use matmul='f16x2', 'bf16x3' or 'bf16'")
        return 1 if self.attn == "fp8" else 0

    # ---- pointer table for the composite (rebuilt per call: parameters may have moved) ----
    def weight_table(self):
        dev = next(self.final_norm.parameters()).device
        arr = (L.BlockWeights * len(self.blocks))()
        keep = []
        for i, b in enumerate(self.blocks):
            ln = isinstance(b.norm1, LayerNorm)
            ps = dict(norm1_scale=b.norm1.weight if ln else b.norm1.scale, in_proj_weight=b.attn.mha.in_proj_weight,
                      in_proj_bias=b.attn.mha.in_proj_bias, out_proj_weight=b.attn.mha.out_proj.weight,
                      out_proj_bias=b.attn.mha.out_proj.bias, norm2_scale=b.norm2.weight if ln else b.norm2.scale,
                      fc1_weight=b.mlp.fc1.weight, fc1_bias=b.mlp.fc1.bias,
                      fc2_weight=b.mlp.fc2.weight, fc2_bias=b.mlp.fc2.bias)
            if ln:
                ps.update(norm1_bias=b.norm1.bias, norm2_bias=b.norm2.bias)
            for k, p in ps.items():
                t = L.dev_f32(p.detach(), k)
                if t.device != dev:
                    raise L.AvdError("all MMDiT parameters must live on one device")
                keep.append(t)
                setattr(arr[i], k, t.data_ptr())
            if self.matmul not in L.MATMUL_TERMS:
                raise ValueError(f"matmul must be one of {sorted(L.MATMUL_TERMS)}, got {self.matmul!r}")
            if self.fold_norms and not ln:      # every mode: below the split kernels' row threshold all of them run the folded fp32 path
                for k, sc in (("in_proj_weight", "norm1_scale"), ("fc1_weight", "norm2_scale")):
                    t = self._folded_weight(f"{i}.{k}", L.dev_f32(ps[k].detach(), k), L.dev_f32(ps[sc].detach(), sc))
                    keep.append(t)
                    setattr(arr[i], k + "_n", t.data_ptr())
            if self.matmul != "f32" and not ln:
                sc = self._f16x2_scales(i, ps) if self.matmul == "f16x2" else None
                for j, k in enumerate(("in_proj_weight", "out_proj_weight", "fc1_weight", "fc2_weight")):
                    img = self._split3_image(f"{i}.{k}", ps[k], sc[j] if sc else 0.0)
                    keep.append(img)
                    setattr(arr[i], k + "3", img.data_ptr())
                if sc:
                    for j in range(8):
                        arr[i].f16x2_scale[j] = sc[j]
                elif self.fold_norms:
                    # bf16 plane modes: images of W * norm.scale[None, :] let the composite fold norm1 / norm2 into the neighbouring
                    # epilogues (csrc/composite.hip core_use_split_fold)
                    for k, scn in (("in_proj_weight", "norm1_scale"), ("fc1_weight", "norm2_scale")):
                        wn = self._folded_weight(f"{i}.{k}", L.dev_f32(ps[k].detach(), k), L.dev_f32(ps[scn].detach(), scn))
                        img = self._split3_image(f"{i}.{k}.n", wn)
                        keep.append(img)
                        setattr(arr[i], k + "3n", img.data_ptr())
        ln = isinstance(self.final_norm, LayerNorm)
        fin = L.dev_f32((self.final_norm.weight if ln else self.final_norm.scale).detach(), "final_norm.scale")
        keep.append(fin)
        fin_b = None
        if ln:
            fin_b = L.dev_f32(self.final_norm.bias.detach(), "final_norm.bias")
            keep.append(fin_b)
        hidden = self.blocks[0].mlp.fc1.weight.shape[0]
        cw = L.CoreWeights(self.cfg.d_model, len(self.blocks), self.cfg.n_heads, hidden, self.final_norm.eps,
                           C.cast(arr, C.POINTER(L.BlockWeights)), fin.data_ptr(), 1 if ln else 0, L.ptr(fin_b),
                           L.MATMUL_TERMS[self.matmul], self._attn_code())
        return cw, (arr, keep)

    def forward(self, x: torch.Tensor, key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B,N,d]; key_padding_mask [B,N] bool, True = padding key (mmdt.py:134-149)."""
        if self.training and (self.cfg.token_dropout > 0 or self.cfg.dropout > 0 or self.cfg.attn_dropout > 0):
            raise NotImplementedError("HIP path is inference-only; call .eval()")
        x = L.dev_f32(x, "x")
        B, N, d = x.shape
        if d != self.cfg.d_model:
            raise RuntimeError(f"expected last dim {self.cfg.d_model}, got {d}")
        kpm = Fn.key_padding_bytes(key_padding_mask, B, N, x.device)
        cw, keep = self.weight_table()
        need = L.lib().avd_core_workspace_bytes(C.byref(cw), B, N)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        L.check(L.lib().avd_core_forward_f32(C.byref(cw), x.data_ptr(), y.data_ptr(), B, N, 0, N, L.ptr(kpm),
                                             self._ws.data_ptr(), self._ws.numel(), L.stream_ptr(x.device)))
        del keep
        return y
