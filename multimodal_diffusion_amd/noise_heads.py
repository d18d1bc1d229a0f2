"""ε-prediction head — host-side mirror of ``avdiff/models/heads/noise_heads.py:94-229`` over the HIP C ABI.

``MultiModalNoiseHead`` keeps the reference's constructor, ``forward(dict) -> dict`` contract and
``state_dict`` keys (``input_proj.{m}``, ``shared.{j}.0`` Linear / ``shared.{j}.1`` LayerNorm, ``spec.{m}``,
``out_proj.{m}``).  torch.nn.Linear / LayerNorm objects are used purely as parameter containers — their
``forward`` is never called; each modality path is one ``avd_head_forward_f32`` call.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Union

import torch
import torch.nn as nn

from . import _lib as L

__all__ = ["MultiModalNoiseHead"]

_ACTS = {"gelu": L.ACT_GELU, "relu": L.ACT_RELU, "leaky_relu": L.ACT_LEAKY_RELU}      # noise_heads.py:28-36 (LeakyReLU slope 0.1)


def _act_code(name: str) -> int:
    name = (name or "gelu").lower()
    if name in _ACTS:
        return _ACTS[name]
    raise ValueError(f"Unsupported activation: {name}")


class _Marker(nn.Module):
    """Parameter-free placeholder keeping Sequential indices (activation / dropout slots) aligned."""


def _trunk_block(hidden: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(hidden, hidden), nn.LayerNorm(hidden), _Marker(), _Marker())


def _init_linear(m: nn.Module):
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.zeros_(m.bias)


class MultiModalNoiseHead(nn.Module):
    def __init__(self, input_dims: Dict[str, int], output_dims: Dict[str, int], hidden_dim: int = 512,
                 num_shared_layers: int = 2, num_modality_specific_layers: int = 1, dropout: float = 0.1,
                 activation: str = "gelu", share_parameters: bool = False):
        super().__init__()
        self.modalities = list(input_dims.keys())
        self.input_dims = {k: int(v) for k, v in input_dims.items()}
        self.output_dims = {k: int(v) for k, v in output_dims.items()}
        self.hidden_dim = int(hidden_dim)
        self.share_parameters = bool(share_parameters)
        self.dropout = float(dropout)
        self._act = _act_code(activation)

        self.input_proj = nn.ModuleDict({m: nn.Linear(self.input_dims[m], self.hidden_dim) for m in self.modalities})
        self.shared = None if num_shared_layers <= 0 else \
            nn.Sequential(*[_trunk_block(self.hidden_dim) for _ in range(num_shared_layers)])
        n_spec = max(0, num_modality_specific_layers - 1)
        if num_modality_specific_layers <= 0 or self.share_parameters:
            if num_modality_specific_layers > 0:
                self.shared_specific_trunk = nn.Sequential(*[_trunk_block(self.hidden_dim) for _ in range(n_spec)]) \
                    if n_spec > 0 else nn.Identity()
            self.spec = nn.ModuleDict({m: nn.Identity() for m in self.modalities})
        else:
            self.spec = nn.ModuleDict({
                m: nn.Sequential(*[_trunk_block(self.hidden_dim) for _ in range(n_spec)]) if n_spec > 0 else nn.Identity()
                for m in self.modalities})
        self.out_proj = nn.ModuleDict({m: nn.Linear(self.hidden_dim, self.output_dims[m]) for m in self.modalities})
        self.apply(_init_linear)
        self._ws: Optional[torch.Tensor] = None
        # matrix-pipe mode of the Linears, as MMDiT.matmul ("auto" | "f32" | "bf16x3" | "bf16x3_strict" | "bf16" | "f16x2"); shapes the
        # split kernels do not cover (d_out % 256, fewer than 6,144 rows) stay on the fp32 MFMA kernels
        self.matmul = "auto"
        self._images: dict = {}

    def get_output_dim(self, modality: str) -> int:
        return int(self.output_dims[modality])

    # ---- Linear→LayerNorm→act blocks a modality's rows pass through, in order ----
    def _trunk(self, m: str) -> List[nn.Sequential]:
        blocks: List[nn.Sequential] = list(self.shared) if self.shared is not None else []
        if self.share_parameters and hasattr(self, "shared_specific_trunk"):
            if not isinstance(self.shared_specific_trunk, nn.Identity):
                blocks += list(self.shared_specific_trunk)
        elif not isinstance(self.spec[m], nn.Identity):
            blocks += list(self.spec[m])
        return blocks

    def _image(self, name: str, p: torch.Tensor, h2_scale: float) -> torch.Tensor:
        """Operand image of a weight for the split matmul modes (cached per parameter version, refreshed in place)."""
        from . import functional as Fn
        key = (p.data_ptr(), p._version, tuple(p.shape), float(h2_scale))
        hit = self._images.get(name)
        if hit is None or hit[0] != key:
            old = hit[1] if hit is not None and hit[0][2] == key[2] and hit[1].device == p.device else None
            img = Fn.split_f16x2(p.detach(), h2_scale, out=old)[0] if h2_scale > 0 else Fn.split3(p.detach(), out=old)
            hit = (key, img)
            self._images[name] = hit
        return hit[1]

    def _f16x2_scales(self, m: str, blocks, in_bound: float, in_norm: float):
        """Power-of-two scales of the f16x2 images of modality m's path, from bounds that hold for every input row x with
        |x_i| <= in_bound, ||x||_2 <= in_norm:  input_proj output  |o_n| <= ||x|| ||w_n|| + |b_n|  (Cauchy-Schwarz);
        LayerNorm output  |y_i| <= sqrt(hidden) |gamma_i| + |beta_i|  (every activation here satisfies |act(y)| <= |y|)."""
        from . import functional as Fn
        ps = [self.input_proj[m].weight, self.input_proj[m].bias] + [q for b in blocks for q in (b[0].weight, b[1].weight, b[1].bias)] + \
            [self.out_proj[m].weight]
        key = (tuple((q.data_ptr(), q._version) for q in ps), float(in_bound), float(in_norm))
        hit = self._images.get(f"{m}.f16x2")
        if hit is None or hit[0] != key:
            wi, bi = self.input_proj[m].weight, self.input_proj[m].bias
            ts = [wi, bi] + [t for b in blocks for t in (b[0].weight, b[1].weight, b[1].bias)] + [self.out_proj[m].weight]
            bd = Fn.weight_bounds(ts)                      # (max |w|, max row norm) per tensor, one sync
            q = [bd[0][0], bd[0][1], bd[1][0]] + [bd[2 + j][0] for j in range(3 * len(blocks))] + [bd[-1][0]]
            w_in, wn_in, b_in = q[0], q[1], q[2]
            w_sc = [Fn.f16x2_scale(w_in)] + [Fn.f16x2_scale(q[3 + 3 * j]) for j in range(len(blocks))] + [Fn.f16x2_scale(q[-1])]
            a_sc = [Fn.f16x2_scale(in_bound), Fn.f16x2_scale(in_norm * wn_in + b_in)]
            a_sc += [Fn.f16x2_scale(self.hidden_dim ** 0.5 * q[4 + 3 * j] + q[5 + 3 * j]) for j in range(len(blocks))]
            hit = (key, w_sc + a_sc)
            self._images[f"{m}.f16x2"] = hit
        return hit[1]

    def weight_table(self, m: str, matmul: Optional[str] = None, in_bound: Optional[float] = None, in_norm: Optional[float] = None):
        """Pointer table of modality m's path.  matmul (default: self.matmul) other than "f32" adds the operand images of the split
        matmul modes; "f16x2" also needs bounds on the rows the head will see (in_bound >= |x_i|, in_norm >= ||x||_2) — the
        engine derives them from the core's final norm."""
        matmul = self.matmul if matmul is None else matmul
        if matmul not in L.MATMUL_TERMS:
            raise ValueError(f"matmul must be one of {sorted(L.MATMUL_TERMS)}, got {matmul!r}")
        blocks = self._trunk(m)
        n = len(blocks)
        keep = []

        def dp(p):
            t = L.dev_f32(p.detach(), "head parameter")
            keep.append(t)
            return t.data_ptr()

        arrs = [(C.c_void_p * max(n, 1))() for _ in range(5)]
        for j, blk in enumerate(blocks):
            arrs[0][j], arrs[1][j] = dp(blk[0].weight), dp(blk[0].bias)
            arrs[2][j], arrs[3][j] = dp(blk[1].weight), dp(blk[1].bias)
        eps = blocks[0][1].eps if n else 1e-5
        hw = L.HeadWeights(self.input_dims[m], self.hidden_dim, self.output_dims[m], n, eps, self._act,
                           dp(self.input_proj[m].weight), dp(self.input_proj[m].bias),
                           C.cast(arrs[0], C.POINTER(C.c_void_p)), C.cast(arrs[1], C.POINTER(C.c_void_p)),
                           C.cast(arrs[2], C.POINTER(C.c_void_p)), C.cast(arrs[3], C.POINTER(C.c_void_p)),
                           dp(self.out_proj[m].weight), dp(self.out_proj[m].bias))
        terms = L.MATMUL_TERMS[matmul]
        shapes_ok = self.input_dims[m] % 16 == 0 and self.hidden_dim % 256 == 0 and self.output_dims[m] % 256 == 0
        if terms and shapes_ok:
            sc = None
            if matmul == "f16x2":
                if in_bound is None or in_norm is None:
                    raise ValueError("matmul='f16x2' needs in_bound / in_norm (bounds on the head's input rows)")
                sc = self._f16x2_scales(m, blocks, in_bound, in_norm)
                scales = (C.c_float * len(sc))(*sc)
                arrs.append(scales)                 # host array: kept alive with the table, not a device tensor
                hw.f16x2_scale = C.cast(scales, C.POINTER(C.c_float))
            ws = [self.input_proj[m].weight] + [b[0].weight for b in blocks] + [self.out_proj[m].weight]
            imgs = [self._image(f"{m}.{i}", w, sc[i] if sc else 0.0) for i, w in enumerate(ws)]
            keep += imgs
            hw.split_terms = terms
            hw.input_proj_weight3 = imgs[0].data_ptr()
            for j in range(n):
                arrs[4][j] = imgs[1 + j].data_ptr()
            hw.shared_lin_weight3 = C.cast(arrs[4], C.POINTER(C.c_void_p))
            hw.out_proj_weight3 = imgs[-1].data_ptr()
        return hw, (arrs, keep)

    def forward(self, inputs: Dict[str, torch.Tensor], return_dict: bool = True
                ) -> Union[Dict[str, torch.Tensor], torch.Tensor]:
        if self.training and self.dropout > 0:
            raise NotImplementedError("HIP path is inference-only; call .eval()")
        outputs: Dict[str, torch.Tensor] = {}
        for m in self.modalities:
            if m not in inputs or inputs[m] is None:
                continue
            x = L.dev_f32(inputs[m], f"inputs[{m}]")
            d_in = x.shape[-1]
            if d_in != self.input_dims[m]:
                raise RuntimeError(f"{m}: expected last dim {self.input_dims[m]}, got {d_in}")
            rows = x.numel() // d_in
            if self.matmul == "f16x2":      # standalone call: bound the rows from the data (one device sync)
                from . import functional as Fn
                amax = Fn.weight_bounds([x.reshape(rows, d_in)])[0][0]
                hw, keep = self.weight_table(m, in_bound=amax, in_norm=amax * d_in ** 0.5)
            else:
                hw, keep = self.weight_table(m)
            need = L.lib().avd_head_workspace_bytes(C.byref(hw), rows)
            if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
                self._ws = torch.empty(max(need, 1), dtype=torch.uint8, device=x.device)
            out = torch.empty(*x.shape[:-1], self.output_dims[m], device=x.device, dtype=torch.float32)
            L.check(L.lib().avd_head_forward_f32(C.byref(hw), x.data_ptr(), d_in, 0, 0, rows, out.data_ptr(),
                                                 self._ws.data_ptr(), self._ws.numel(), L.stream_ptr(x.device)))
            del keep
            outputs[m] = out
        if return_dict:
            return outputs
        for m in self.modalities:
            if m in outputs:
                return outputs[m]
        raise ValueError("No modalities found in inputs for MultiModalNoiseHead.forward")
