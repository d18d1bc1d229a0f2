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

    def weight_table(self, m: str):
        blocks = self._trunk(m)
        n = len(blocks)
        keep = []

        def dp(p):
            t = L.dev_f32(p.detach(), "head parameter")
            keep.append(t)
            return t.data_ptr()

        arrs = [(C.c_void_p * max(n, 1))() for _ in range(4)]
        for j, blk in enumerate(blocks):
            arrs[0][j], arrs[1][j] = dp(blk[0].weight), dp(blk[0].bias)
            arrs[2][j], arrs[3][j] = dp(blk[1].weight), dp(blk[1].bias)
        eps = blocks[0][1].eps if n else 1e-5
        hw = L.HeadWeights(self.input_dims[m], self.hidden_dim, self.output_dims[m], n, eps, self._act,
                           dp(self.input_proj[m].weight), dp(self.input_proj[m].bias),
                           C.cast(arrs[0], C.POINTER(C.c_void_p)), C.cast(arrs[1], C.POINTER(C.c_void_p)),
                           C.cast(arrs[2], C.POINTER(C.c_void_p)), C.cast(arrs[3], C.POINTER(C.c_void_p)),
                           dp(self.out_proj[m].weight), dp(self.out_proj[m].bias))
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
