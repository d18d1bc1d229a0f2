"""AudioCodec — host-side mirror of ``avdiff/models/encoders/audio_codec.py`` (loop boundary).

Same config dataclass / ``from_config`` / ``state_dict`` keys (``pre.{0,1}.0``, ``to_lat``, ``from_lat``,
``smooth.{0,2,4}``).  ``encode`` (:184-199) = conv k9 → GELU → conv k9 → GELU → exact-frame average pool → 1x1;
``decode`` (:200-214) = 1x1 → nearest upsample x hop → conv k → GELU → conv k → GELU → conv k → tanh.  Every layer is an
``avd_conv1d_act_f32`` / ``avd_avgpool_frames_f32`` launch; the upsample is folded into the first smoothing conv.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L


@dataclass
class AudioCodecConfig:
    in_ch: int = 1
    lat_ch: int = 8
    sr: int = 16000
    hop_samples: int = 320
    hidden: int = 64
    smooth_kernel: int = 7
    frames_per_clip: Optional[int] = None

    @staticmethod
    def from_dict(d: Dict) -> "AudioCodecConfig":
        lat, codec = d.get("latent", {}), d.get("codec", {})
        sr = int(d.get("sr", 16000))
        if "frame_hop_ms" in lat:                       # preferred: hop in milliseconds
            hop = max(1, int(round(sr * float(lat["frame_hop_ms"]) / 1000.0)))
        else:
            hop = int(codec.get("hop_samples", 320))
        return AudioCodecConfig(in_ch=int(d.get("in_ch", 1)), lat_ch=int(lat.get("channels", 8)), sr=sr, hop_samples=hop,
                                hidden=int(codec.get("hidden", 64)), smooth_kernel=int(codec.get("smooth_kernel", 7)),
                                frames_per_clip=int(lat.get("frames_per_clip", 0)) or None)


def _block(c_in: int, c_out: int, k: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv1d(c_in, c_out, kernel_size=k, padding=k // 2), nn.GELU())   # containers only


def _conv1d(x: torch.Tensor, conv: nn.Conv1d, act: int, upsample: int = 1) -> torch.Tensor:
    x = L.dev_f32(x, "x")
    B, Cin, Lin = x.shape
    w = L.dev_f32(conv.weight.detach(), "conv weight")
    Cout, Cw, k = w.shape
    if Cw != Cin:
        raise RuntimeError(f"conv1d expects {Cw} input channels, got {Cin}")
    b = None if conv.bias is None else L.dev_f32(conv.bias.detach(), "conv bias")
    out = torch.empty(B, Cout, Lin * upsample, device=x.device, dtype=torch.float32)
    L.check(L.lib().avd_conv1d_act_f32(x.data_ptr(), w.data_ptr(), L.ptr(b), out.data_ptr(), B, Cin, Cout, Lin, upsample, k,
                                       act, L.stream_ptr(x.device)))
    return out


class AudioCodec(nn.Module):
    def __init__(self, cfg: AudioCodecConfig):
        super().__init__()
        self.cfg = cfg
        k = max(3, int(cfg.smooth_kernel))
        self.pre = nn.Sequential(_block(cfg.in_ch, cfg.hidden, 9), _block(cfg.hidden, cfg.hidden, 9))
        self.to_lat = nn.Conv1d(cfg.hidden, cfg.lat_ch, kernel_size=1)
        self.from_lat = nn.Conv1d(cfg.lat_ch, cfg.hidden, kernel_size=1)
        pad = k // 2
        self.smooth = nn.Sequential(nn.Conv1d(cfg.hidden, cfg.hidden, kernel_size=k, padding=pad), nn.GELU(),
                                    nn.Conv1d(cfg.hidden, cfg.hidden, kernel_size=k, padding=pad), nn.GELU(),
                                    nn.Conv1d(cfg.hidden, cfg.in_ch, kernel_size=k, padding=pad))
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                nn.init.kaiming_uniform_(m.weight, a=0.2)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    @classmethod
    def from_config(cls, d: Dict) -> "AudioCodec":
        return cls(AudioCodecConfig.from_dict(d))

    @property
    def hop(self) -> int:
        return int(self.cfg.hop_samples)

    @staticmethod
    def _compute_exact_pool_params(L_: int, Fa: int) -> Tuple[int, int]:
        """Integer hop with Fa*hop >= L (audio_codec.py:143-156)."""
        assert Fa > 0
        hop = max(1, int(round(L_ / Fa)))
        total = Fa * hop
        if total < L_:
            hop += 1
            total = Fa * hop
        return hop, total

    def _avgpool_frames(self, x: torch.Tensor, target_Fa: Optional[int] = None) -> torch.Tensor:
        B, H, L_ = x.shape
        if target_Fa is None:
            hop = self.hop
            Fa = math.ceil(L_ / hop)
        else:
            Fa = int(target_Fa)
            hop, _ = self._compute_exact_pool_params(L_, Fa)
        out = torch.empty(B, H, Fa, device=x.device, dtype=torch.float32)
        L.check(L.lib().avd_avgpool_frames_f32(L.dev_f32(x).data_ptr(), out.data_ptr(), B * H, L_, Fa, hop,
                                               L.stream_ptr(x.device)))
        return out

    @torch.no_grad()
    def encode(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [B,1,L] -> z [B,Ca,Fa]."""
        assert wav.dim() == 3 and wav.size(1) == 1, "AudioCodec.encode expects [B,1,L]"
        h = _conv1d(wav, self.pre[0][0], L.ACT_GELU)
        h = _conv1d(h, self.pre[1][0], L.ACT_GELU)
        h = self._avgpool_frames(h, target_Fa=self.cfg.frames_per_clip)
        return _conv1d(h, self.to_lat, L.ACT_NONE)

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B,Ca,Fa] -> wav [B,1,Fa*hop] in [-1,1]."""
        assert z.dim() == 3, "AudioCodec.decode expects [B,Ca,Fa]"
        h = _conv1d(z, self.from_lat, L.ACT_NONE)
        h = _conv1d(h, self.smooth[0], L.ACT_GELU, upsample=self.hop)      # nearest x hop folded into the indexing
        h = _conv1d(h, self.smooth[2], L.ACT_GELU)
        return _conv1d(h, self.smooth[4], L.ACT_TANH)
