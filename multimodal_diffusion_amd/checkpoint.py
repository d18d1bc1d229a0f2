"""Checkpoint interop.

Two on-disk layouts exist in the reference and they do not agree with each other:
  * the sampler's loader looks for ``{name}_state_dict`` entries (``sample_clip.py:112-132``);
  * the trainer writes ``{"step", "core", "head", "adapt_v", "adapt_a", "vid_vae", "aud_codec", "opt", "ema"}``
    (``train/trainer.py:407-423``), with adapters of width d that *add* a d-wide timestep embedding
    (``trainer.py:45-49,195-202``) where the sampler *concatenates* a 256-wide one onto d-256 adapters.
Both load here.  Files are opened with ``torch.load(..., weights_only=True)`` only — nothing in a checkpoint is
executed.  ``load_trainer_checkpoint`` reports which embedding mode the adapters were trained for, so the caller
can build ``DenoiseEngine(temb_mode=...)`` to match.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Mapping, Optional, Union

import torch

_NAMES = ("vid_vae", "aud_codec", "adapt_v", "adapt_a", "core", "head")


def _read(path_or_state: Union[str, Path, Mapping]) -> Mapping:
    if isinstance(path_or_state, Mapping):
        return path_or_state
    path = Path(path_or_state)
    if not path.exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    return torch.load(path, map_location="cpu", weights_only=True)


def _strip_ddp(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_checkpoint_maybe(cfg: Dict, modules: Dict[str, torch.nn.Module]) -> None:
    """``sample_clip.load_checkpoint_maybe`` (:112-132): ``paths.ckpt_path`` / ``paths.ckpt`` → ``{name}_state_dict``
    entries, ``strict=False``; no path → random weights (the reference prints a notice); missing file → FileNotFoundError."""
    ckpt_path = cfg.get("paths", {}).get("ckpt_path") or cfg.get("paths", {}).get("ckpt")
    if not ckpt_path:
        print("[info] no ckpt_path provided in config; sampling with random weights.")
        return
    state = _read(ckpt_path)
    for name, mod in modules.items():
        key = f"{name}_state_dict"
        if isinstance(state, Mapping) and key in state:
            missing, unexpected = mod.load_state_dict(_strip_ddp(state[key]), strict=False)
            print(f"[ckpt] loaded {name} (missing={len(missing)} unexpected={len(unexpected)})")


def load_trainer_checkpoint(path_or_state: Union[str, Path, Mapping], modules: Dict[str, Optional[torch.nn.Module]],
                            use_ema: bool = False, strict: bool = True) -> Dict[str, object]:
    """Load a ``AVTrainer.save_checkpoint`` dict into the mirrors.

    ``modules`` maps any of vid_vae / aud_codec / adapt_v / adapt_a / core / head to the module to fill (None skips).
    ``use_ema`` takes the core weights from ``state["ema"]`` (the trainer keeps an EMA of the core only).
    Returns {"step", "loaded": [...], "temb_mode": "add" | "concat" | None} — the embedding mode implied by the
    adapter width against the core width.
    """
    state = _read(path_or_state)
    loaded = []
    for name in _NAMES:
        mod = modules.get(name)
        if mod is None or name not in state:
            continue
        sd = state["ema"] if (name == "core" and use_ema and "ema" in state) else state[name]
        mod.load_state_dict(_strip_ddp(sd), strict=strict)
        loaded.append(name)
    mode = None
    core, av = modules.get("core"), modules.get("adapt_v")
    if core is not None and av is not None:
        d = core.cfg.d_model
        w = av.proj.weight.shape[0]
        mode = "add" if w == d else "concat" if w < d else None
    return {"step": int(state.get("step", 0)) if isinstance(state, Mapping) else 0, "loaded": loaded, "temb_mode": mode}
