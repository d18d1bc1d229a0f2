"""Sliding-window long-form generation — mirror of ``avdiff/models/infer/stream_infer.py``.

The reference splits the prompt into windows (3 s window / 1 s hop), calls ``sample_one_direction`` once per window
(B = 1, sequentially) and cross-fades the results on the host.  Windows are independent samples, so here they are one
batch: the prompt windows are encoded together, ``DenoiseEngine`` steps all windows at once (they are the natural
large-batch feed for the data-parallel path), the outputs are decoded together and stitched by a HIP kernel.
With ``shard=True`` under ``torch.distributed`` the windows are the units of the data-parallel layout: rank 0 encodes the
prompt windows, ONE broadcast hands them to every rank, each rank steps AND DECODES its contiguous share of the windows with no
further communication (the reference ends every window's call with its own decode, sample_clip.py:392 via stream_infer.py:182-188,
207-213), one gather brings the decoded windows — uint8 frames or waveforms — to rank 0, which only stitches.

``split_*`` are host-side slicing (as in the reference); the fade tables are built on the host with the reference's
fp32 numpy expressions and uploaded; ``crossfade_*`` keep the reference's numpy-in / numpy-out signatures.
File I/O and the CLI of the reference script are out of scope.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib as L
from . import dist as D
from . import schedule_utils as su
from .sampler import DenoiseEngine


def split_audio_into_windows(y: np.ndarray, sr: int, win_s: float, hop_s: float) -> Tuple[np.ndarray, int, int]:
    """[L] -> [N, win] (last window zero-padded), stream_infer.py:40-58."""
    n = len(y)
    win, hop = int(round(sr * win_s)), int(round(sr * hop_s))
    if n <= win:
        return y[None, :], win, hop
    starts = list(range(0, n, hop))
    out = []
    for s in starts:
        seg = y[s:min(n, s + win)]
        out.append(np.pad(seg, (0, win - len(seg))) if len(seg) < win else seg)
        if s + win >= n:
            break
    return np.stack(out, axis=0), win, hop


def split_frames_into_windows(frames: np.ndarray, fps: int, win_s: float, hop_s: float) -> Tuple[np.ndarray, int, int]:
    """[T,H,W,3] -> [N, win, H, W, 3] (last window padded by repeating its last frame), stream_infer.py:61-82."""
    n = frames.shape[0]
    win, hop = int(round(fps * win_s)), int(round(fps * hop_s))
    if n <= win:
        return frames[None, ...], win, hop
    out = []
    for s in range(0, n, hop):
        seg = frames[s:min(n, s + win)]
        if seg.shape[0] < win:
            seg = np.concatenate([seg, np.repeat(seg[-1:], win - seg.shape[0], axis=0)], axis=0)
        out.append(seg)
        if s + win >= n:
            break
    return np.stack(out, axis=0), win, hop


def audio_fade_window(L_: int, fade: int) -> np.ndarray:
    """Cosine fade-in / fade-out table (stream_infer.py:103-106); all ones for fade <= 0."""
    w = np.ones(L_, dtype=np.float32)
    if fade > 0:
        w[:fade] = 0.5 * (1 - np.cos(np.linspace(0, np.pi, fade, dtype=np.float32)))
        w[-fade:] = 0.5 * (1 + np.cos(np.linspace(0, np.pi, fade, dtype=np.float32)))
    return w


def video_fade_window(L_: int, fade: int) -> np.ndarray:
    """Triangular ramp table in frames (stream_infer.py:130-137)."""
    w = np.ones(L_, dtype=np.float32)
    if fade > 0:
        ramp = np.linspace(0, 1, fade, dtype=np.float32)
        w[:fade] *= ramp
        w[-fade:] *= ramp[::-1]
    return w


def crossfade_tensor(chunks: torch.Tensor, w: torch.Tensor, hop: int) -> torch.Tensor:
    """chunks [N, L, ...] (float32 or uint8, on the device), w [L] -> [(N-1)*hop + L, ...]."""
    if not chunks.is_cuda:
        raise L.AvdError("crossfade needs ROCm device tensors (no CPU fallback)")
    chunks = chunks.contiguous()
    N, L_ = chunks.shape[:2]
    inner = int(np.prod(chunks.shape[2:])) if chunks.dim() > 2 else 1
    w = w.to(chunks.device, torch.float32).contiguous()
    out = torch.empty((N - 1) * hop + L_, *chunks.shape[2:], device=chunks.device, dtype=chunks.dtype)
    fn = L.lib().avd_crossfade_u8 if chunks.dtype == torch.uint8 else L.lib().avd_crossfade_f32
    if chunks.dtype not in (torch.uint8, torch.float32):
        raise TypeError("crossfade takes float32 or uint8 chunks")
    L.check(fn(chunks.data_ptr(), w.data_ptr(), out.data_ptr(), N, L_, hop, inner, L.stream_ptr(chunks.device)))
    return out


def crossfade_audio(chunks: np.ndarray, sr: int, hop: int, win: int, fade_s: float, device="cuda") -> np.ndarray:
    """[N, L] float32 -> stitched [ (N-1)*hop + L ] (stream_infer.py:85-116)."""
    fade = int(round(sr * fade_s))
    w = torch.from_numpy(audio_fade_window(chunks.shape[1], fade))
    out = crossfade_tensor(torch.from_numpy(np.ascontiguousarray(chunks, dtype=np.float32)).to(device), w, hop)
    return out.cpu().numpy()


def crossfade_video(chunks: np.ndarray, hop: int, win: int, fade_f: int, device="cuda") -> np.ndarray:
    """[N, T, H, W, 3] uint8 -> stitched [T_total, H, W, 3] uint8 (stream_infer.py:119-143)."""
    w = torch.from_numpy(video_fade_window(chunks.shape[1], int(fade_f)))
    out = crossfade_tensor(torch.from_numpy(np.ascontiguousarray(chunks)).to(device), w, hop)
    return out.cpu().numpy()


@torch.no_grad()
def stream_generate(*, cfg: Dict, vid_vae, aud_codec, adapt_v, adapt_a, core, head, tstep_dim: int, prompt_modality: str,
                    prompt_video: Optional[np.ndarray], prompt_audio: Optional[np.ndarray], device: torch.device,
                    init_noise: Optional[torch.Tensor] = None, max_windows_per_batch: int = 32, shard: bool = False,
                    seed: Optional[int] = None, comm_device: Optional[torch.device] = None) -> Optional[Dict[str, np.ndarray]]:
    """The body of the reference's ``main()`` (stream_infer.py:146-225) minus file I/O, with all windows batched.

    Returns {"audio": wav, "sr"} for a video prompt or {"video": frames uint8, "fps"} for an audio prompt.
    ``init_noise`` [N_windows, *latent] fixes the initial latents (the reference draws them window by window); ``seed`` draws them
    from a CPU generator for the WHOLE window list instead (the same numbers whatever the number of ranks).

    ``shard=True`` (every rank of the default process group calls this with the same arguments; one process per GPU): rank 0 encodes
    the prompt windows, ONE broadcast (``dist.broadcast_conditioning``; over ``comm_device``, default: ``device`` for the nccl = RCCL
    backend, the CPU for gloo) hands them to all ranks, rank r steps windows ``dist.shard_range(N_windows, r, world)`` and decodes them
    (``vid_vae.decode`` -> uint8 frames / ``aud_codec.decode`` -> waveform: per rank N_windows / world loops AND decodes — the decode is
    8 ms per 256 x 256 window against 14 ms for its 50-step loop, so leaving it on one rank would cap an 8-rank run near 2x), one
    gather brings the decoded windows to rank 0, which stitches and returns the result — the other ranks return None.
    If rank 0 fails before the broadcast (encode error, a prompt shape the config does not imply), every rank raises.
    Windows never interact inside the loop or the decode, so the stitched output equals the single-process one bit for bit as long as
    both runs take the same kernels (the matrix-pipe mode "auto" switches kernels at 2,048 / 6,144 rows: fix ``core.matmul`` to
    compare across sizes).
    """
    st = cfg.get("streaming", {})
    win_s, hop_s = float(st.get("window_seconds", 3.0)), float(st.get("hop_seconds", 1.0))
    xfade_s = float(st.get("crossfade_seconds", 0.25))
    fps, sr = int(cfg["video"]["fps"]), int(cfg["audio"]["sr"])
    t_p, p = int(cfg["tokenizer"]["video"]["tube"]["t"]), int(cfg["tokenizer"]["video"]["tube"]["h"])
    l_chunk, s_chunk = int(cfg["tokenizer"]["audio"]["chunk"]["length"]), int(cfg["tokenizer"]["audio"]["chunk"]["stride"])
    Cv, t_down, s_down = (int(cfg["video"]["latent"][k]) for k in ("channels", "t_down", "s_down"))
    Ca, Fa = int(cfg["audio"]["latent"]["channels"]), int(cfg["audio"]["latent"]["frames_per_clip"])
    H, W = int(cfg["video"]["size"][0]), int(cfg["video"]["size"][1])
    eta = float(cfg["sampling"].get("ddim_eta", 0.0))

    import torch.distributed as tdist
    world = tdist.get_world_size() if (shard and tdist.is_initialized()) else 1
    rank = tdist.get_rank() if world > 1 else 0
    root = rank == 0

    # the prompt windows and the shape of their encoded form (known on every rank without encoding: only rank 0 runs the encoder)
    if prompt_modality == "video":
        if prompt_video is None:
            raise ValueError("prompt_video frames required for prompt_modality=video")
        chunks, win, hop = split_frames_into_windows(prompt_video, fps=fps, win_s=win_s, hop_s=hop_s)
        zp_shape = (chunks.shape[0], Cv, chunks.shape[1] // t_down, chunks.shape[2] // s_down, chunks.shape[3] // s_down)
        target, lat = "audio", (Ca, Fa)
        n_prompt = (zp_shape[2] // t_p) * (zp_shape[3] // p) * (zp_shape[4] // p)
        guide = float(cfg["sampling"]["guidance_scale"].get("audio", 3.0))
    elif prompt_modality == "audio":
        if prompt_audio is None:
            raise ValueError("prompt_audio required for prompt_modality=audio")
        chunks, win, hop = split_audio_into_windows(prompt_audio, sr=sr, win_s=win_s, hop_s=hop_s)
        zp_shape = (chunks.shape[0], Ca, Fa)
        T_in = int(round(cfg["data"]["clip_seconds"] * fps))
        target, lat = "video", (Cv, max(1, T_in // t_down), H // s_down, W // s_down)
        n_prompt = (Fa - l_chunk) // s_chunk + 1
        guide = float(cfg["sampling"]["guidance_scale"].get("video", 3.0))
    else:
        raise ValueError("prompt_modality must be 'video' or 'audio'")
    z_p, root_error = None, None
    if root:
        try:
            if prompt_modality == "video":
                frames = torch.from_numpy(np.ascontiguousarray(chunks)).to(device).float() / 255.0      # [N,T,H,W,3]
                z_p = vid_vae.encode(frames.permute(0, 4, 1, 2, 3).contiguous())
            else:
                z_p = aud_codec.encode(torch.from_numpy(np.ascontiguousarray(chunks, dtype=np.float32)).to(device)[:, None, :])
            if tuple(z_p.shape) != zp_shape:
                if world > 1:       # the other ranks sized their buffers from the config
                    raise L.AvdError(f"encoded prompt windows have shape {tuple(z_p.shape)}, the config implies {zp_shape}")
                # single process (e.g. a caller-supplied encoder): the encoder's own output decides, as in the reference
                zp_shape = tuple(z_p.shape)
                if prompt_modality == "video":
                    n_prompt = (zp_shape[2] // t_p) * (zp_shape[3] // p) * (zp_shape[4] // p)
                else:
                    n_prompt = (zp_shape[-1] - l_chunk) // s_chunk + 1
        except Exception as exc:            # noqa: BLE001 — re-raised below; a sharded run first tells the other ranks
            if world == 1:
                raise
            root_error = exc

    c = cfg["diffusion"][target]
    abar = su.alphas_cumprod_from_betas(su.make_beta_schedule(int(c["steps"]), kind=c["schedule"], min_beta=c["min_beta"],
                                                              max_beta=c["max_beta"]))[1]
    sched = su.make_sampling_schedule(int(c["steps"]), int(c["sampler_steps"]))
    Nw = zp_shape[0]
    if init_noise is not None:
        z0 = init_noise
    elif seed is not None:
        z0 = torch.randn(Nw, *lat, generator=torch.Generator().manual_seed(int(seed)))
    elif world > 1:
        raise ValueError("a sharded run needs init_noise or seed: every rank must start its windows from the same global draw")
    else:
        z0 = torch.randn(Nw, *lat, device=device)
    if tuple(z0.shape) != (Nw, *lat):
        raise ValueError(f"init_noise has shape {tuple(z0.shape)}, expected {(Nw, *lat)}")

    def denoise(zp_part: torch.Tensor, lo0: int, hi0: int) -> torch.Tensor:
        """the windows [lo0, hi0) of the list, stepped in batches of at most max_windows_per_batch; no communication"""
        outs = []
        for lo in range(lo0, hi0, max_windows_per_batch):
            hi = min(hi0, lo + max_windows_per_batch)
            eng = DenoiseEngine(adapt_v=adapt_v, adapt_a=adapt_a, core=core, head=head, tstep_dim=tstep_dim, target=target,
                                latent_shape=(hi - lo, *lat), prompt_tokens=n_prompt, alpha_bar=abar, guidance=guide, eta=eta,
                                tube=(t_p, p, p), chunk=(l_chunk, s_chunk))
            eng.set_prompt(zp_part[lo - lo0:hi - lo0].to(device).float().contiguous())
            outs.append(eng.run(z0[lo:hi].to(device).contiguous(), sched))
        if not outs:
            return torch.empty(0, *lat, device=device)
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]

    def decode_windows(z: torch.Tensor) -> torch.Tensor:
        """finished latents of some windows -> what the stitcher takes: waveforms [n, L] float32 or frames [n, T, H, W, 3] uint8"""
        if target == "audio":
            if z.shape[0] == 0:
                hop_a = int(getattr(aud_codec, "hop", 0)) or int(round(sr * float(cfg["data"]["clip_seconds"]))) // Fa
                return torch.empty(0, Fa * hop_a, device=device)
            return aud_codec.decode(z)[:, 0, :].contiguous()
        if z.shape[0] == 0:
            return torch.empty(0, lat[1] * t_down, lat[2] * s_down, lat[3] * s_down, 3, dtype=torch.uint8, device=device)
        x = vid_vae.decode(z).clamp(0, 1)                                         # [n,3,T,H,W]
        return (x.permute(0, 2, 3, 4, 1) * 255.0).to(torch.uint8).contiguous()    # as the reference's astype(np.uint8)

    if world > 1:
        if comm_device is None:
            comm_device = device if tdist.get_backend() == "nccl" else torch.device("cpu")
        out = D.run_sharded(Nw, z_p if root else None, zp_shape, comm_device, lambda part, lo, hi: decode_windows(denoise(part, lo, hi)),
                            result="root", error=root_error)
        if not root:
            return None
        out = out.to(device)
    else:
        out = decode_windows(denoise(z_p, 0, Nw))

    if target == "audio":
        w = torch.from_numpy(audio_fade_window(out.shape[1], int(round(sr * xfade_s))))
        return {"audio": crossfade_tensor(out, w, int(round(sr * hop_s))).cpu().numpy(), "sr": sr}
    w = torch.from_numpy(video_fade_window(out.shape[1], int(round(xfade_s * fps))))
    return {"video": crossfade_tensor(out, w, int(round(fps * hop_s))).cpu().numpy(), "fps": fps}
