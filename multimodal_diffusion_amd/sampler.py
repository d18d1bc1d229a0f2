"""DDIM + CFG sampler — host-side mirror of ``avdiff/models/infer/sample_clip.py`` over the HIP C ABI.

Keeps the reference names (``LinearAdapter``, ``add_sinusoidal_timestep``, ``build_components``,
``latents_to_tokens_*``, ``tokens_to_latents_audio``, ``sample_one_direction``) and adds what the reference
lacks: ``DenoiseEngine`` — a *batched* (B >= 1) on-device loop where one denoising step (sample_clip.py:359-389
or :318-348) is a single ``avd_denoise_step_f32`` call with the cond/null branches stacked to 2B, the constant
prompt rows embedded once, the timestep schedule resident on the device and the step optionally replayed
from a captured HIP graph.

File I/O and the CLI (sample_clip.py:112-174, 399-461) are out of scope; ``VideoVAE`` / ``AudioCodec`` are the
loop *boundary*  and are taken as caller-supplied modules.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import functional as Fn
from . import ops
from . import schedule_utils as su
from .mmdt import MMDiT
from .noise_heads import MultiModalNoiseHead


class LinearAdapter(nn.Module):
    """Per-modality linear projection to token width (sample_clip.py:48-56); PyTorch-default Linear init."""

    def __init__(self, d_in: int, d_out: int):
        super().__init__()
        self.proj = nn.Linear(d_in, d_out)     # parameter container only

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return Fn.linear(x, self.proj.weight, self.proj.bias)


def add_sinusoidal_timestep(tokens: torch.Tensor, t_scalar: torch.Tensor, dim: int) -> torch.Tensor:
    """[B,N,d] + t[B] -> [B,N,d+dim] by concatenation (sample_clip.py:59-70).  Stand-alone helper; the engine fuses
    this into the sequence-assembly kernel instead."""
    emb = su.timestep_embedding(L.dev_i64(t_scalar, tokens.device), dim)
    out = torch.empty(tokens.shape[0], tokens.shape[1], tokens.shape[2] + dim, device=tokens.device, dtype=tokens.dtype)
    out[..., :tokens.shape[2]] = tokens
    out[..., tokens.shape[2]:] = emb[:, None, :]
    return out


def build_components(cfg: Dict, device: torch.device, vid_vae: Optional[nn.Module] = None,
                     aud_codec: Optional[nn.Module] = None):
    """(vid_vae, aud_codec, adapt_v, adapt_a, core, head, tstep_dim) as sample_clip.py:75-109.

    The codec / VAE are built from ``cfg["video"]`` / ``cfg["audio"]`` like the reference unless the caller passes
    its own modules (anything with ``encode`` / ``decode``); they sit outside the per-step path.
    Extension over the reference config: ``cfg["runtime"]["matmul"]`` in {"auto", "f32", "bf16x3", "bf16x3_strict", "f16x2", "bf16"}
    selects the matrix-pipe mode of the MMDiT core, the noise head and the VAE (default "auto": the exact three-plane bf16x3 kernels
    where they engage, fp32 MFMA elsewhere — what ``bench.py`` times; same fp32-level error either way).
    """
    matmul = str(cfg.get("runtime", {}).get("matmul", "auto"))
    if matmul not in L.MATMUL_TERMS:
        raise ValueError(f"runtime.matmul must be one of {sorted(L.MATMUL_TERMS)}, got {matmul!r}")
    if vid_vae is None and "video" in cfg:
        from .vae_video3d import VideoVAE
        vid_vae = VideoVAE.from_config(cfg["video"]).to(device).eval()
        vid_vae.matmul = matmul if matmul in ("auto", "f32", "bf16x3", "f16x2") else "bf16x3"     # bf16 / strict: the exact three-plane convolutions
    if aud_codec is None and "audio" in cfg:
        from .audio_codec import AudioCodec
        aud_codec = AudioCodec.from_config(cfg["audio"]).to(device).eval()
    d = int(cfg["tokenizer"]["width"])
    out_v = int(cfg["model"]["heads"]["video"]["out_dim"])
    out_a = int(cfg["model"]["heads"]["audio"]["out_dim"])
    tstep_dim = int(cfg["embeddings"].get("timestep_dim", 256))
    adapt_v = LinearAdapter(out_v, d - tstep_dim).to(device)
    adapt_a = LinearAdapter(out_a, d - tstep_dim).to(device)
    core = MMDiT(**cfg["model"]["core"]).to(device).eval()
    core.matmul = matmul
    head = MultiModalNoiseHead(
        input_dims={"video": d, "audio": d}, output_dims={"video": out_v, "audio": out_a},
        hidden_dim=int(cfg["model"]["heads"]["video"]["hidden_dim"]), num_shared_layers=2,
        num_modality_specific_layers=1, dropout=float(cfg["model"]["core"].get("dropout", 0.1)),
        activation=cfg["model"]["heads"]["video"].get("activation", "gelu")).to(device).eval()
    head.matmul = matmul
    return vid_vae, aud_codec, adapt_v, adapt_a, core, head, tstep_dim


def latents_to_tokens_video(z_v: torch.Tensor, t_p: int, p: int) -> torch.Tensor:
    return ops.tube_patch_video(z_v, t=t_p, h=p, w=p)


def latents_to_tokens_audio(z_a: torch.Tensor, l_chunk: int, s_chunk: int) -> torch.Tensor:
    return Fn.audio_tokens(z_a, l_chunk, s_chunk)


def tokens_to_latents_audio(tokens: torch.Tensor, Ca: int, l_chunk: int, Fa: int, stride: int) -> torch.Tensor:
    """Overlap-add back to [B,Ca,Fa] (crop / zero-pad), sample_clip.py:191-215 — one kernel, no Python loops."""
    return Fn.audio_untokens(tokens, Ca, l_chunk, Fa, stride)


# ----------------------------------------------------------------------------------------------------------
# batched on-device loop
# ----------------------------------------------------------------------------------------------------------

class _CapturedPair:
    """A captured two-step HIP graph of one engine.  ``replay()`` first re-checks the engine's weight tables: parameters updated in
    place behind unchanged pointers are picked up (derived images are refreshed in place).  The graph holds every device pointer
    and every by-value scale of the tables it was captured with, so it carries the engine's table GENERATION of that moment: a
    re-allocated parameter, workspace or prompt buffer, or a changed by-value scale, moves the engine to a new generation and every
    pair captured before stays refused for good — also after the engine has captured again."""

    def __init__(self, engine: "DenoiseEngine", graph: "torch.cuda.CUDAGraph"):
        self.engine, self.graph, self.generation = engine, graph, engine._generation

    def replay(self) -> None:
        self.engine._sync_weights()
        if self.generation != self.engine._generation:
            raise L.AvdError(f"this captured graph is stale ({self.engine._stale_reason or 'the engine tables changed'} since "
                             "capture_pair()): replaying it would launch kernels that hold the old values; capture again")
        self.graph.replay()


class DenoiseEngine:
    """One direction (``target`` in {"video","audio"}) of the CFG + DDIM loop for a batch of independent samples.

    step(z, t_now, t_prev)   one step, explicit timesteps (int64 [B] on the device) — parity entry point
    run(z, sched)            whole trajectory with the schedule cursor on the device; ``graph=True`` replays a
                             captured HIP graph per step (no per-step host work beyond one graph launch)
    """

    def __init__(self, *, adapt_v: LinearAdapter, adapt_a: LinearAdapter, core: MMDiT, head: MultiModalNoiseHead,
                 tstep_dim: int, target: str, latent_shape: Tuple[int, ...], prompt_tokens: int, alpha_bar: torch.Tensor,
                 guidance: float, eta: float = 0.0, tube=(2, 4, 4), chunk=(4, 4), split_streams: Optional[bool] = None,
                 temb_mode: str = "concat", matmul: Optional[str] = None, attn: Optional[str] = None):
        if target not in ("video", "audio"):
            raise ValueError("target must be 'video' or 'audio'")
        if eta < 0:
            raise ValueError("eta must be >= 0")
        if temb_mode not in ("concat", "add"):
            raise ValueError("temb_mode must be 'concat' (sampler, sample_clip.py:59-70) or 'add' (trainer, trainer.py:45-49)")
        self.temb_mode = temb_mode
        self.target = target
        self.core, self.head = core, head
        self.adapt_t = adapt_v if target == "video" else adapt_a
        self.adapt_p = adapt_a if target == "video" else adapt_v
        self.d = core.cfg.d_model
        self.tdim = self.d if temb_mode == "add" else int(tstep_dim)     # the trainer embeds at token width
        self.tube, self.chunk = tuple(tube), tuple(chunk)
        self.guidance, self.eta = float(guidance), float(eta)
        self.device = next(core.final_norm.parameters()).device
        if not self.device.type == "cuda":
            raise L.AvdError("DenoiseEngine needs its modules on a ROCm device (no CPU fallback)")
        self.latent_shape = tuple(int(s) for s in latent_shape)      # with batch dim
        B = self.latent_shape[0]
        e = L.EmbedDesc()
        e.B, e.d, e.tdim = B, self.d, self.tdim
        if target == "video":
            _, Cc, T, H, W = self.latent_shape
            t, h, w = self.tube
            assert T % t == 0 and H % h == 0 and W % w == 0, "tube sizes must divide latent dims"
            e.target_kind, e.target_first = 0, 1                     # sequence order is always [video ; audio]
            e.C, e.T, e.H, e.W = Cc, T, H, W
            e.p0, e.p1, e.p2 = t, h, w
            e.Nt = (T // t) * (H // h) * (W // w)
        else:
            _, Ca, Fa = self.latent_shape
            ln, st = self.chunk
            e.target_kind, e.target_first = 1, 0
            e.C, e.T, e.H, e.W = Ca, Fa, 1, 1
            e.p0, e.p1, e.p2 = ln, st, 1
            e.Nt = (Fa - ln) // st + 1
        e.Np = int(prompt_tokens)
        self._freqs = Fn.temb_freqs(self.tdim, 10000, self.device) if self.tdim >= 2 else None
        e.temb_freqs = L.ptr(self._freqs)
        e.temb_add = 1 if temb_mode == "add" else 0
        w_out = self.adapt_t.proj.weight.shape[0]
        if w_out != (self.d if temb_mode == "add" else self.d - self.tdim):
            raise ValueError(f"adapter width {w_out} does not match temb_mode='{temb_mode}' (d={self.d}, tdim={self.tdim})")
        self.embed = e
        self.N = e.Nt + e.Np
        self.alpha_bar = alpha_bar.to(self.device, torch.float32).contiguous()

        # matrix-pipe mode of this engine (a key of _lib.MATMUL_TERMS; default: the core's own setting, "auto" unless changed); the core
        # module keeps its setting
        self.matmul = core.matmul if matmul is None else matmul
        self.attn = core.attn if attn is None else attn
        # cond / null halves as two kernel chains on two HIP streams (bit-identical results).  Default: on where it was measured to
        # pay — the f16x2 mode, whose kernels alternate matrix-bound loops with memory-bound epilogues, with >= 6144 rows per half
        # (C3: 166 -> 174 steps/s, 512x512: 137 -> 144); the fp32 and bf16x3 modes are matrix-bound throughout and gain nothing
        if split_streams is None:
            split_streams = self.matmul == "f16x2" and B * self.N >= 6144
        self._split_streams = bool(split_streams)
        # generation of the pointer / scalar tables: bumped whenever something a captured HIP graph holds by value changes
        # (see _CapturedPair); _stale_reason says what changed last
        self._generation = 0
        self._stale_reason = ""
        self.workspace: Optional[torch.Tensor] = None
        self.Xp: Optional[torch.Tensor] = None
        self._prompt_latent: Optional[torch.Tensor] = None
        self._bind_weights()

    # ---- pointer tables.  They hold derived copies (norm-folded / split3 weights), so they are re-derived whenever a
    # parameter's (address, version) changes: load_state_dict, EMA copy_to, an optimiser step or .to(device) after the
    # engine was built all show up here, at the next step, instead of mixing old and new weights silently.
    def _weights_key(self):
        mods = (self.core, self.head, self.adapt_t, self.adapt_p)
        return tuple((p.data_ptr(), p._version) for m in mods for p in m.parameters())

    def _bind_weights(self) -> None:
        core, head = self.core, self.head
        prev = core.matmul, core.attn
        core.matmul, core.attn = self.matmul, self.attn
        try:
            self._core_tab, self._keep_core = core.weight_table()
        finally:
            core.matmul, core.attn = prev
        # the head follows the engine's mode; its input rows are the core's final-norm output, bounded by that norm's parameters:
        # RMSNorm |y_i| <= sqrt(d) |g_i|, ||y|| <= sqrt(d) max|g|; LayerNorm adds |b_i| resp. ||b||
        hb = hn = None
        if self.matmul == "f16x2":
            fn_ = core.final_norm
            bias = getattr(fn_, "bias", None)
            g = fn_.weight if bias is not None else fn_.scale
            bd = Fn.weight_bounds([g] + ([bias] if bias is not None else []))
            gmax = bd[0][0] * core.cfg.d_model ** 0.5
            hb = gmax + (bd[1][0] if bias is not None else 0.0)
            hn = gmax + (bd[1][1] if bias is not None else 0.0)
        self._head_tab, self._keep_head = head.weight_table(self.target, matmul=self.matmul, in_bound=hb, in_norm=hn)
        self._aw = L.dev_f32(self.adapt_t.proj.weight.detach(), "adapter weight")
        self._ab = L.dev_f32(self.adapt_t.proj.bias.detach(), "adapter bias")
        if self._aw.device != self.device:
            raise L.AvdError("engine modules moved to another device after the engine was built")
        s = L.StepDesc()
        s.embed = self.embed
        s.core = C.pointer(self._core_tab)
        s.head = C.pointer(self._head_tab)
        s.adapt_w, s.adapt_b = self._aw.data_ptr(), self._ab.data_ptr()
        s.alpha_bar, s.T_train = self.alpha_bar.data_ptr(), self.alpha_bar.numel()
        s.guidance, s.eta = self.guidance, self.eta
        s.split_streams = 1 if self._split_streams else 0
        self.desc = s
        need = L.lib().avd_step_workspace_bytes(C.byref(s))
        if need < 0:
            raise ValueError(L.lib().avd_last_error().decode())
        if self.workspace is None or self.workspace.numel() < need:
            self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ptrs = self._table_ptrs()
        self._scalars = self._table_scalars()
        self._wkey = self._weights_key()

    def _table_ptrs(self):
        return tuple(t.data_ptr() for t in self._keep_core[1]) + tuple(t.data_ptr() for t in self._keep_head[1]) + \
            (self._aw.data_ptr(), self._ab.data_ptr(), self.workspace.data_ptr(), self.alpha_bar.data_ptr())

    def _table_scalars(self):
        """Every BY-VALUE number of the tables: a captured HIP graph bakes these into its kernel arguments (the f16x2 image scales
        become 1/(s_A s_W) factors and split scales of the launches), so unlike weights behind unchanged pointers they do not follow
        an in-place parameter update."""
        cw, hw = self._core_tab, self._head_tab
        blocks = self._keep_core[0]
        out = [cw.d, cw.n_layers, cw.n_heads, cw.mlp_hidden, cw.norm_eps, cw.norm_kind, cw.split_terms, cw.attn_mode,
               hw.d_in, hw.hidden, hw.d_out, hw.n_shared, hw.ln_eps, hw.act, hw.split_terms, self.guidance, self.eta]
        if cw.split_terms == 3:
            for i in range(cw.n_layers):
                out += list(blocks[i].f16x2_scale)
        if hw.split_terms == 3 and hw.f16x2_scale:
            out += [hw.f16x2_scale[i] for i in range(2 * (hw.n_shared + 2))]
        return tuple(float(v) for v in out)

    def _sync_weights(self) -> None:
        if self._weights_key() == self._wkey:
            return
        old, old_sc = self._ptrs, self._scalars
        self._bind_weights()             # derived copies are refreshed in place where shapes allow
        if self._ptrs != old:
            self._generation += 1
            self._stale_reason = "a parameter or the workspace was re-allocated"
        elif self._scalars != old_sc:
            self._generation += 1
            self._stale_reason = ("an in-place parameter update changed a scale that a captured HIP graph holds by value (f16x2 image "
                                  "scales follow max|w| and the norm gains)")
        if self._prompt_latent is not None:
            self.set_prompt(self._prompt_latent)     # the cached prompt rows depend on the prompt adapter (same buffer: in place)

    # ---- prompt rows: adapter(tokens(prompt latent)) | temb(0); constant over the trajectory ----
    def set_prompt(self, prompt_latent: torch.Tensor) -> torch.Tensor:
        z = L.dev_f32(prompt_latent, "prompt latent")
        self._prompt_latent = z
        B = self.embed.B
        if z.shape[0] != B:
            raise ValueError("prompt batch size must match the engine's")
        tok = latents_to_tokens_audio(z, *self.chunk) if self.target == "video" else \
            latents_to_tokens_video(z, self.tube[0], self.tube[1])
        if tok.shape[1] != self.embed.Np:
            raise ValueError(f"prompt yields {tok.shape[1]} tokens, engine was built for {self.embed.Np}")
        d, td = self.d, self.tdim
        # the prompt rows keep their buffer from call to call (a captured graph holds its address); a first call, or one after the
        # buffer was dropped, starts a new table generation
        Xp = self.Xp
        if Xp is None or tuple(Xp.shape) != (B, tok.shape[1], d) or Xp.device != self.device:
            Xp = torch.empty(B, tok.shape[1], d, device=self.device, dtype=torch.float32)
            self._generation += 1
            self._stale_reason = "the prompt rows were (re-)allocated by set_prompt()"
        w, b = self.adapt_p.proj.weight.detach(), self.adapt_p.proj.bias.detach()
        t0 = su.timestep_embedding(torch.zeros(B, dtype=torch.long, device=self.device), td) if td else None
        if self.temb_mode == "add":
            # adapter(tok) + temb(0): the broadcast embedding rides in as the GEMM's residual operand
            res = t0[:, None, :].expand(B, tok.shape[1], d).contiguous()
            L.check(L.lib().avd_gemm_bias_act_f32(tok.data_ptr(), tok.shape[2], L.dev_f32(w).data_ptr(), L.dev_f32(b).data_ptr(),
                                                  res.data_ptr(), d, Xp.data_ptr(), d, B * tok.shape[1], d, tok.shape[2],
                                                  L.ACT_NONE, L.stream_ptr(self.device)))
        else:
            # GEMM writes straight into the first d-tdim columns (ldc = d)
            L.check(L.lib().avd_gemm_bias_act_f32(tok.data_ptr(), tok.shape[2], L.dev_f32(w).data_ptr(), L.dev_f32(b).data_ptr(),
                                                  None, 0, Xp.data_ptr(), d, B * tok.shape[1], d - td, tok.shape[2],
                                                  L.ACT_NONE, L.stream_ptr(self.device)))
            if td:
                Xp[..., d - td:] = t0[:, None, :]
        self.Xp = Xp
        return Xp

    def step(self, z: torch.Tensor, t_now: torch.Tensor, t_prev: torch.Tensor,
             noise: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.Xp is None:
            raise RuntimeError("call set_prompt() first")
        z = L.dev_f32(z, "z")
        if tuple(z.shape) != self.latent_shape:
            raise ValueError(f"latent shape {tuple(z.shape)} != engine shape {self.latent_shape}")
        tn, tp = L.dev_i64(t_now, self.device), L.dev_i64(t_prev, self.device)
        if not torch.cuda.is_current_stream_capturing():
            self._sync_weights()
        if self.eta > 0 and noise is None:
            noise = torch.randn_like(z)
        out = torch.empty_like(z) if out is None else out
        L.check(L.lib().avd_denoise_step_f32(C.byref(self.desc), z.data_ptr(), self.Xp.data_ptr(), tn.data_ptr(),
                                             tp.data_ptr(), L.ptr(noise), out.data_ptr(), self.workspace.data_ptr(),
                                             self.workspace.numel(), L.stream_ptr(self.device)))
        return out

    def eps_tokens(self) -> torch.Tensor:
        """cond/null ε̂ tokens [2B,Nt,D] left in the workspace by the last step (debug / parity only)."""
        e = self.embed
        D = self.head.output_dims[self.target]
        n = 2 * e.B * e.Nt * D
        tail = self.workspace[self.workspace.numel() - ((n * 4 + 255) // 256) * 256:]
        return tail[: n * 4].view(torch.float32).view(2 * e.B, e.Nt, D).clone()

    # ---- whole trajectory -------------------------------------------------------------------------------
    def begin(self, sched: torch.Tensor) -> None:
        """Upload a sampling schedule and put the device-side cursor at its start (no per-step H2D afterwards)."""
        dev, B = self.device, self.embed.B
        self._sched = sched.to(dev, torch.long).contiguous()
        self._cursor = torch.zeros(1, dtype=torch.int32, device=dev)
        self._tn = torch.empty(B, dtype=torch.long, device=dev)
        self._tp = torch.empty(B, dtype=torch.long, device=dev)

    def rewind(self) -> None:
        self._cursor.zero_()

    def advance(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        """dst = one step from src at the cursor's (t_now, t_prev); the cursor moves on, all on the stream."""
        L.check(L.lib().avd_sched_advance(self._sched.data_ptr(), self._sched.numel(), self._cursor.data_ptr(),
                                          self._tn.data_ptr(), self._tp.data_ptr(), self.embed.B,
                                          L.stream_ptr(self.device)))
        self.step(src, self._tn, self._tp, out=dst)

    def capture_pair(self, za: torch.Tensor, zb: torch.Tensor) -> "_CapturedPair":
        """Capture two steps (za -> zb -> za) into one HIP graph; replaying it advances the trajectory by two."""
        if self.eta > 0:
            raise NotImplementedError("graph replay with eta > 0 would replay the same noise")
        self._sync_weights()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.advance(za, zb)
            self.advance(zb, za)
        return _CapturedPair(self, g)

    GRAPH_BELOW_ROWS = 6144      # 2B*N under which a step's ~60-95 launches are host-bound: replay them from a HIP graph

    def run(self, z: torch.Tensor, sched: torch.Tensor, graph: Optional[bool] = None) -> torch.Tensor:
        """Apply len(sched)-1 steps.  ``graph=True`` replays a captured two-step HIP graph; ``None`` (default) does so when the
        batch is small enough for the step to be launch-bound (2B*N < 6,144 rows, eta == 0) — results are bit-identical either
        way (tests: test_chained_sampler_golden)."""
        if graph is None:
            graph = self.eta == 0 and 2 * self.embed.B * self.N < self.GRAPH_BELOW_ROWS
        self.begin(sched)
        za = L.dev_f32(z, "z").clone()
        zb = torch.empty_like(za)
        n_steps = self._sched.numel() - 1
        if not graph or n_steps < 3:
            for _ in range(n_steps):
                self.advance(za, zb)
                za, zb = zb, za
            return za
        self.advance(za, zb)                        # warm-up step outside capture
        za, zb = zb, za
        g = self.capture_pair(za, zb)               # capture enqueues nothing: the cursor still reads 1
        for _ in range((n_steps - 1) // 2):
            g.replay()
        if (n_steps - 1) % 2:
            self.advance(za, zb)
            za = zb
        return za


# ----------------------------------------------------------------------------------------------------------
# reference-signature entry point (B = 1, codec / VAE supplied by the caller)
# ----------------------------------------------------------------------------------------------------------

@torch.no_grad()
def sample_one_direction(*, cfg: Dict, vid_vae, aud_codec, adapt_v: LinearAdapter, adapt_a: LinearAdapter,
                         core: MMDiT, head: MultiModalNoiseHead, tstep_dim: int, prompt_modality: str,
                         prompt_video: Optional[np.ndarray], prompt_audio: Optional[np.ndarray],
                         device: torch.device, init_noise: Optional[torch.Tensor] = None) -> Dict[str, np.ndarray]:
    """sample_clip.py:220-394 with the loop on the HIP engine.  The V->A branch uses the [1,3,T,H,W] layout the
    reference's comment intends (its own permute at :288 is a bug that crashes in conv3d).
    ``init_noise`` (extension; default None = draw it as the reference does, :297 / :304): the target's initial latent, so that a
    run can be repeated across devices — the reference's only RNG draw comes from the device generator."""
    dcfg, scfg = cfg["diffusion"], cfg["sampling"]
    eta = float(scfg.get("ddim_eta", 0.0))
    t_p, p = int(cfg["tokenizer"]["video"]["tube"]["t"]), int(cfg["tokenizer"]["video"]["tube"]["h"])
    l_chunk = int(cfg["tokenizer"]["audio"]["chunk"]["length"])
    s_chunk = int(cfg["tokenizer"]["audio"]["chunk"]["stride"])
    Cv, t_down, s_down = (int(cfg["video"]["latent"][k]) for k in ("channels", "t_down", "s_down"))
    Ca, Fa = int(cfg["audio"]["latent"]["channels"]), int(cfg["audio"]["latent"]["frames_per_clip"])
    fps, sr = int(cfg["video"]["fps"]), int(cfg["audio"]["sr"])
    H, W = int(cfg["video"]["size"][0]), int(cfg["video"]["size"][1])

    def table(m):
        c = dcfg[m]
        betas = su.make_beta_schedule(int(c["steps"]), kind=c["schedule"], min_beta=c["min_beta"], max_beta=c["max_beta"])
        return su.alphas_cumprod_from_betas(betas)[1], su.make_sampling_schedule(int(c["steps"]), int(c["sampler_steps"]))

    if prompt_modality == "video":
        if prompt_video is None:
            raise ValueError("prompt_video frames required for prompt_modality=video")
        frames = torch.from_numpy(prompt_video).to(device).float() / 255.0          # [T,H,W,3]
        z_p = vid_vae.encode(frames.permute(3, 0, 1, 2).unsqueeze(0).contiguous())   # [1,3,T,H,W] -> [1,Cv,T',H',W']
        z = torch.randn(1, Ca, Fa, device=device) if init_noise is None else init_noise.to(device).float()
        if tuple(z.shape) != (1, Ca, Fa):
            raise ValueError(f"init_noise has shape {tuple(z.shape)}, expected {(1, Ca, Fa)}")
        target, guide = "audio", float(scfg["guidance_scale"].get("audio", 3.0))
        n_prompt = (z_p.shape[2] // t_p) * (z_p.shape[3] // p) * (z_p.shape[4] // p)
    elif prompt_modality == "audio":
        if prompt_audio is None:
            raise ValueError("prompt_audio required for prompt_modality=audio")
        wav = torch.from_numpy(prompt_audio).to(device).view(1, 1, -1)
        z_p = aud_codec.encode(wav)                                                  # [1,Ca,Fa]
        T_in = prompt_video.shape[0] if prompt_video is not None else int(round(cfg["data"]["clip_seconds"] * fps))
        lat_shape = (1, Cv, max(1, T_in // t_down), H // s_down, W // s_down)
        z = torch.randn(*lat_shape, device=device) if init_noise is None else init_noise.to(device).float()
        if tuple(z.shape) != lat_shape:
            raise ValueError(f"init_noise has shape {tuple(z.shape)}, expected {lat_shape}")
        target, guide = "video", float(scfg["guidance_scale"].get("video", 3.0))
        n_prompt = (z_p.shape[-1] - l_chunk) // s_chunk + 1
    else:
        raise ValueError("prompt_modality must be 'video' or 'audio'")

    abar, sched = table(target)
    eng = DenoiseEngine(adapt_v=adapt_v, adapt_a=adapt_a, core=core, head=head, tstep_dim=tstep_dim, target=target,
                        latent_shape=tuple(z.shape), prompt_tokens=n_prompt, alpha_bar=abar, guidance=guide, eta=eta,
                        tube=(t_p, p, p), chunk=(l_chunk, s_chunk))
    eng.set_prompt(z_p.float())
    z = eng.run(z, sched)
    if target == "audio":
        return {"audio": aud_codec.decode(z).squeeze(0).squeeze(0).detach().cpu().numpy(), "sr": sr}
    x_hat = vid_vae.decode(z).clamp(0, 1)
    return {"video": (x_hat[0].permute(1, 2, 3, 0).detach().cpu().numpy() * 255.0).astype(np.uint8), "fps": fps}
