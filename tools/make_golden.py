#!/usr/bin/env python3
"""Generate golden vectors by importing the *reference* implementation (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py [--full-size-report]

The reference tree (/root/reference) is read-only and never travels to the GPU box; only the small
``.npz`` fixtures written to ``tests/golden/`` do.  A fixture holds inputs, weights and the outputs
the reference produced for them — no reference source.  If /root/reference is absent the script
exits with a message (tests then rely on the committed fixtures alone).

Fixture list: G1 schedules, G2 timestep embedding, G3 token index maps,
G4 RMSNorm, G5 Block/MMDiT, G6 MultiModalNoiseHead, G7 ddim_step, G8 one CFG step both directions,
G9 chained A->V via the reference's own ``sample_one_direction``, G10 TimestepEmbedder(mlp),
G11 VideoVAE.decode, G12 VideoVAE.encode, G13 AudioCodec,
G14 stream_infer splitting / cross-fade stitching,
G15 one CFG step with the TRAINER's embedding (d-wide adapters, timestep embedding added; train/trainer.py:36-49),
G16 drop-in corners: MMDiT with key_padding_mask, norm="layernorm", VideoVAE variational eval encode, non-GELU head.
G17 one CFG step at the bench's full model width (d=512, L=8, 421 tokens), batch 8, weights by seeded recipe.
G19 the reference's SHIPPED configuration through its own entry point: sample_one_direction(prompt_modality="audio") on unmodified
    configs/mvp.yaml + configs/a2v.yaml (d=512, L=8, 128 x 128, 60 DDIM steps, g=3.5, B=1); weights by seeded recipes (not stored),
    the initial latent, the pre-decode latent and every 6th uint8 frame stored.
G18 the reference CLASS-DEFAULT geometry (mmdt.py:125-126: d_model=1024, n_heads=16; 2 layers for time): MMDiT.forward on 16 x 421
    tokens, and MultiModalNoiseHead at d=1024 with the reference shape test's token counts (tests/test_shapes.py:86-107: Nv=96, Na=37).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"


def _np(t):
    return t.detach().cpu().numpy()


def _sd(mod):
    return {k: _np(v) for k, v in mod.state_dict().items()}


ONLY = None          # --only g15,g16: write just these fixtures (the others are computed and dropped)


def _save(name, **arrays):
    if ONLY is not None and name.split("_")[0] not in ONLY:
        return
    OUT.mkdir(parents=True, exist_ok=True)
    path = OUT / name
    np.savez_compressed(path, **arrays)
    print(f"  wrote {path.name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def _flat(prefix, d):
    return {f"{prefix}/{k}": v for k, v in d.items()}


def small_cfg(load_config):
    """mvp.yaml + a2v.yaml shrunk to a model whose weights fit in a small fixture."""
    cfg = load_config(str(REF / "configs/mvp.yaml"), str(REF / "configs/a2v.yaml"))
    cfg = copy.deepcopy(cfg)
    cfg["paths"] = {}
    cfg["video"]["size"] = [64, 64]            # latent 8x8
    cfg["data"]["clip_seconds"] = 1.0          # 16 frames -> T'=4
    cfg["audio"]["latent"]["frames_per_clip"] = 22
    cfg["tokenizer"]["width"] = 128
    cfg["embeddings"]["timestep_dim"] = 64
    cfg["model"]["core"].update(d_model=128, n_layers=1, n_heads=2, mlp_ratio=2.0)
    for m in ("video", "audio"):
        cfg["model"]["heads"][m]["hidden_dim"] = 64
    cfg["diffusion"]["video"]["sampler_steps"] = 4
    cfg["diffusion"]["audio"]["sampler_steps"] = 4
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-size-report", action="store_true",
                    help="also compare oracle vs reference at mvp.yaml size (slow, writes a JSON report)")
    ap.add_argument("--only", default=None, help="comma-separated fixture ids to write, e.g. g15,g16 (default: all)")
    args = ap.parse_args()
    global ONLY
    ONLY = set(args.only.split(",")) if args.only else None
    if not REF.exists():
        print("reference tree not present; nothing to do (committed fixtures stay authoritative)")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, str(REF))
    from avdiff.utils import schedule_utils as su, ops
    from avdiff.utils.io import load_config
    from avdiff.models.mmdt import MMDiT, Block, RMSNorm
    from avdiff.models.heads.noise_heads import MultiModalNoiseHead
    from avdiff.models.adapters import TimestepEmbedder, TimestepCfg
    from avdiff.models.infer import sample_clip as sc

    torch.set_grad_enabled(False)

    # ---- G1 schedules -------------------------------------------------------------------
    g1 = {}
    for kind in ("cosine", "linear", "sigmoid"):
        b = su.make_beta_schedule(1000, kind=kind, min_beta=1e-4, max_beta=0.02)
        _, ab = su.alphas_cumprod_from_betas(b)
        g1[f"betas/{kind}"] = _np(b)
        g1[f"abar/{kind}"] = _np(ab)
    for S in (10, 25, 50, 60, 100):
        g1[f"sched/{S}"] = _np(su.make_sampling_schedule(1000, S))
    g1["sched/T50_S7"] = _np(su.make_sampling_schedule(50, 7))
    _save("g1_schedules.npz", **g1)

    # ---- G2 timestep embedding ----------------------------------------------------------
    t = torch.tensor([0, 1, 17, 500, 982, 999], dtype=torch.long)
    _save("g2_temb.npz", t=_np(t), e256=_np(su.timestep_embedding(t, 256)), e64=_np(su.timestep_embedding(t, 64)),
          e7=_np(su.timestep_embedding(t, 7)))

    # ---- G3 index maps ------------------------------------------------------------------
    g3 = {}
    for (C, T, H, W) in ((8, 12, 16, 16), (8, 12, 32, 32), (8, 4, 8, 8)):
        z = torch.arange(C * T * H * W, dtype=torch.float32).view(1, C, T, H, W)
        tok = ops.tube_patch_video(z, 2, 4, 4)
        back = ops.tube_unpatch_video(tok, C=C, T=T, H=H, W=W, t=2, h=4, w=4)
        assert torch.equal(back, z)
        g3[f"patch/{C}x{T}x{H}x{W}"] = _np(tok).astype(np.int32)
    za = torch.arange(2 * 8 * 150, dtype=torch.float32).view(2, 8, 150)
    tok_a = sc.latents_to_tokens_audio(za, 4, 4)
    g3["audio_tok/150"] = _np(tok_a).astype(np.int32)
    g3["audio_untok/150"] = _np(sc.tokens_to_latents_audio(tok_a + 1.0, Ca=8, l_chunk=4, Fa=150, stride=4)).astype(np.int32)
    zb = torch.randn(2, 8, 22, generator=torch.Generator().manual_seed(3))
    tb = sc.latents_to_tokens_audio(zb, 4, 4)
    g3["audio22/z"] = _np(zb)
    g3["audio22/tok"] = _np(tb)
    g3["audio22/untok"] = _np(sc.tokens_to_latents_audio(tb, Ca=8, l_chunk=4, Fa=22, stride=4))
    # overlapping windows (stride < length) exercise the overlap-count normalisation
    tc = sc.latents_to_tokens_audio(zb, 4, 2)
    g3["audio22s2/tok"] = _np(tc)
    g3["audio22s2/untok"] = _np(sc.tokens_to_latents_audio(tc, Ca=8, l_chunk=4, Fa=22, stride=2))
    _save("g3_index.npz", **g3)

    # ---- G4 RMSNorm ---------------------------------------------------------------------
    gen = torch.Generator().manual_seed(4)
    x = torch.randn(6, 512, generator=gen)
    x[1] *= 1e-7          # near-zero row: eps (outside sqrt) dominates
    x[2] = 0.0
    x[3] *= 1e3
    n = RMSNorm(512)
    n.scale.copy_(1.0 + 0.1 * torch.randn(512, generator=gen))
    _save("g4_rmsnorm.npz", x=_np(x), scale=_np(n.scale), y=_np(n(x)))

    # ---- reduced model ------------------------------------------------------------------
    cfg = small_cfg(load_config)
    torch.manual_seed(0)
    vid_vae, aud_codec, adapt_v, adapt_a, core, head, tdim = sc.build_components(cfg, torch.device("cpu"))
    # perturb zero-init biases / unit norm scales so they are exercised
    g = torch.Generator().manual_seed(5)
    for mod in (core, head):
        for name, p in mod.named_parameters():
            if name.endswith("bias") or name.endswith("scale") or (".1.weight" in name):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    Wsmall = {**_flat("core", _sd(core)), **_flat("head", _sd(head)),
              **_flat("adapt_v", _sd(adapt_v)), **_flat("adapt_a", _sd(adapt_a))}
    meta = dict(d=128, n_layers=1, n_heads=2, tdim=64, mlp_ratio=2.0, head_hidden=64)

    # ---- G5 Block / MMDiT ---------------------------------------------------------------
    gen = torch.Generator().manual_seed(6)
    x5 = torch.randn(2, 13, 128, generator=gen)
    y_blk = core.blocks[0](x5)
    y_core = core(x5)
    x5b = torch.randn(3, 70, 128, generator=gen) * 2.0     # > one 64-key tile, ragged
    _save("g5_mmdit_small.npz", x=_np(x5), y_block0=_np(y_blk), y=_np(y_core), x_b=_np(x5b), y_b=_np(core(x5b)),
          meta=np.array(json.dumps(meta)), **Wsmall)

    # ---- G6 MultiModalNoiseHead ---------------------------------------------------------
    hv = torch.randn(2, 8, 128, generator=gen)
    ha = torch.randn(2, 5, 128, generator=gen)
    o = head({"video": hv, "audio": ha})
    _save("g6_head_small.npz", hv=_np(hv), ha=_np(ha), out_v=_np(o["video"]), out_a=_np(o["audio"]))

    # ---- G7 ddim_step -------------------------------------------------------------------
    _, abar = su.alphas_cumprod_from_betas(su.make_beta_schedule(1000, "cosine", 1e-4, 0.02))
    xt = torch.randn(5, 8, 2, 4, 4, generator=gen)
    eh = torch.randn(5, 8, 2, 4, 4, generator=gen)
    tn = torch.tensor([999, 500, 19, 982, 16], dtype=torch.long)
    tp = torch.tensor([979, 480, -1, 966, -1], dtype=torch.long)
    _save("g7_ddim.npz", x_t=_np(xt), eps=_np(eh), t_now=_np(tn), t_prev=_np(tp), abar=_np(abar),
          x_prev=_np(su.ddim_step(xt, tn, tp, eh, abar, eta=0.0)))

    # ---- G8 one CFG step in both directions (reference loop-body statements, batched) ----
    B = 2
    Cv, t_p, p = 8, 2, 4
    l_chunk = s_chunk = 4
    z_v = torch.randn(B, 8, 4, 8, 8, generator=gen)
    z_a = torch.randn(B, 8, 22, generator=gen)
    tn = torch.tensor([982, 500], dtype=torch.long)
    tp = torch.tensor([966, 480], dtype=torch.long)
    zero_t = torch.zeros(B, dtype=torch.long)
    g8 = dict(z_v=_np(z_v), z_a=_np(z_a), t_now=_np(tn), t_prev=_np(tp), abar=_np(abar))
    for guide in (0.0, 1.0, 3.5):
        # A->V  (sample_clip.py:363-389)
        tok_v = sc.latents_to_tokens_video(z_v, t_p=t_p, p=p)
        tok_a = sc.latents_to_tokens_audio(z_a, l_chunk=l_chunk, s_chunk=s_chunk)
        Nv = tok_v.size(1)
        Xv = sc.add_sinusoidal_timestep(adapt_v(tok_v), tn, tdim)
        Xa = sc.add_sinusoidal_timestep(adapt_a(tok_a), zero_t, tdim)
        hc = core(torch.cat([Xv, Xa], 1))
        ec = head({"video": hc[:, :Nv], "audio": hc[:, Nv:]})["video"]
        hn = core(torch.cat([Xv, torch.zeros_like(Xa)], 1))
        en = head({"video": hn[:, :Nv], "audio": hn[:, Nv:]})["video"]
        et = en + guide * (ec - en)
        el = ops.tube_unpatch_video(et, C=Cv, T=4, H=8, W=8, t=t_p, h=p, w=p)
        g8[f"a2v/g{guide}/eps_tok"] = _np(et)
        g8[f"a2v/g{guide}/z_next"] = _np(su.ddim_step(z_v, tn, tp, el, abar, eta=0.0))
        if guide == 3.5:
            g8["a2v/X"] = _np(torch.cat([Xv, Xa], 1))
            g8["a2v/eps_cond"] = _np(ec)
            g8["a2v/eps_null"] = _np(en)
        # V->A  (sample_clip.py:322-348)
        Xv = sc.add_sinusoidal_timestep(adapt_v(tok_v), zero_t, tdim)
        Xa = sc.add_sinusoidal_timestep(adapt_a(tok_a), tn, tdim)
        hc = core(torch.cat([Xv, Xa], 1))
        ec = head({"video": hc[:, :Nv], "audio": hc[:, Nv:]})["audio"]
        hn = core(torch.cat([torch.zeros_like(Xv), Xa], 1))
        en = head({"video": hn[:, :Nv], "audio": hn[:, Nv:]})["audio"]
        et = en + guide * (ec - en)
        el = sc.tokens_to_latents_audio(et, Ca=8, l_chunk=l_chunk, Fa=22, stride=s_chunk)
        g8[f"v2a/g{guide}/eps_tok"] = _np(et)
        g8[f"v2a/g{guide}/z_next"] = _np(su.ddim_step(z_a, tn, tp, el, abar, eta=0.0))
    _save("g8_cfg_step_small.npz", **g8)

    # ---- G9 chained A->V through the reference's own sampler (B=1, 4 steps) ---------------
    rec = {}
    orig_enc, orig_dec = aud_codec.encode, vid_vae.decode

    def enc_hook(w):
        z = orig_enc(w)
        rec["z_a0"] = z.clone()
        return z

    def dec_hook(z, *a, **k):
        rec["z_final"] = z.clone()
        return orig_dec(z, *a, **k)

    aud_codec.encode, vid_vae.decode = enc_hook, dec_hook
    wav = (0.1 * torch.randn(16000, generator=torch.Generator().manual_seed(9))).numpy().astype(np.float32)
    torch.manual_seed(1234)
    res = sc.sample_one_direction(cfg=cfg, vid_vae=vid_vae, aud_codec=aud_codec, adapt_v=adapt_v, adapt_a=adapt_a,
                                  core=core, head=head, tstep_dim=tdim, prompt_modality="audio",
                                  prompt_video=None, prompt_audio=wav, device=torch.device("cpu"))
    aud_codec.encode, vid_vae.decode = orig_enc, orig_dec
    torch.manual_seed(1234)
    z_init = torch.randn(1, 8, 4, 8, 8)         # the sampler's only RNG draw (sample_clip.py:304)
    _save("g9_chain_small.npz", z_init=_np(z_init), z_a0=_np(rec["z_a0"]), z_final=_np(rec["z_final"]),
          sched=_np(su.make_sampling_schedule(1000, 4)), abar=_np(abar), guidance=np.float32(3.5),
          frames_shape=np.array(res["video"].shape))

    # ---- G10 TimestepEmbedder(mode="mlp") -------------------------------------------------
    torch.manual_seed(10)
    te = TimestepEmbedder(TimestepCfg(dim=64, mode="mlp"))
    tt = torch.tensor([0, 3, 500, 999], dtype=torch.long)
    _save("g10_tmlp.npz", t=_np(tt), y=_np(te(tt)), **_flat("w", _sd(te)))

    # ---- G11 VideoVAE.decode (loop boundary, next-1): reference module at its default decoder width ----------
    from avdiff.models.encoders.vae_video3d import VideoVAE
    torch.manual_seed(11)
    vae = VideoVAE.from_config({"latent": {"channels": 8, "t_down": 4, "s_down": 8}}).eval()
    g = torch.Generator().manual_seed(12)
    for name, p_ in vae.named_parameters():          # exercise biases / GroupNorm affine (zeros / ones at init)
        if name.startswith(("dec_net", "from_lat", "to_img")) and p_.dim() == 1:
            p_.add_(0.1 * torch.randn(p_.shape, generator=g))
    zz = torch.randn(2, 8, 2, 4, 4, generator=g)
    dec_sd = {k: v for k, v in _sd(vae).items() if k.startswith(("from_lat", "dec_net", "to_img"))}
    _save("g11_vae_decode.npz", z=_np(zz), x=_np(vae.decode(zz)), x_odd=_np(vae.decode(zz[:1], out_size=(6, 24, 40))),
          **_flat("w", dec_sd))

    # ---- G12 VideoVAE.encode (prompt side of V->A): same module, incl. the center-crop of a non-divisible clip ----
    for name, p_ in vae.named_parameters():
        if name.startswith(("enc_net", "to_lat")) and p_.dim() == 1:
            p_.add_(0.1 * torch.randn(p_.shape, generator=g))
    xx = torch.rand(2, 3, 8, 16, 24, generator=g)
    x_crop = torch.rand(1, 3, 9, 18, 17, generator=g)           # -> crops to (8,16,16)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        z_crop = vae.encode(x_crop)
    enc_sd = {k: v for k, v in _sd(vae).items() if k.startswith(("enc_net", "to_lat"))}
    _save("g12_vae_encode.npz", x=_np(xx), z=_np(vae.encode(xx)), x_crop=_np(x_crop), z_crop=_np(z_crop), **_flat("w", enc_sd))

    # ---- G13 AudioCodec encode / decode (next-2), shipped geometry: hop 320, 150 frames, 1 s and 3 s clips ------
    from avdiff.models.encoders.audio_codec import AudioCodec
    torch.manual_seed(13)
    codec = AudioCodec.from_config({"sr": 16000, "latent": {"channels": 8, "frames_per_clip": 150},
                                    "codec": {"hop_samples": 320, "hidden": 64, "smooth_kernel": 7}}).eval()
    g = torch.Generator().manual_seed(14)
    for name, p_ in codec.named_parameters():
        if p_.dim() == 1:
            p_.add_(0.05 * torch.randn(p_.shape, generator=g))
    wav = 0.3 * torch.randn(2, 1, 16000, generator=g)
    zc = torch.randn(2, 8, 10, generator=g)
    _save("g13_audio_codec.npz", wav=_np(wav), z=_np(codec.encode(wav)), z_in=_np(zc), wav_out=_np(codec.decode(zc)),
          **_flat("w", _sd(codec)))

    # ---- G14 stream_infer stitching (next-3).  The reference module imports `avdiff.infer.sample_clip`, a path that
    # does not exist in its own tree (the package lives at avdiff.models.infer — SURVEY §0.3); alias the package name
    # it asks for to the one that exists so its pure-numpy helpers can be driven unmodified.
    import avdiff.models.infer as _infer_pkg
    import avdiff.models.infer.sample_clip as _sc_mod
    sys.modules.setdefault("avdiff.infer", _infer_pkg)
    sys.modules.setdefault("avdiff.infer.sample_clip", _sc_mod)
    from avdiff.models.infer import stream_infer as si
    rng = np.random.default_rng(15)
    a_chunks = rng.standard_normal((4, 1000)).astype(np.float32)
    v_chunks = rng.integers(0, 256, size=(3, 12, 4, 5, 3), dtype=np.uint8)
    y_long = rng.standard_normal(7300).astype(np.float32)
    f_long = rng.integers(0, 256, size=(23, 2, 2, 3), dtype=np.uint8)
    sa, wa, ha = si.split_audio_into_windows(y_long, sr=1000, win_s=3.0, hop_s=1.0)
    sf, wf, hf = si.split_frames_into_windows(f_long, fps=4, win_s=3.0, hop_s=1.0)
    _save("g14_stream_stitch.npz", a_chunks=a_chunks, v_chunks=v_chunks,
          a_fade=si.crossfade_audio(a_chunks, sr=1000, hop=400, win=1000, fade_s=0.25),
          a_rect=si.crossfade_audio(a_chunks, sr=1000, hop=400, win=1000, fade_s=0.0),
          v_fade=si.crossfade_video(v_chunks, hop=4, win=12, fade_f=3),
          v_rect=si.crossfade_video(v_chunks, hop=4, win=12, fade_f=0),
          y_long=y_long, f_long=f_long, split_a=sa, split_f=sf, split_meta=np.array([wa, ha, wf, hf]))

    # ---- G15 trainer-style embedding (next-4).  avdiff.models.train.trainer cannot be imported (tensorboard is absent), but
    # its two embedding helpers (trainer.py:36-49: LinearAdapter, add_sinusoidal_timestep) depend only on torch and `su`:
    # take exactly those two definitions out of the module's syntax tree and execute them, unmodified, against the reference's
    # own schedule_utils.  One CFG step A->V with the reduced core/head above and d-wide adapters, statements as in G8.
    import ast
    tsrc = (REF / "avdiff/models/train/trainer.py").read_text()
    keep = [n for n in ast.parse(tsrc).body
            if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in ("LinearAdapter", "add_sinusoidal_timestep")]
    assert [n.name for n in keep] == ["LinearAdapter", "add_sinusoidal_timestep"]
    tns = {"torch": torch, "su": su}
    exec(compile(ast.Module(body=keep, type_ignores=[]), str(REF / "avdiff/models/train/trainer.py"), "exec"), tns)
    torch.manual_seed(15)
    av_t, aa_t = tns["LinearAdapter"](256, 128), tns["LinearAdapter"](32, 128)
    add_t = tns["add_sinusoidal_timestep"]
    gen = torch.Generator().manual_seed(16)
    z_v = torch.randn(B, 8, 4, 8, 8, generator=gen)
    z_a = torch.randn(B, 8, 22, generator=gen)
    tn = torch.tensor([982, 500], dtype=torch.long)
    tp = torch.tensor([966, 480], dtype=torch.long)
    tok_v = sc.latents_to_tokens_video(z_v, t_p=t_p, p=p)
    tok_a = sc.latents_to_tokens_audio(z_a, l_chunk=l_chunk, s_chunk=s_chunk)
    Nv = tok_v.size(1)
    Xv = add_t(av_t(tok_v), tn, 128)
    Xa = add_t(aa_t(tok_a), zero_t, 128)
    hc = core(torch.cat([Xv, Xa], 1))
    ec = head({"video": hc[:, :Nv], "audio": hc[:, Nv:]})["video"]
    hn = core(torch.cat([Xv, torch.zeros_like(Xa)], 1))
    en = head({"video": hn[:, :Nv], "audio": hn[:, Nv:]})["video"]
    et = en + 3.0 * (ec - en)
    el = ops.tube_unpatch_video(et, C=Cv, T=4, H=8, W=8, t=t_p, h=p, w=p)
    _save("g15_add_mode_step.npz", z_v=_np(z_v), z_a=_np(z_a), t_now=_np(tn), t_prev=_np(tp), abar=_np(abar),
          guidance=np.float32(3.0), X=_np(torch.cat([Xv, Xa], 1)), eps_cond=_np(ec), eps_null=_np(en), eps_tok=_np(et),
          z_next=_np(su.ddim_step(z_v, tn, tp, el, abar, eta=0.0)), **_flat("adapt_v", _sd(av_t)), **_flat("adapt_a", _sd(aa_t)))

    # ---- G16 drop-in corners: options the reference API accepts although the shipped sampler never uses them -----------------
    g16 = {}
    # (a) MMDiT.forward(key_padding_mask) (mmdt.py:134-149): reduced core above, ragged padding per sample
    gen = torch.Generator().manual_seed(17)
    xm = torch.randn(3, 70, 128, generator=gen)
    kpm = torch.zeros(3, 70, dtype=torch.bool)
    kpm[0, 50:] = True
    kpm[1, 69:] = True
    kpm[2, 3:40] = True
    g16["mask/x"], g16["mask/kpm"], g16["mask/y"] = _np(xm), _np(kpm), _np(core(xm, key_padding_mask=kpm))
    # (b) norm="layernorm" (build_norm, mmdt.py:44-45)
    torch.manual_seed(18)
    core_ln = MMDiT(d_model=128, n_layers=2, n_heads=2, mlp_ratio=2.0, norm="layernorm").eval()
    for name, p_ in core_ln.named_parameters():
        if p_.dim() == 1:
            p_.add_(0.05 * torch.randn(p_.shape, generator=gen))
    xl = torch.randn(2, 21, 128, generator=gen)
    g16["ln/x"], g16["ln/y"] = _np(xl), _np(core_ln(xl))
    g16.update(_flat("ln_core", _sd(core_ln)))
    # (c) head activations other than gelu (noise_heads.py:28-36)
    for act in ("relu", "leaky_relu"):
        torch.manual_seed(19)
        hd = MultiModalNoiseHead({"video": 128, "audio": 128}, {"video": 256, "audio": 32}, hidden_dim=64, num_shared_layers=2,
                                 num_modality_specific_layers=1, dropout=0.1, activation=act).eval()
        for name, p_ in hd.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.05 * torch.randn(p_.shape, generator=gen))
        hv = torch.randn(2, 8, 128, generator=gen)
        g16[f"head_{act}/hv"], g16[f"head_{act}/out_v"] = _np(hv), _np(hd({"video": hv})["video"])
        g16.update(_flat(f"head_{act}_w", _sd(hd)))
    # (d) VideoVAE(variational=True).eval().encode: z = to_mu(h) and the cached KL term (vae_video3d.py:175-185)
    torch.manual_seed(20)
    vv = VideoVAE.from_config({"latent": {"channels": 8, "t_down": 4, "s_down": 8}, "variational": True}).eval()
    for name, p_ in vv.named_parameters():
        if name.startswith(("enc_net", "to_mu", "to_logv")) and p_.dim() == 1:
            p_.add_(0.1 * torch.randn(p_.shape, generator=gen))
    xv = torch.rand(2, 3, 8, 16, 24, generator=gen)
    zv = vv.encode(xv)
    g16["vvae/x"], g16["vvae/z"], g16["vvae/kld"] = _np(xv), _np(zv), _np(vv.kld_loss())
    g16.update(_flat("vvae_w", {k: v for k, v in _sd(vv).items() if k.startswith(("enc_net", "to_mu", "to_logv"))}))
    # (e) overlap_add_1d(apply_hann=True) (ops.py:48-93)
    wnd = torch.randn(2, 3, 5, 8, generator=gen)
    g16["hann/windows"], g16["hann/y"] = _np(wnd), _np(ops.overlap_add_1d(wnd, stride=4, apply_hann=True))
    g16["hann/y_rect"] = _np(ops.overlap_add_1d(wnd, stride=3))
    _save("g16_dropin_corners.npz", **g16)

    # ---- G17 one CFG step A->V at the bench's full width (mvp.yaml model: d=512, L=8, H=8, 384+37 tokens) through the
    # reference's own modules, batch 8 (2B*N = 6,736 rows: large enough for every matrix-pipe mode of the HIP path to engage).
    # The 26 M weights are not stored: they are the seeded recipe oracle.synth_weights(0), loaded into the reference modules with
    # strict=True; inputs are seeded too.  Stored: the reference's eps tokens for sample 0 and its next latents for samples 0-1.
    if ONLY is None or "g17" in ONLY:
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
        from oracle import ref_cpu as R
        ws = R.synth_weights(seed=0)
        coreF = MMDiT(d_model=512, n_layers=8, n_heads=8, mlp_ratio=4.0).eval()
        coreF.load_state_dict(ws["core"], strict=True)
        headF = MultiModalNoiseHead({"video": 512, "audio": 512}, {"video": 256, "audio": 32}, hidden_dim=512).eval()
        headF.load_state_dict(ws["head"], strict=True)
        avF, aaF = sc.LinearAdapter(256, 256), sc.LinearAdapter(32, 256)
        avF.load_state_dict(ws["adapt_v"], strict=True)
        aaF.load_state_dict(ws["adapt_a"], strict=True)
        B17 = 8
        gen = torch.Generator().manual_seed(1717)
        z_v = torch.randn(B17, 8, 12, 32, 32, generator=gen)
        z_a = torch.randn(B17, 8, 150, generator=gen)
        tn17 = torch.tensor([982, 500, 16, 999, 700, 300, 64, 5], dtype=torch.long)
        tp17 = torch.tensor([966, 480, -1, 979, 680, 280, 48, -1], dtype=torch.long)
        abarF = su.alphas_cumprod_from_betas(su.make_beta_schedule(1000, "cosine", 1e-4, 0.02))[1]
        with torch.no_grad():
            tok_v = sc.latents_to_tokens_video(z_v, t_p=2, p=4)
            tok_a = sc.latents_to_tokens_audio(z_a, l_chunk=4, s_chunk=4)
            Nv = tok_v.size(1)
            Xv = sc.add_sinusoidal_timestep(avF(tok_v), tn17, 256)
            Xa = sc.add_sinusoidal_timestep(aaF(tok_a), torch.zeros(B17, dtype=torch.long), 256)
            hc = coreF(torch.cat([Xv, Xa], 1))
            ec = headF({"video": hc[:, :Nv], "audio": hc[:, Nv:]})["video"]
            hn = coreF(torch.cat([Xv, torch.zeros_like(Xa)], 1))
            en = headF({"video": hn[:, :Nv], "audio": hn[:, Nv:]})["video"]
            et = en + 3.5 * (ec - en)
            el = ops.tube_unpatch_video(et, C=8, T=12, H=32, W=32, t=2, h=4, w=4)
            zn = su.ddim_step(z_v, tn17, tp17, el, abarF, eta=0.0)
        _save("g17_full_step_c3.npz", meta=np.array(json.dumps(dict(seed_weights=0, seed_inputs=1717, B=B17, guidance=3.5, tokens=[int(Nv), int(tok_a.size(1))]))),
              t_now=_np(tn17), t_prev=_np(tp17), eps_tok0=_np(et[:1]), z_next01=_np(zn[:2]), z_next_absmax=np.float32(zn.abs().max()))

    # ---- G18 the reference's class-default width through its own modules: MMDiT(d_model=1024, n_heads=16) (mmdt.py:125-126; two of
    # the sixteen layers, to keep the CPU run and the GPU test short) on [16, 421, 1024] (6,736 rows: every matrix-pipe mode of the HIP
    # path engages), and MultiModalNoiseHead(input 1024 -> hidden 512 -> 256 | 32) on the reference shape test's token counts
    # (tests/test_shapes.py:86-107: Nv = 96, Na = 37) at batch 64 (6,144 video rows).  Weights: the seeded recipe
    # oracle.synth_weights(18, d=1024, n_layers=2), loaded with strict=True, not stored.  Stored: every 8th row of the first and the
    # last sample of the core output; the head's video tokens for the first and last sample and its audio tokens for the first.
    if ONLY is None or "g18" in ONLY:
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
        from oracle import ref_cpu as R
        ws18 = R.synth_weights(seed=18, d=1024, n_layers=2)
        core18 = MMDiT(d_model=1024, n_layers=2, n_heads=16).eval()          # every other kwarg at the class default
        core18.load_state_dict(ws18["core"], strict=True)
        head18 = MultiModalNoiseHead(input_dims={"video": 1024, "audio": 1024}, output_dims={"video": 256, "audio": 32}, hidden_dim=512,
                                     num_shared_layers=2, num_modality_specific_layers=1, dropout=0.1, activation="gelu").eval()
        head18.load_state_dict(ws18["head"], strict=True)
        gen = torch.Generator().manual_seed(1818)
        x18 = torch.randn(16, 421, 1024, generator=gen)
        hv18 = torch.randn(64, 96, 1024, generator=gen)
        ha18 = torch.randn(64, 37, 1024, generator=gen)
        with torch.no_grad():
            y18 = core18(x18)
            o18 = head18({"video": hv18, "audio": ha18})
        _save("g18_class_default_width.npz",
              meta=np.array(json.dumps(dict(seed_weights=18, seed_inputs=1818, d_model=1024, n_layers=2, n_heads=16, core_in=[16, 421, 1024],
                                            head_video_in=[64, 96, 1024], head_audio_in=[64, 37, 1024], row_stride=8))),
              core_first=_np(y18[0, ::8]), core_last=_np(y18[-1, ::8]), core_absmax=np.float32(y18.abs().max()),
              head_video_first=_np(o18["video"][0]), head_video_last=_np(o18["video"][-1]), head_audio_first=_np(o18["audio"][0]))

    # ---- G19 the reference's shipped configuration through its OWN sampler entry point (VERDICT r4 missing 4): configs/mvp.yaml merged
    # with configs/a2v.yaml, unmodified (d = 512, 8 layers, 128 x 128 x 48 frames, 60 DDIM steps for the video target, guidance 3.5, eta 0),
    # sample_one_direction(prompt_modality="audio"), B = 1 as the function hard-codes.  Weights: seeded recipes loaded with strict=True and
    # not stored — oracle.synth_weights(19) for core / head / adapters, oracle.synth_like(<state_dict shapes>, seed) for the VideoVAE (1919)
    # and the AudioCodec (2019).  Prompt: 0.1 N(0,1) x 48,000 samples from generator seed 190.  The sampler's single RNG draw (the
    # initial video latent, sample_clip.py:304) is fixed by torch.manual_seed(1900) and stored.  Stored besides: the codec's prompt
    # latent, the latent handed to VideoVAE.decode, every 6th of the 48 uint8 frames, and the subset of the merged config the sampler reads.
    if ONLY is None or "g19" in ONLY:
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
        from oracle import ref_cpu as R
        cfg19 = load_config(str(REF / "configs/mvp.yaml"), str(REF / "configs/a2v.yaml"))
        torch.manual_seed(19)
        vae19, codec19, av19, aa19, core19, head19, tdim19 = sc.build_components(cfg19, torch.device("cpu"))
        ws19 = R.synth_weights(seed=19)
        core19.load_state_dict(ws19["core"], strict=True)
        head19.load_state_dict(ws19["head"], strict=True)
        av19.load_state_dict(ws19["adapt_v"], strict=True)
        aa19.load_state_dict(ws19["adapt_a"], strict=True)
        vshapes = {k: tuple(v.shape) for k, v in vae19.state_dict().items()}
        cshapes = {k: tuple(v.shape) for k, v in codec19.state_dict().items()}
        vae19.load_state_dict(R.synth_like(vshapes, 1919), strict=True)
        codec19.load_state_dict(R.synth_like(cshapes, 2019), strict=True)
        wav19 = (0.1 * torch.randn(48000, generator=torch.Generator().manual_seed(190))).numpy().astype(np.float32)
        rec19 = {}
        enc0, dec0 = codec19.encode, vae19.decode

        def enc19(w):
            z = enc0(w)
            rec19["z_a0"] = z.clone()
            return z

        def dec19(z, *a, **k):
            rec19["z_final"] = z.clone()
            return dec0(z, *a, **k)

        codec19.encode, vae19.decode = enc19, dec19
        torch.manual_seed(1900)
        res19 = sc.sample_one_direction(cfg=cfg19, vid_vae=vae19, aud_codec=codec19, adapt_v=av19, adapt_a=aa19, core=core19, head=head19,
                                        tstep_dim=tdim19, prompt_modality="audio", prompt_video=None, prompt_audio=wav19,
                                        device=torch.device("cpu"))
        codec19.encode, vae19.decode = enc0, dec0
        torch.manual_seed(1900)
        z_init19 = torch.randn(1, 8, 12, 16, 16)
        assert tuple(rec19["z_final"].shape) == (1, 8, 12, 16, 16) and res19["video"].shape == (48, 128, 128, 3)
        keep = {k: cfg19[k] for k in ("video", "audio", "tokenizer", "embeddings", "model", "diffusion", "sampling")}
        keep["data"] = {"clip_seconds": cfg19["data"]["clip_seconds"]}
        _save("g19_shipped_config_a2v.npz",
              meta=np.array(json.dumps(dict(cfg=keep, seed_weights=19, seed_vae=1919, seed_codec=2019, seed_wav=190, seed_init=1900, frame_stride=6,
                                            vae_shapes={k: list(v) for k, v in vshapes.items()}, codec_shapes={k: list(v) for k, v in cshapes.items()}))),
              z_init=_np(z_init19), z_a0=_np(rec19["z_a0"]), z_final=_np(rec19["z_final"]), frames=res19["video"][::6],
              frames_mean=np.float64(res19["video"].astype(np.float64).mean()))

    # ---- optional: full-size live comparison oracle vs reference --------------------------
    if args.full_size_report:
        sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
        from oracle import ref_cpu as R
        rep = {}
        cfgF = load_config(str(REF / "configs/mvp.yaml"), str(REF / "configs/a2v.yaml"))
        cfgF["paths"] = {}
        torch.manual_seed(0)
        _, _, av, aa, coreF, headF, tdF = sc.build_components(cfgF, torch.device("cpu"))
        Wc = {k: v for k, v in coreF.state_dict().items()}
        Wh = {k: v for k, v in headF.state_dict().items()}
        for N in (133, 421):
            x = torch.randn(2, N, 512, generator=torch.Generator().manual_seed(N))
            ref = coreF(x)
            mine = R.mmdit_forward(x, Wc, 8, 8)
            rep[f"mmdit_N{N}_maxabs"] = float((ref - mine).abs().max())
            rep[f"mmdit_N{N}_refmax"] = float(ref.abs().max())
            hv = ref[:, : N - 37]
            rep[f"head_N{N}_maxabs"] = float((headF({"video": hv})["video"] - R.noise_head(hv, Wh, "video")).abs().max())
        (OUT / "fullsize_report.json").write_text(json.dumps(rep, indent=1))
        print(json.dumps(rep, indent=1))
    return 0


if __name__ == "__main__":
    sys.exit(main())
