#!/usr/bin/env python3
"""End-to-end audio->video generation for one batch on one GPU (GPU box): 50 DDIM + CFG steps at 256x256 (the bench.py
workload) followed by VideoVAE.decode of the whole batch — the part of sample_one_direction (sample_clip.py:220-394) that runs
on the device.  Prints the sampler / decode split for the matrix-pipe modes, then the other direction (video -> audio,
sample_clip.py:313-352): VideoVAE.encode of the batch of prompt clips, 50 steps on the audio latent, AudioCodec.decode."""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import multimodal_diffusion_amd as A                                   # noqa: E402
from multimodal_diffusion_amd import schedule_utils as su              # noqa: E402
import bench                                                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--sampler-steps", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, S, size = args.batch, args.sampler_steps, args.size
mods, tdim = bench.build_modules(dev)
av, aa, core, head = mods
torch.manual_seed(0)
vae = A.VideoVAE.from_config({"latent": {"channels": 8, "t_down": 4, "s_down": 8}}).eval().to(dev)
abar = su.alphas_cumprod_from_betas(su.make_beta_schedule(1000, "cosine", 1e-4, 0.02))[1]
sched = su.make_sampling_schedule(1000, S)
z0 = torch.randn(B, 8, 12, size // 8, size // 8, generator=torch.Generator().manual_seed(1)).to(dev)
za = torch.randn(B, 8, 150, generator=torch.Generator().manual_seed(2)).to(dev)
for mode in ("f32", "bf16x3", "f16x2"):
    vae.matmul = mode
    eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=tdim, target="video", latent_shape=tuple(z0.shape),
                          prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode)
    eng.set_prompt(za)
    z = eng.run(z0, sched[:3])                      # warm-up
    x = vae.decode(torch.randn_like(z))             # full-batch warm-up: the decoder's workspace (tens of GB) is allocated here
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z = eng.run(z0, sched)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    # the random-weight trajectory explodes numerically (|z| ~ 1e5); decode a unit-scale latent of the same shape
    x = vae.decode(torch.randn_like(z))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"[{mode:6s}] B={B} {size}x{size}: sampler {S} steps {1e3 * (t1 - t0):7.1f} ms ({1e3 * (t1 - t0) / S:.2f} ms/step)  "
          f"VAE decode {1e3 * (t2 - t1):7.1f} ms ({1e3 * (t2 - t1) / B:.2f} ms/sample)  total {1e3 * (t2 - t0):7.1f} ms  "
          f"= {B / (t2 - t0):.1f} clips/s   out {tuple(x.shape)}", flush=True)

# ---- video -> audio: encode the prompt clips, 50 steps on [B, 8, 150] with 384 prompt tokens, codec decode
codec = A.AudioCodec.from_config({"sr": 16000, "latent": {"channels": 8}, "codec": {"hop_samples": 320}}).eval().to(dev)
xv = (torch.rand(B, 3, 48, size, size, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(dev)
for mode in ("f32", "bf16x3", "f16x2"):
    vae.matmul = mode
    eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=tdim, target="audio", latent_shape=tuple(za.shape),
                          prompt_tokens=(12 // 2) * (size // 8 // 4) ** 2, alpha_bar=abar, guidance=3.5, matmul=mode)      # tube patches 2 x 4 x 4
    zp = vae.encode(xv)                              # warm-up (workspace)
    eng.set_prompt(zp)
    eng.run(za, sched[:3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    zp = vae.encode(xv)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    eng.set_prompt(zp)
    z = eng.run(za, sched)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t3 = t2
    if codec is not None:
        wav = codec.decode(torch.randn_like(z))
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print(f"[{mode:6s}] V->A B={B} {size}x{size}: VAE encode {1e3 * (t1 - t0):7.1f} ms ({1e3 * (t1 - t0) / B:.2f} ms/sample)  sampler {S} steps "
          f"{1e3 * (t2 - t1):7.1f} ms ({1e3 * (t2 - t1) / S:.2f} ms/step)  codec decode {1e3 * (t3 - t2):6.1f} ms  total {1e3 * (t3 - t0):7.1f} ms  "
          f"= {B / (t3 - t0):.1f} clips/s", flush=True)
