#!/usr/bin/env python3
"""VideoVAE.decode timing at the shipped geometry (GPU box): z [B,8,12,S/8,S/8] -> [B,3,48,S,S]."""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import multimodal_diffusion_amd as A            # noqa: E402
from multimodal_diffusion_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--matmul", default="f32", choices=["f32", "bf16x3", "f16x2"])
ap.add_argument("--lat", type=int, default=1, help="0: from_lat -> upsample -> 64-channel first conv (rounds 1-4); 1: the latent-composed first conv")
ap.add_argument("--encode", action="store_true", help="time VideoVAE.encode of [B,3,48,size,size] instead of the decode")
ap.add_argument("--power", type=float, default=0.0, help="loop the decode for this many seconds under bench.py's clock / socket-power sampler")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
vae = A.VideoVAE.from_config({"latent": {"channels": 8, "t_down": 4, "s_down": 8}}).eval().to(dev)
vae.matmul = args.matmul
vae.lat_composed = bool(args.lat)
z = torch.randn(args.batch, 8, 12, args.size // 8, args.size // 8, device=dev)
if args.encode:
    xin = torch.rand(args.batch, 3, 48, args.size, args.size, device=dev) * 2 - 1
    for _ in range(2):
        zz = vae.encode(xin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        zz = vae.encode(xin)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    print(f"[{args.matmul}] encode B={args.batch} {args.size}x{args.size}: {dt*1e3:.2f} ms, out {tuple(zz.shape)} finite={bool(torch.isfinite(zz).all())}")
    L.prof_enable(True)
    vae.encode(xin)
    torch.cuda.synchronize()
    L.prof_enable(False)
    for k, (n, ms, w) in L.prof_report().items():
        if n:
            unit = f"{w/ms/1e9:8.1f} TFLOP/s" if "conv3d" in k else f"{w/ms/1e6:8.1f} GB/s"
            print(f"  {k:40s} x{n}  {ms/n:9.3f} ms  {unit}")
    sys.exit(0)
for _ in range(2):
    x = vae.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    x = vae.decode(z)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.iters
vox = args.batch * 48 * args.size * args.size
fl = 2 * 2.0 * vox * 64 * 27 * 64
print(f"[{args.matmul}{'' if args.lat else ', lat 0'}] decode B={args.batch} {args.size}x{args.size}: {dt*1e3:.2f} ms  ({fl/dt/1e12:.1f} TFLOP/s counting the reference's two 64-channel 3x3x3 convs), "
      f"out {tuple(x.shape)} finite={bool(torch.isfinite(x).all())}")
L.prof_enable(True)
vae.decode(z)
torch.cuda.synchronize()
L.prof_enable(False)
for k, (n, ms, w) in L.prof_report().items():
    if n:
        unit = f"{w/ms/1e9:8.1f} TFLOP/s" if "conv3d" in k else f"{w/ms/1e6:8.1f} GB/s"
        print(f"  {k:40s} x{n}  {ms/n:9.3f} ms  {unit}")
if args.power > 0:
    from bench import PowerClockSampler
    smp = PowerClockSampler(0)
    smp.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.power:
        for _ in range(5):
            vae.decode(z)
        torch.cuda.synchronize()
        n += 5
    dt = (time.perf_counter() - t0) / n
    r = smp.stop(settle_s=1.0)
    print(f"  looped {n} decodes: {dt*1e3:.2f} ms each | sclk {r.get('sclk_mhz_median')} MHz, socket {r.get('socket_power_w_median')} W of {r.get('socket_power_cap_w')} W")
