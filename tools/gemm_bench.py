#!/usr/bin/env python3
"""Per-shape timing of the fp32 MFMA GEMM and attention kernels at the C3 workload's shapes (GPU box only)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from multimodal_diffusion_amd import functional as Fn, _lib as L   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rows", type=int, default=64 * 421)
ap.add_argument("--attn", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
M = args.rows
shapes = [("qkv", 1536, 512, L.ACT_NONE, False), ("out_proj", 512, 512, L.ACT_NONE, True),
          ("fc1+gelu", 2048, 512, L.ACT_GELU, False), ("fc2+res", 512, 2048, L.ACT_NONE, True),
          ("fc1 noact", 2048, 512, L.ACT_NONE, False)]
g = torch.Generator().manual_seed(0)
for name, N, K, act, res in shapes:
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if res else None
    for _ in range(3):
        Fn.linear(x, w, b, act=act, residual=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        Fn.linear(x, w, b, act=act, residual=r)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"{name:10s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
if args.attn:
    qkv = torch.randn(64, 421, 1536, generator=g).to(dev)
    for _ in range(3):
        Fn.attention(qkv, 8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        Fn.attention(qkv, 8)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    print(f"attention B=64 N=421 H=8: {ms*1e3:8.1f} us  {4*64*8*421*421*64/ms/1e9:7.1f} TFLOP/s", flush=True)
