#!/usr/bin/env python3
"""Runs one bf16x3 GEMM shape repeatedly (for rocprofv3 --pmc passes on the GPU box)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from multimodal_diffusion_amd import functional as Fn, _lib as L   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=64 * 421)
ap.add_argument("--n", type=int, default=1536)
ap.add_argument("--k", type=int, default=512)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--mode", default="plain", choices=["plain", "res", "gelu"])
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(a.rows, a.k, generator=g).to(dev)
w = (torch.randn(a.n, a.k, generator=g) / a.k ** 0.5).to(dev)
b = torch.randn(a.n, generator=g).to(dev)
r = torch.randn(a.rows, a.n, generator=g).to(dev) if a.mode == "res" else None
x3, w3 = Fn.split3(x), Fn.split3(w)
for _ in range(a.iters):
    Fn.linear_bf16x3(x3, a.rows, w3, a.n, a.k, bias=b, residual=r, act=L.ACT_GELU if a.mode == "gelu" else L.ACT_NONE,
                     out_split3=a.mode == "gelu")
torch.cuda.synchronize()
print("done")
