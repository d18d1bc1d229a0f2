#!/usr/bin/env python3
"""K sweep of the split-operand GEMMs at the C3 row count: time(K) = rounds x (fixed + K/16 x per_step) separates a tile's fixed cost
(prologue + epilogue) from the main loop's cost per 16-k step.   python tools/micro/s3_ksweep.py [--rows 26944]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from multimodal_diffusion_amd import functional as Fn, _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=26944)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="", help="run only the epilogue kinds whose name contains this")
args = ap.parse_args()
dev = torch.device("cuda:0")
M = args.rows
g = torch.Generator().manual_seed(0)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(args.iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / args.iters * 1e3      # us


for mode in ("f16x2", "bf16x3", "bf16"):
    for epi, N in (("res", 512), ("gelu_split", 2048), ("qkv-like plain", 1536)):
        if args.only and args.only not in epi:
            continue
        res = []
        for K in (256, 512, 1024, 2048, 4096):
            x = torch.randn(M, K, generator=g).to(dev)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
            b = torch.randn(N, generator=g).to(dev)
            r = torch.randn(M, N, generator=g).to(dev) if epi == "res" else None
            if mode == "f16x2":
                x2, sx = Fn.split_f16x2(x)
                w2, sw = Fn.split_f16x2(w)
                if epi == "gelu_split":
                    fn = lambda: Fn.linear_f16x2(x2, M, w2, N, K, sx * sw, bias=b, act=L.ACT_GELU, out_scale=256.0)
                else:
                    fn = lambda: Fn.linear_f16x2(x2, M, w2, N, K, sx * sw, bias=b, residual=r)
            else:
                terms = 6 if mode == "bf16x3" else 1
                x3, w3 = Fn.split3(x), Fn.split3(w)
                if epi == "gelu_split":
                    fn = lambda: Fn.linear_bf16x3(x3, M, w3, N, K, bias=b, act=L.ACT_GELU, out_split3=True, terms=terms)
                else:
                    fn = lambda: Fn.linear_bf16x3(x3, M, w3, N, K, bias=b, residual=r, terms=terms)
            res.append((K, timed(fn)))
        (k0, t0), (k1, t1) = res[1], res[-1]
        per_step = (t1 - t0) / ((k1 - k0) / 16)
        fixed = t0 - per_step * k0 / 16
        print(f"{mode:7s} {epi:15s} N={N:5d}: " + "  ".join(f"K={k}: {t:7.1f}us" for k, t in res) +
              f"   -> launch fixed {fixed:6.1f} us, {per_step:.3f} us per 16-k step of the whole grid")
