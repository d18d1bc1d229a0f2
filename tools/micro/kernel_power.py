#!/usr/bin/env python3
"""Which kernels of the C3 step sit on the socket power cap?  Each of the step's four big launches is looped alone for ~2.5 s while a
thread samples this GPU's shader clock (pp_dpm_sclk) and socket power (hwmon) — bench.py's PowerClockSampler.  GPU box only."""
import argparse
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from bench import PowerClockSampler  # noqa: E402
from multimodal_diffusion_amd import functional as Fn, _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.5)
ap.add_argument("--rows", type=int, default=26944)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = L.lib()
M, TOK, H = args.rows, 421, 8
g = torch.Generator().manual_seed(0)


def image(rows, k):
    return Fn.split3(torch.randn(rows, k, generator=g).to(dev))


x512, x2048 = image(M, 512), image(M, 2048)
w_fc1, w_in, w_out, w_fc2 = image(2048, 512), image(1536, 512), image(512, 512), image(512, 2048)
b2048, b1536, b512 = (torch.randn(n, generator=g).to(dev) for n in (2048, 1536, 512))
res = torch.randn(M, 512, generator=g).to(dev)
q3 = torch.empty(int(lib.avd_qkv3_bytes(M // TOK, TOK, H)), dtype=torch.uint8, device=dev)
L.check(lib.avd_gemm_bf16x3_qkv3_f32(x512.data_ptr(), w_in.data_ptr(), b1536.data_ptr(), q3.data_ptr(), M, TOK, H, 512, 0.18, 6, L.stream_ptr(dev)))
o3 = torch.empty_like(x512)
cases = [
    ("fc1 (GELU -> image)", 2.0 * M * 2048 * 512, lambda: Fn.linear_bf16x3(x512, M, w_fc1, 2048, 512, bias=b2048, act=L.ACT_GELU, out_split3=True)),
    ("in_proj (-> q|k|v image)", 2.0 * M * 1536 * 512, lambda: L.check(lib.avd_gemm_bf16x3_qkv3_f32(x512.data_ptr(), w_in.data_ptr(), b1536.data_ptr(), q3.data_ptr(), M, TOK, H, 512, 0.18, 6, L.stream_ptr(dev)))),
    ("fc2 (+ residual)", 2.0 * M * 512 * 2048, lambda: Fn.linear_bf16x3(x2048, M, w_fc2, 512, 2048, bias=b512, residual=res)),
    ("out_proj (+ residual)", 2.0 * M * 512 * 512, lambda: Fn.linear_bf16x3(x512, M, w_out, 512, 512, bias=b512, residual=res)),
    ("attention (bf16x3, 421 tokens)", 4.0 * (M // TOK) * H * TOK * TOK * 64, lambda: L.check(lib.avd_attn_fwd_qkv3_f32(q3.data_ptr(), None, o3.data_ptr(), M // TOK, TOK, H, TOK, 6, L.stream_ptr(dev)))),
]
for name, flops, fn in cases:
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s = PowerClockSampler(0)
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    r = s.stop(settle_s=1.0)
    print(f"{name:34s} {dt / n * 1e6:8.1f} us  {flops / (dt / n) / 1e12:6.1f} TFLOP/s fp32-equivalent | sclk {r.get('sclk_mhz_median')} MHz, socket {r.get('socket_power_w_median')} W of {r.get('socket_power_cap_w')} W")
    time.sleep(1.0)
