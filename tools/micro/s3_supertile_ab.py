#!/usr/bin/env python3
"""Interleaved A/B of the split GEMMs' block order (super-tile shape, first-generation stagger) on fc1 and in_proj at the C3 row count,
ONE process, N rounds (cdna_hip_programming.md rule 24): the times that belong beside profiles/r05_fetch_ab.txt's bytes.  GPU box only."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from multimodal_diffusion_amd import functional as Fn, _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=26944)
ap.add_argument("--rounds", type=int, default=9)
args = ap.parse_args()
dev = torch.device("cuda:0")
M, K = args.rows, 512
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).to(dev)
x3 = Fn.split3(x)
lib = L.lib()
lib.avd_tune_set.argtypes = [C.c_char_p, C.c_int64]


def tune(**kw):
    for k, v in kw.items():
        L.check(lib.avd_tune_set(k.encode(), v))


VARIANTS = [("default", dict(s3_sn=0, s3_super4=0, s3_stagger=-1)), ("stagger0", dict(s3_sn=0, s3_super4=0, s3_stagger=0)),
            ("sn4 (8x4)", dict(s3_sn=4, s3_super4=0, s3_stagger=-1)), ("sn8 (4x8)", dict(s3_sn=8, s3_super4=0, s3_stagger=-1)),
            ("super64 (4 rows)", dict(s3_sn=0, s3_super4=64, s3_stagger=-1)), ("super16 (1 row)", dict(s3_sn=0, s3_super4=16, s3_stagger=-1)),
            ("sn4 super64 (16x4)", dict(s3_sn=4, s3_super4=64, s3_stagger=-1)), ("sn2 super32 (16x2)", dict(s3_sn=2, s3_super4=0, s3_stagger=-1)),
            ("sn4 super128 (32x4)", dict(s3_sn=4, s3_super4=128, s3_stagger=-1)), ("sn4 super64 stagger0", dict(s3_sn=4, s3_super4=64, s3_stagger=0)),
            ("sn3 super48 (16x3)", dict(s3_sn=3, s3_super4=48, s3_stagger=-1)), ("sn6 super48 (8x6)", dict(s3_sn=6, s3_super4=48, s3_stagger=-1))]
TOK = 421
for name, N in (("fc1 (GELU -> image)", 2048), ("in_proj (-> q|k|v image, N = 1536)", 1536), ("head-like (bias -> image, N = 512)", 512)):
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    w3 = Fn.split3(w)
    if N == 1536 and M % TOK == 0:
        q3 = torch.empty(int(lib.avd_qkv3_bytes(M // TOK, TOK, 8)), dtype=torch.uint8, device=dev)
        fn = lambda: L.check(lib.avd_gemm_bf16x3_qkv3_f32(x3.data_ptr(), w3.data_ptr(), b.data_ptr(), q3.data_ptr(), M, TOK, 8, K, 0.18, 6, L.stream_ptr(dev)))
    else:
        fn = lambda: Fn.linear_bf16x3(x3, M, w3, N, K, bias=b, act=L.ACT_GELU if N == 2048 else L.ACT_NONE, out_split3=True)
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    res = {v[0]: [] for v in VARIANTS}
    for r in range(args.rounds):
        for vname, kw in VARIANTS:
            tune(**kw)
            for _ in range(3):
                fn()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            res[vname].append(a.elapsed_time(e) * 50.0)
    tune(s3_sn=0, s3_super4=0, s3_stagger=-1)
    print(f"== {name}, {M} rows: us per launch, median / min over {args.rounds} interleaved rounds of 20 launches")
    for vname, _ in VARIANTS:
        print(f"   {vname:22s} {np.median(res[vname]):7.1f} / {np.min(res[vname]):7.1f}")
