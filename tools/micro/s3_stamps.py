#!/usr/bin/env python3
"""Diagnostic: where does a block of the split-operand GEMMs (gemm_bf16x3.hip) spend its K loop?  (GPU box only; not a product path)

tools/micro/libs3_stamps.so = gemm_bf16x3.hip built with -DAVD_S3_STAMPS (+ lab_stub.hip).  Every block stamps (core clock + 100 MHz
real time) its entry / loop start / loop end / exit — no stamp inside the K loop.  Printed per shape and mode: kernel time, in-kernel
clock, share of block lifetime in prologue / loop / epilogue, cycles per K step against the MFMA-issue floor.
--variant AVD_LAB_NODMA / AVD_LAB_NOLDS: diagnostic builds without the in-loop DMA / without the fragment reads (wrong results by design)."""
import argparse
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--rows", type=int, default=64 * 421)
ap.add_argument("--warm-s", type=float, default=1.0)
ap.add_argument("--modes", default="f16x2,bf16x3,bf16")
ap.add_argument("--variant", default="", help="extra -D flags for the diagnostic build, e.g. AVD_LAB_HALFLDS")
args = ap.parse_args()
so = HERE / ("libs3_stamps" + ("_" + args.variant.replace(",", "_") if args.variant else "") + ".so")
if args.build or not so.exists():
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-DAVD_S3_STAMPS"] +
                   ["-D" + v for v in args.variant.split(",") if v] + ["-o", str(so),
                    str(ROOT / "multimodal_diffusion_amd/csrc/gemm_bf16x3.hip"), str(HERE / "lab_stub.hip")], check=True)
lib = C.CDLL(str(so))
P, I, L, F = C.c_void_p, C.c_int, C.c_int64, C.c_float
lib.avd_split_f16x2_f32.argtypes = [P, P, L, I, F, P]
lib.avd_split3_f32.argtypes = [P, P, L, I, P]
lib.avd_split3_bytes.restype = L
lib.avd_split3_bytes.argtypes = [L, I]
lib.avd_gemm_f16x2_f32.argtypes = [P, P, P, P, P, P, L, I, I, I, F, F, P]
lib.avd_gemm_bf16x3_f32.argtypes = [P, P, P, P, P, P, L, I, I, I, I, P]
lib.lab_set_dbg.argtypes = [P]
dev = torch.device("cuda:0")
M = args.rows
g = torch.Generator().manual_seed(0)
shapes = [("out_proj (256x256 tiles, +res)", 512, 512, "res"), ("fc2 (256x256 tiles, +res)", 512, 2048, "res"),
          ("fc1 (256x128 tiles, GELU->image)", 2048, 512, "gelu")]
for mode, terms in (("f16x2", 3), ("bf16x3", 6), ("bf16", 1)):
    if mode not in args.modes.split(","):
        continue
    mfma_per_step = {3: 24, 6: 48, 1: 8}[terms] * 32        # cycles of matrix-pipe issue per wave per 16-k step (128 x 64 wave tile)
    for name, N, K, epi in shapes:
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        r = torch.randn(M, N, generator=g).to(dev)
        x3 = torch.empty(lib.avd_split3_bytes(M, K), dtype=torch.uint8, device=dev)
        w3 = torch.empty(lib.avd_split3_bytes(N, K), dtype=torch.uint8, device=dev)
        if terms == 3:
            lib.avd_split_f16x2_f32(x.data_ptr(), x3.data_ptr(), M, K, 1024.0, None)
            lib.avd_split_f16x2_f32(w.data_ptr(), w3.data_ptr(), N, K, 65536.0, None)
        else:
            lib.avd_split3_f32(x.data_ptr(), x3.data_ptr(), M, K, None)
            lib.avd_split3_f32(w.data_ptr(), w3.data_ptr(), N, K, None)
        y = torch.empty(M, N, device=dev)
        y3 = torch.empty(lib.avd_split3_bytes(M, N), dtype=torch.uint8, device=dev)
        nblk = 4096
        dbg = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
        lib.lab_set_dbg(dbg.data_ptr())

        def run():
            if terms == 3:
                if epi == "gelu":
                    return lib.avd_gemm_f16x2_f32(x3.data_ptr(), w3.data_ptr(), b.data_ptr(), None, None, y3.data_ptr(), M, N, K, 1, 1024.0 * 65536.0, 256.0, None)
                return lib.avd_gemm_f16x2_f32(x3.data_ptr(), w3.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), None, M, N, K, 0, 1024.0 * 65536.0, 1.0, None)
            if epi == "gelu":
                return lib.avd_gemm_bf16x3_f32(x3.data_ptr(), w3.data_ptr(), b.data_ptr(), None, None, y3.data_ptr(), M, N, K, 1, terms, None)
            return lib.avd_gemm_bf16x3_f32(x3.data_ptr(), w3.data_ptr(), b.data_ptr(), r.data_ptr(), y.data_ptr(), None, M, N, K, 0, terms, None)

        # back-to-back launches for ~1 s first: the clock a kernel holds under sustained load is what is being measured
        import time
        t_w = time.time()
        while time.time() - t_w < args.warm_s:
            for _ in range(50):
                assert run() == 0
            torch.cuda.synchronize()
        dbg.zero_()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        assert run() == 0
        e.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(e) * 1e3
        d = dbg.cpu().numpy().reshape(nblk, 16)
        d = d[d[:, 8] > 0]
        life = (d[:, 3] - d[:, 0]).astype(np.float64)
        pro, loop, epil = (d[:, 1] - d[:, 0]) / life, (d[:, 2] - d[:, 1]) / life, (d[:, 3] - d[:, 2]) / life
        steps = d[:, 8].astype(np.float64)
        rt = (d[:, 9] - d[:, 10]).astype(np.float64) / 100e6      # s_memrealtime: 100 MHz, block entry -> exit
        ghz = float(np.median(life / 1e9 / np.maximum(rt, 1e-9)))
        per_step = float(((d[:, 2] - d[:, 1]) / steps).mean())
        drain = float(((d[:, 3] - d[:, 4]) / life).mean())        # share of the block's life between the last store's issue and its completion
        print(f"{mode:6s} {name:34s} {us:7.1f} us  blocks {len(d):4d}  clock ~{ghz:.2f} GHz | block life: prologue {pro.mean():.2f} loop {loop.mean():.2f} "
              f"epilogue {epil.mean():.2f} of which store drain {drain:.2f} ({life.mean():.0f} cyc) | {per_step:6.0f} cyc per K step (MFMA issue floor {mfma_per_step} per wave, x2 waves per SIMD = {2 * mfma_per_step})")
