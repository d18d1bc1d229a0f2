#!/usr/bin/env python3
"""Diagnostic: where does a block of the fp32 LDS-DMA GEMM spend its life?  (GPU box only; not a product path)

Builds gemm_f32.hip with -DAVD_GEMM_STAMPS into tools/micro/libgemm_stamps.so (done by the caller, see --build), runs the C3
projection shapes and reads the per-block stamps: s_memtime / s_memrealtime at kernel entry, main-loop start, main-loop end,
block end, plus HW_ID / XCC_ID.  Prints, per shape: kernel time, effective clock, share of block lifetime in prologue / main
loop / epilogue, main-loop cycles per K tile against the MFMA-issue floor, and blocks resident per CU over time."""
import argparse
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--rows", type=int, default=64 * 421)
ap.add_argument("--tile", type=int, default=-1)
ap.add_argument("--dump-map", action="store_true")
ap.add_argument("--stages", type=int, default=2)
ap.add_argument("--variant", default="", help="extra -D flags for the diagnostic build, e.g. AVD_LAB_NODMA or AVD_LAB_NODMA,AVD_LAB_NOLDS")
args = ap.parse_args()
so = HERE / ("libgemm_stamps" + ("_" + args.variant.replace(",", "_") if args.variant else "") + ".so")
if args.build or not so.exists():
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-DAVD_GEMM_STAMPS"] +
                   ["-D" + v for v in args.variant.split(",") if v] + [
                    "-o", str(so), str(ROOT / "multimodal_diffusion_amd/csrc/gemm_f32.hip"), str(HERE / "lab_stub.hip")], check=True)
    if args.build and not torch.cuda.is_available():
        sys.exit(0)
lib = C.CDLL(str(so))
P, I, L, F = C.c_void_p, C.c_int, C.c_int64, C.c_float
lib.avd_gemm_rmsfold_f32.argtypes = [P, P, P, P, P, L, I, I, I, P, I, F, P, P]
lib.lab_set_dbg.argtypes = [P]
lib.lab_set_tile.argtypes = [I]
dev = torch.device("cuda:0")
M = args.rows
lib.lab_set_tile(args.tile)
lib.lab_set_stages.argtypes = [I]
lib.lab_set_stages(args.stages)
g = torch.Generator().manual_seed(0)
shapes = [("in_proj", 1536, 512, 0, False, False), ("out_proj", 512, 512, 0, True, True), ("fc1+gelu", 2048, 512, 1, False, False),
          ("fc2+res", 512, 2048, 0, True, True)]
for name, N, K, act, res, ssout in shapes:
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    r = torch.randn(M, N, generator=g).to(dev) if res else None
    y = torch.empty(M, N, device=dev)
    ss = torch.empty(M, N // 32, device=dev) if ssout else None
    nblk_max = ((M + 63) // 64) * ((N + 63) // 64)
    dbg = torch.zeros(nblk_max * 16, dtype=torch.int64, device=dev)
    lib.lab_set_dbg(dbg.data_ptr())

    def call():
        rc = lib.avd_gemm_rmsfold_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), None if r is None else r.data_ptr(), y.data_ptr(), M, N, K,
                                      act, None, 0, 1e-6, None if ss is None else ss.data_ptr(), None)
        assert rc == 0
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    d = dbg.cpu().numpy().reshape(-1, 16)
    d = d[d[:, 0] != 0]
    t = d[:, 0:8:2].astype(np.float64)          # memtime at the 4 stamps
    rt = d[:, 1:8:2].astype(np.float64)         # realtime (100 MHz)
    nb = len(d)
    life = t[:, 3] - t[:, 0]
    clk = np.median(life / np.maximum(rt[:, 3] - rt[:, 0], 1)) * 100e6
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    nk = K // 32
    hw = d[:, 8]
    if args.dump_map and name == "in_proj":
        ids = np.nonzero(dbg.cpu().numpy().reshape(-1, 16)[:, 0])[0]
        cu0 = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 8) & 15))
        first = {}
        for b, c in zip(ids[:1024].tolist(), cu0[:1024].tolist()):
            first.setdefault(c, []).append(b)
        print("   first blocks per CU (sample):", list(first.items())[:6], "| CUs whose first two blocks are b, b+256:",
              sum(1 for v in first.values() if len(v) > 1 and v[1] - v[0] == 256), "of", len(first))
    cu = ((hw >> 32) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 8) & 15))      # xcc | se | cu
    span = (rt[:, 3].max() - rt[:, 0].min()) / 100.0       # us
    # floor: the block's own MFMAs at 64 cycles each per wave
    print(f"{name:9s} N={N} K={K}: {us:7.1f} us {2*M*N*K/us/1e6:6.1f} TF | blocks {nb} on {len(set(cu.tolist()))} CUs, span {span:6.1f} us, "
          f"clock {clk/1e9:4.2f} GHz | life {np.median(life):8.0f} cyc: prologue {np.median(pro):6.0f} loop {np.median(loop):8.0f} "
          f"({np.median(loop)/nk:6.0f}/Ktile) epilogue {np.median(epi):6.0f}", flush=True)
    # per-CU occupancy: total block-lifetime / (CUs * span)
    occ = (rt[:, 3] - rt[:, 0]).sum() / 100.0 / (len(set(cu.tolist())) * span)
    first = rt[:, 0].min()
    late = np.sort(rt[:, 3] - first) / 100.0
    print(f"          mean resident blocks per CU {occ:4.2f}; block end times (us) p50 {late[nb//2]:6.1f} p90 {late[int(nb*0.9)]:6.1f} "
          f"p99 {late[int(nb*0.99)]:6.1f} max {late[-1]:6.1f}; loop cyc/Ktile p10 {np.percentile(loop, 10)/nk:6.0f} p90 {np.percentile(loop, 90)/nk:6.0f}", flush=True)
