#!/usr/bin/env python3
"""Diagnostic: WHERE do the beyond-L2 bytes of fc1 / in_proj come from?  (GPU box only; not a product path)

VERDICT r4 item 1: fc1 (`gemm_bf16x3_m16_kernel<3,4,8,0>`) fetches 494 MB per launch for 89 MB of operands, in_proj (`<4,...>`) 308 for 88.
This tool answers with the stamps build of gemm_bf16x3.hip (-DAVD_S3_STAMPS): every block records its XCD, its CU slot, its tile and
the real-time stamps of its entry, K-loop start, K-loop end and exit.  From those:
  * the phase picture: how far apart in k the blocks that share an XCD's L2 are at any moment;
  * a trace-driven L2 model: every block's stage pieces (1 KiB, A and W) are replayed in time order through an LRU cache of 4 MiB per
    XCD; misses x 1 KiB = predicted bytes beyond L2, to set beside FETCH_SIZE (profiles/r04_traffic_bf16x3.json);
  * the same trace replayed with every block's K loop moved into lock-step per generation ("ideal"), and with other tile orders, says
    what a schedule fix could buy BEFORE anyone writes it.
--alias builds the AVD_LAB_ALIAS variants too (every block stages the same A / W panels: same instruction stream and L2 -> LDS bytes,
all L2 hits): the time difference against the normal build is the most that L2 residency of the panels can buy."""
import argparse
import ctypes as C
import subprocess
import time
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=64 * 421)
ap.add_argument("--warm-s", type=float, default=1.0)
ap.add_argument("--alias", action="store_true")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--l2-mib", type=float, default=4.0)
ap.add_argument("--build-only", action="store_true", help="cross-compile the diagnostic libraries (no GPU needed) and exit")
args = ap.parse_args()
P, I, L, F = C.c_void_p, C.c_int, C.c_int64, C.c_float


def so_path(variant):
    return HERE / ("libs3_phase" + ("_" + variant.replace("=", "") if variant else "") + ".so")


def compile_lib(variant):
    return subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-shared", "-DAVD_S3_STAMPS"] +
                   (["-D" + variant] if variant else []) + ["-o", str(so_path(variant)), str(ROOT / "multimodal_diffusion_amd/csrc/gemm_bf16x3.hip"),
                                                            str(HERE / "lab_stub.hip")])


def build(variant):
    if not so_path(variant).exists():
        assert compile_lib(variant).wait() == 0
    lib = C.CDLL(str(so_path(variant)))
    lib.avd_split3_f32.argtypes = [P, P, L, I, P]
    lib.avd_split3_bytes.restype = L
    lib.avd_split3_bytes.argtypes = [L, I]
    lib.avd_gemm_bf16x3_f32.argtypes = [P, P, P, P, P, P, L, I, I, I, I, P]
    lib.avd_gemm_bf16x3_qkv3_f32.argtypes = [P, P, P, P, L, I, I, I, F, I, P]
    lib.lab_set_dbg.argtypes = [P]
    return lib


VARIANTS = [""] + (["AVD_LAB_ALIAS=1", "AVD_LAB_ALIAS=2", "AVD_LAB_ALIAS=3"] if args.alias else [])
if args.build_only:
    procs = [compile_lib(v) for v in VARIANTS]
    assert all(p.wait() == 0 for p in procs)
    raise SystemExit(0)
dev = torch.device("cuda:0")
M = args.rows
gen = torch.Generator().manual_seed(0)
TOK = 421
NBLK = 8192


def setup(lib, N, K):
    x = torch.randn(M, K, generator=gen).to(dev)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=gen).to(dev)
    x3 = torch.empty(lib.avd_split3_bytes(M, K), dtype=torch.uint8, device=dev)
    w3 = torch.empty(lib.avd_split3_bytes(N, K), dtype=torch.uint8, device=dev)
    lib.avd_split3_f32(x.data_ptr(), x3.data_ptr(), M, K, None)
    lib.avd_split3_f32(w.data_ptr(), w3.data_ptr(), N, K, None)
    return x3, w3, b


def runner(lib, kind, N, K):
    x3, w3, b = setup(lib, N, K)
    if kind == "fc1":
        y3 = torch.empty(lib.avd_split3_bytes(M, N), dtype=torch.uint8, device=dev)
        return lambda: lib.avd_gemm_bf16x3_f32(x3.data_ptr(), w3.data_ptr(), b.data_ptr(), None, None, y3.data_ptr(), M, N, K, 1, 6, None), (x3, w3, b, y3)
    heads = N // 192
    npad = (TOK + 63) // 64 * 64
    q3 = torch.empty(3 * (M // TOK) * heads * npad * 384 + 4096, dtype=torch.uint8, device=dev)
    return lambda: lib.avd_gemm_bf16x3_qkv3_f32(x3.data_ptr(), w3.data_ptr(), b.data_ptr(), q3.data_ptr(), M, TOK, heads, K, 0.18, 6, None), (x3, w3, b, q3)


def timed(run, reps):
    t_w = time.time()
    while time.time() - t_w < args.warm_s:
        for _ in range(30):
            assert run() == 0
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            assert run() == 0
        e.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 100.0)
    return float(np.median(ts)), float(np.min(ts))


def lru_replay(events, cap_pieces):
    """events: time-ordered list of (xcd, [piece ids]); returns misses per XCD summed (pieces)"""
    caches = {}
    miss = 0
    for xcd, pcs in events:
        c = caches.setdefault(xcd, OrderedDict())
        for p in pcs:
            if p in c:
                c.move_to_end(p)
            else:
                miss += 1
                c[p] = True
                if len(c) > cap_pieces:
                    c.popitem(last=False)
    return miss


def analyse(lib, kind, N, K):
    run, keep = runner(lib, kind, N, K)
    dbg = torch.zeros(NBLK * 16, dtype=torch.int64, device=dev)
    lib.lab_set_dbg(dbg.data_ptr())
    med, mn = timed(run, args.reps)
    dbg.zero_()
    assert run() == 0
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(NBLK, 16)
    d = d[d[:, 8] > 0]
    nk = int(d[0, 8])
    # per block: linear map core clock -> real time (100 MHz) from the entry / exit stamp pairs
    t0c, t3c = d[:, 0].astype(np.float64), d[:, 3].astype(np.float64)
    t0r, t3r = d[:, 10].astype(np.float64), d[:, 9].astype(np.float64)
    scale = (t3r - t0r) / np.maximum(t3c - t0c, 1.0)
    base = t0r.min()
    ent = (t0r - base) / 100.0                                   # us
    lo = (t0r + (d[:, 1] - t0c) * scale - base) / 100.0          # loop start
    le = (t0r + (d[:, 2] - t0c) * scale - base) / 100.0          # loop end
    ex = (t3r - base) / 100.0
    xcd = (d[:, 11] & 15).astype(np.int64)
    hw = (d[:, 11] >> 8).astype(np.int64)
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    bm = (d[:, 12] & 0xffffffff).astype(np.int64)
    bn = (d[:, 12] >> 32).astype(np.int64)
    span = float(ex.max())
    print(f"== {kind}: M {M} N {N} K {K}: {len(d)} blocks, launch {med:.1f} us median / {mn:.1f} min (10-launch batches), stamped launch spans {span:.1f} us")
    print(f"   block life {np.mean(ex - ent):.1f} us (loop {np.mean(le - lo):.1f}, epilogue {np.mean(ex - le):.1f}); blocks per XCD {np.bincount(xcd, minlength=8).tolist()}")
    # phase picture: at sample times, the k positions of the blocks of one XCD that are inside their loop
    spreads, inloop, inepi = [], [], []
    for t in np.linspace(0.15 * span, 0.85 * span, 40):
        for x in range(8):
            sel = (xcd == x) & (lo <= t) & (le > t)
            if sel.sum() >= 2:
                kpos = (t - lo[sel]) / (le[sel] - lo[sel]) * nk
                spreads.append(np.std(kpos))
                inloop.append(sel.sum())
            inepi.append(((xcd == x) & (le <= t) & (ex > t)).sum())
    print(f"   per XCD at a time: {np.mean(inloop):.1f} blocks inside their K loop, {np.mean(inepi):.1f} in their epilogue; std of their k position {np.mean(spreads):.1f} of {nk} steps"
          f" (uniformly random would be {nk / 12 ** 0.5:.1f})")
    # L2 model: A piece ids (row group of 128, k, plane-quarter) / W piece ids; one event per (block, k step) at its interpolated time
    nA = 24 if kind else 24
    BMr = 256
    ev = []
    for i in range(len(d)):
        ts_ = lo[i] + (le[i] - lo[i]) * (np.arange(nk) + 0.0) / nk
        for k in range(nk):
            a_ids = [("A", bm[i] * 2 + r, k, q) for r in range(2) for q in range(12)]
            w_ids = [("W", bn[i], k, q) for q in range(12)]
            ev.append((ts_[k], xcd[i], a_ids + w_ids))
    cap = int(args.l2_mib * 1024)
    ev.sort(key=lambda e: e[0])
    total_pieces = sum(len(e[2]) for e in ev)
    miss = lru_replay([(e[1], e[2]) for e in ev], cap)
    print(f"   L2 model (LRU, {args.l2_mib} MiB per XCD, 1-KiB pieces, measured block times): L2->LDS {total_pieces / 1e6 * 1.024:.0f} MB, beyond L2 {miss * 1024 / 1e6:.0f} MB")
    for mib in (2.0, 3.0, 8.0, 16.0):
        print(f"      ... with {mib} MiB: {lru_replay([(e[1], e[2]) for e in ev], int(mib * 1024)) * 1024 / 1e6:.0f} MB")
    # ideal lock-step: same (xcd, tile) assignment, blocks of an XCD ordered by entry, generations of 64 walk k together
    ev2 = []
    for x in range(8):
        idx = np.where(xcd == x)[0]
        idx = idx[np.argsort(ent[idx])]
        for gi in range(0, len(idx), 64):
            for k in range(nk):
                for i in idx[gi:gi + 64]:
                    ev2.append((x, [("A", bm[i] * 2 + r, k, q) for r in range(2) for q in range(12)] + [("W", bn[i], k, q) for q in range(12)]))
    print(f"   ... the same blocks in lock-step generations of 64 per XCD: beyond L2 {lru_replay(ev2, cap) * 1024 / 1e6:.0f} MB")
    uniq = len({p for e in ev for p in e[2]})
    print(f"   ... unique operand bytes {uniq * 1024 / 1e6:.0f} MB")
    return med


for variant in VARIANTS:
    lib = build(variant)
    print(f"#### build: stamps {variant or '(normal operands)'}")
    if variant:
        for kind, N in (("fc1", 2048), ("in_proj", 1536)):
            run, keep = runner(lib, kind, N, 512)
            dbg = torch.zeros(NBLK * 16, dtype=torch.int64, device=dev)
            lib.lab_set_dbg(dbg.data_ptr())
            med, mn = timed(run, args.reps)
            print(f"== {kind}: launch {med:.1f} us median / {mn:.1f} min")
    else:
        analyse(lib, "fc1", 2048, 512)
        analyse(lib, "in_proj", 1536, 512)
