#!/usr/bin/env python3
"""CPU baselines BASELINE.md section 3 asks for, on the host cores of the box this runs on: the oracle (oracle/ref_cpu.py, the
from-scratch torch port of the reference step, pinned to the reference by tests/golden) timed at the C1, C2 and C3 shapes.

  C1  32x32,  B=4,  the WHOLE 50-step DDIM trajectory (BASELINE configs[0]: "50 DDIM steps, batch=4 on CPU reference path")
  C2  64x64,  B=32, 1 warm-up + 3 timed steps
  C3  256x256, B=32, 1 warm-up + 3 timed steps

    python tools/cpu_baselines.py --out profiles/r02_cpu_baselines.json
"""
import argparse
import json
import os
import platform
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import host_cores, step_flops_per_sample   # noqa: E402
from oracle import ref_cpu as R                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="profiles/cpu_baselines.json")
ap.add_argument("--skip-c3", action="store_true")
args = ap.parse_args()
cores = host_cores()
torch.set_num_threads(cores)
ws = R.synth_weights(seed=0)
abar = R.alpha_bar_table(R.beta_table(1000))
kw = dict(adapt_v=ws["adapt_v"], adapt_a=ws["adapt_a"], core=ws["core"], head=ws["head"], n_layers=8, n_heads=8, guidance=3.5)
cpu = ""
try:
    cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except (OSError, IndexError):
    cpu = platform.processor()
res = {"host": {"cpu": cpu, "threads_used": cores, "torch": torch.__version__}, "configs": {}}


def inputs(size, B):
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, 8, 12, size // 8, size // 8, generator=g)
    za = torch.randn(B, 8, 150, generator=torch.Generator().manual_seed(2))
    return z, za


with torch.no_grad():
    # C1: whole trajectory
    z, za = inputs(32, 4)
    sched = R.sampling_schedule(1000, 50)
    R.denoise_step_a2v(z, za, sched[0].repeat(4), sched[1].repeat(4), abar, **kw)            # warm-up
    t0 = time.perf_counter()
    zf = R.sample_a2v(z, za, sched, abar, **kw)
    dt = time.perf_counter() - t0
    res["configs"]["C1 32x32 B=4 DDIM-50 trajectory"] = {"seconds": dt, "steps": 50, "ms_per_step": 1e3 * dt / 50, "steps_per_s": 50 / dt,
                                                        "final_latent_finite": bool(torch.isfinite(zf).all()),
                                                        "gflop_per_step": step_flops_per_sample(6, 37) * 4 / 1e9}
    print(json.dumps(res["configs"]), flush=True)
    for name, size, B in (("C2 64x64 B=32", 64, 32), ("C3 256x256 B=32", 256, 32)):
        if size == 256 and args.skip_c3:
            continue
        z, za = inputs(size, B)
        tn, tp = torch.full((B,), 999), torch.full((B,), 979)
        R.denoise_step_a2v(z, za, tn, tp, abar, **kw)
        t0 = time.perf_counter()
        for _ in range(3):
            R.denoise_step_a2v(z, za, tn, tp, abar, **kw)
        dt = (time.perf_counter() - t0) / 3
        nv = 6 * (size // 32) ** 2
        res["configs"][name] = {"timed_steps": 3, "ms_per_step": 1e3 * dt, "steps_per_s": 1 / dt, "sample_steps_per_s": B / dt,
                                "gflop_per_step": step_flops_per_sample(nv, 37) * B / 1e9,
                                "achieved_tflops": step_flops_per_sample(nv, 37) * B / dt / 1e12}
        print(json.dumps({name: res["configs"][name]}), flush=True)
out = ROOT / args.out
out.parent.mkdir(parents=True, exist_ok=True)
out.write_text(json.dumps(res, indent=1))
