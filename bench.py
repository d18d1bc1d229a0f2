#!/usr/bin/env python3
"""Headline benchmark: denoising steps/sec @256x256 multimodal-cond, batch=32 per GPU, on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one CFG denoising step for a batch of 32 samples (reference loop body
avdiff/models/infer/sample_clip.py:359-389: tokenise -> adapters -> t-emb -> MMDiT x2 (cond+null, stacked to 2B) ->
noise head -> CFG -> un-patch -> DDIM), workload C3 of BASELINE.json (256x256 -> 384 video + 37 audio tokens,
mvp.yaml model dims d=512 L=8 H=8), fp32, synthetic inputs, random-init weights, inputs resident in HBM.
Weak scaling: every rank steps its own 32 samples; the only collective is one RCCL broadcast of the
conditioning latents before the loop.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch

def host_cores() -> int:
    """CPU threads this process may actually use (cgroup quota / affinity), not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        q, p = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("AVD_CPU_THREADS", min(n, 16 * max(1, torch.cuda.device_count()) if n > 64 else n)))


PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0
# bf16x3 matmul: one fp32-accurate product term set costs 6 bf16 MFMAs, so the dense bf16 peak (2,516.6 TFLOP/s) prices
# ALGORITHMIC fp32 flops at 2516.6 / 6 (the chip holds well under 2.4 GHz on this load; that is not priced in)
PEAK_BF16_MATRIX_TFLOPS = 2516.6
PEAK_BF16X3_EQUIV_TFLOPS = PEAK_BF16_MATRIX_TFLOPS / 6.0


def pmc_traffic(kernel_tag: str):
    """HBM-side bytes per launch of `kernel_tag` from the newest committed PMC pass (tools/pmc_traffic.py), or None.
    The PMC passes cannot run inside this process (rocprofv3 wraps the command), so the figure is read back."""
    for f in sorted((ROOT / "profiles").glob("r*_traffic*.json"), reverse=True):      # newest round first
        try:
            ks = json.loads(f.read_text())["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for name, v in ks.items():
            if kernel_tag in name:
                return v["traffic_bytes_per_launch"]
    return None


def step_flops_per_sample(nv: int, na: int, d: int = 512, L: int = 8, hid: int = 2048, tok: int = 256,
                          head_hidden: int = 512, tdim: int = 256) -> float:
    """Algorithmic FLOPs of one CFG step for one sample (head on target rows only)."""
    n = nv + na
    per_fwd = L * n * (2 * d * 3 * d + 2 * d * d + 2 * 2 * d * hid + 4 * n * d)
    head = nv * (2 * d * head_hidden + 2 * 2 * head_hidden * head_hidden + 2 * head_hidden * tok)
    adapt = nv * 2 * tok * (d - tdim)
    return 2 * (per_fwd + head) + adapt


def build_modules(device, seed=0):
    import multimodal_diffusion_amd as A
    torch.manual_seed(seed)
    cfg = {
        "tokenizer": {"width": 512},
        "embeddings": {"timestep_dim": 256},
        "model": {"core": dict(d_model=512, n_layers=8, n_heads=8, mlp_ratio=4.0, dropout=0.1, attn_dropout=0.0,
                               norm="rmsnorm", rope=False, token_dropout=0.0),
                  "heads": {"video": dict(out_dim=256, hidden_dim=512, activation="gelu"),
                            "audio": dict(out_dim=32, hidden_dim=512, activation="gelu")}},
    }
    _, _, av, aa, core, head, tdim = A.build_components(cfg, torch.device("cpu"))
    # non-trivial biases / norm scales so nothing is skipped by zeros
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in (core, head):
            for name, p in mod.named_parameters():
                if p.dim() == 1:
                    p.add_(0.02 * torch.randn(p.shape, generator=g))
    mods = [m.to(device).eval() for m in (av, aa, core, head)]
    return mods, tdim


def cpu_state(mods):
    av, aa, core, head = mods
    f = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return dict(adapt_v=f(av), adapt_a=f(aa), core=f(core), head=f(head))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=256, help="video size (square); 256 = BASELINE C3")
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--sampler-steps", type=int, default=50)
    ap.add_argument("--guidance", type=float, default=3.5)
    ap.add_argument("--graph", action="store_true", help="replay a captured 2-step HIP graph instead of eager launches")
    ap.add_argument("--split-streams", type=int, default=0, help="run the cond/null halves on two HIP streams")
    ap.add_argument("--matmul", default="f32", choices=["f32", "bf16x3"],
                    help="f32: fp32 MFMA everywhere; bf16x3: block projections on the bf16 matrix pipe with exactly split "
                         "fp32 operands (same fp32-level error, see csrc/gemm_bf16x3.hip)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra (untimed-region) measurement of the other matmul mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse N > 1 on a 1-GPU box)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo)")
    args = ap.parse_args()

    from multimodal_diffusion_amd import dist as D, schedule_utils as su, _lib as L
    import multimodal_diffusion_amd as A

    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if multi and args.backend == "nccl" and not args.share_device:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local = D.init_from_env(args.backend if multi else None)
    if args.share_device:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    B, size, S = args.batch, args.size, args.sampler_steps
    lat = (B, 8, 12, size // 8, size // 8)
    nv, na = 6 * (size // 32) ** 2, 37
    mods, tdim = build_modules(dev)
    av, aa, core, head = mods
    abar = su.alphas_cumprod_from_betas(su.make_beta_schedule(1000, "cosine", 1e-4, 0.02))[1]
    sched = su.make_sampling_schedule(1000, S)

    # conditioning: root draws the global batch of prompt latents, ONE broadcast, each rank keeps its shard
    gshape = (B * world, 8, 150)
    cond = torch.randn(gshape, generator=torch.Generator().manual_seed(2)) if rank == 0 else None
    cond_all = D.broadcast_conditioning(cond, gshape, dev if args.backend == "nccl" else torch.device("cpu")).to(dev)
    z_a0 = D.local_conditioning(cond_all, rank, world)
    z0 = torch.randn(lat, generator=torch.Generator().manual_seed(1 + rank)).to(dev)

    eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=tdim, target="video",
                          latent_shape=lat, prompt_tokens=na, alpha_bar=abar, guidance=args.guidance,
                          split_streams=bool(args.split_streams), matmul=args.matmul)
    eng.set_prompt(z_a0)
    eng.begin(sched)
    za, zb = z0.clone(), torch.empty_like(z0)

    graph = None
    if args.graph:
        eng.advance(za, zb)          # lazy init outside capture
        eng.rewind()
        za.copy_(z0)
        graph = eng.capture_pair(za, zb)

    state = {"i": 0}

    def run_steps(k):
        nonlocal za, zb
        done = 0
        while done < k:
            if state["i"] % S == 0:          # new trajectory: restart the schedule from fresh noise
                eng.rewind()
                za.copy_(z0)
            if graph is not None and (S - state["i"] % S) >= 2 and k - done >= 2:
                graph.replay()               # za -> zb -> za on the captured buffers
                state["i"] += 2
                done += 2
            elif graph is not None:          # odd step in graph mode: keep the captured buffer roles
                eng.advance(za, zb)
                za.copy_(zb)
                state["i"] += 1
                done += 1
            else:
                eng.advance(za, zb)
                za, zb = zb, za
                state["i"] += 1
                done += 1

    run_steps(args.warmup)
    state["i"] = 0
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps)
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev if args.backend == "nccl" else torch.device("cpu"))
    assert torch.isfinite(za).all(), "non-finite latent after the timed region"

    out = None
    if rank == 0:
        fl = step_flops_per_sample(nv, na) * B
        out = {
            "metric": "denoising steps/sec @256x256 multimodal-cond batch=32",
            "value": world * args.steps / dt,
            "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.matmul == "f32" else "f32 via 3xbf16 split operands (6-term products, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": f"C3: {size}x{size} audio->video CFG denoising step (cond+null MMDiT d512 L8 H8 + noise head + "
                                   f"DDIM), {nv}+{na} tokens, batch {B} per GPU, DDIM {S} steps, guidance {args.guidance}",
                       "global_batch": B * world, "tokens": nv + na, "sampler_steps": S,
                       "parallelism": f"dp{world}", "launch": "hipgraph" if args.graph else "eager", "matmul": args.matmul},
            "sample_steps_per_s": world * B * args.steps / dt,
            "algorithmic_tflops": fl * world * args.steps / dt / 1e12,
        }

    # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream, separate instrumented pass
    if rank == 0 and not args.no_roofline:
        L.prof_enable(True)
        state["i"] = 0
        saved = graph
        graph = None
        run_steps(5)
        graph = saved
        torch.cuda.synchronize()
        L.prof_enable(False)
        rep = L.prof_report()
        dom = max((k for k in rep if k.startswith("gemm")), key=lambda k: rep[k][1])   # most time => dominant
        n, ms, work = rep[dom]
        achieved = work / (ms * 1e-3) / 1e12
        peak = PEAK_BF16X3_EQUIV_TFLOPS if dom.startswith("gemm_bf16x3") else PEAK_F32_MATRIX_TFLOPS
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": peak,
                           "unit": "TFLOP/s", "frac": achieved / peak, "traffic": pmc_traffic(dom),
                           "launches": n, "avg_launch_us": 1e3 * ms / max(n, 1),
                           "flops_per_launch": work / max(n, 1)}
        tot_ms = sum(v[1] for v in rep.values())
        kern = {}
        for k, (cnt, kms, w) in rep.items():
            if cnt == 0:
                continue
            e = {"launches_per_step": cnt / 5, "ms_per_step": kms / 5, "share": kms / tot_ms}
            if k.startswith("gemm") or k.startswith("attn"):
                e["tflops"] = w / (kms * 1e-3) / 1e12
                if k.startswith("gemm_bf16x3"):
                    e["frac_of_bf16_mfma_peak_div6"] = e["tflops"] / PEAK_BF16X3_EQUIV_TFLOPS
                else:
                    e["frac_of_f32_mfma_peak"] = e["tflops"] / PEAK_F32_MATRIX_TFLOPS
            else:
                e["gbs"] = w / (kms * 1e-3) / 1e9
                e["frac_of_hbm_peak"] = e["gbs"] / PEAK_HBM_GBS
            kern[k] = e
        out["kernels"] = kern

    # ---- CPU baseline: the oracle (a from-scratch torch port of the reference step) on this host's cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_cpu as R
        cores = host_cores()
        torch.set_num_threads(cores)
        W = cpu_state(mods)
        zc = z0.cpu()
        zac = z_a0.cpu()
        tn = torch.full((B,), int(sched[0]))
        tp = torch.full((B,), int(sched[1]))
        kw = dict(adapt_v=W["adapt_v"], adapt_a=W["adapt_a"], core=W["core"], head=W["head"], n_layers=8, n_heads=8,
                  guidance=args.guidance)
        with torch.no_grad():
            ref = R.denoise_step_a2v(zc, zac, tn, tp, abar, **kw)           # warm-up, also the parity reference
            c0 = time.perf_counter()
            for _ in range(args.cpu_steps):
                R.denoise_step_a2v(zc, zac, tn, tp, abar, **kw)
            cdt = (time.perf_counter() - c0) / args.cpu_steps
        got = eng.step(z0, tn.to(dev), tp.to(dev)).cpu()
        err = float((got - ref).abs().max() / max(1.0, float(ref.abs().max())))
        cpu_ref = ref
        out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": f"{args.cpu_steps} timed steps (+1 warm-up) of the same batch-{B} {size}x{size} step, "
                                         f"fp32 torch CPU oracle, {cores} threads",
                               "ms_per_step": 1e3 * cdt}
        out["parity_rel_err_vs_cpu_oracle"] = err

    # ---- the other matmul mode, measured the same way right after (N=1 only; never part of `value`)
    if rank == 0 and world == 1 and not args.no_alt:
        other = "bf16x3" if args.matmul == "f32" else "f32"
        eng2 = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=tdim, target="video",
                               latent_shape=lat, prompt_tokens=na, alpha_bar=abar, guidance=args.guidance,
                               split_streams=bool(args.split_streams), matmul=other)
        eng2.set_prompt(z_a0)
        eng2.begin(sched)
        za2, zb2 = z0.clone(), torch.empty_like(z0)

        def run2(k):
            nonlocal za2, zb2
            for i in range(k):
                if i % S == 0:
                    eng2.rewind()
                    za2.copy_(z0)
                eng2.advance(za2, zb2)
                za2, zb2 = zb2, za2

        run2(args.warmup)
        torch.cuda.synchronize()
        a0 = time.perf_counter()
        run2(args.steps)
        torch.cuda.synchronize()
        adt = time.perf_counter() - a0
        alt = {"matmul": other, "value": args.steps / adt, "unit": "steps/s", "ms_per_step": 1e3 * adt / args.steps,
               "algorithmic_tflops": step_flops_per_sample(nv, na) * B * args.steps / adt / 1e12,
               "dtype": "f32" if other == "f32" else "f32 via 3xbf16 split operands (6-term products, f32 accumulate)"}
        if not args.no_roofline:
            L.prof_enable(True)
            run2(5)
            torch.cuda.synchronize()
            L.prof_enable(False)
            rep = L.prof_report()
            dom = max((k for k in rep if k.startswith("gemm")), key=lambda k: rep[k][1])
            n, ms, work = rep[dom]
            ach = work / (ms * 1e-3) / 1e12
            peak = PEAK_BF16X3_EQUIV_TFLOPS if dom.startswith("gemm_bf16x3") else PEAK_F32_MATRIX_TFLOPS
            alt["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                               "frac": ach / peak, "traffic": pmc_traffic(dom), "avg_launch_us": 1e3 * ms / max(n, 1)}
        if "cpu_baseline" in out:
            got2 = eng2.step(z0, tn.to(dev), tp.to(dev)).cpu()
            alt["parity_rel_err_vs_cpu_oracle"] = float((got2 - cpu_ref).abs().max() / max(1.0, float(cpu_ref.abs().max())))
        out["alt"] = alt

    if rank == 0:
        print(json.dumps(out))
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
