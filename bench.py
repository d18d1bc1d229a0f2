#!/usr/bin/env python3
"""Headline benchmark: denoising steps/sec @256x256 multimodal-cond, batch=32 per GPU, on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one CFG denoising step for a batch of 32 samples (reference loop body
avdiff/models/infer/sample_clip.py:359-389: tokenise -> adapters -> t-emb -> MMDiT x2 (cond+null, stacked to 2B) ->
noise head -> CFG -> un-patch -> DDIM), workload C3 of BASELINE.json (256x256 -> 384 video + 37 audio tokens,
mvp.yaml model dims d=512 L=8 H=8), fp32 results, synthetic inputs, random-init weights, inputs resident in HBM.
Weak scaling: every rank steps its own 32 samples; the only collective is one RCCL broadcast of the
conditioning latents before the loop.  Rank 0 prints ONE JSON line.

Matrix-pipe modes (--matmul), all with fp32 accumulation and fp32 results:
  "bf16x3" (default, the headline `value`) every fp32 operand split EXACTLY into three bf16 planes (24 significant bits: the
           operands ARE the fp32 numbers), six product terms, the dropped ones <= 2^-24 relative — the reference's fp32 arithmetic on
           the bf16 matrix pipe (tests/test_gpu_parity.py bf16x3 suites);
  "f32"    fp32 MFMA everywhere (the round-1 default);  "bf16x3_strict" nine terms.   Both measured in the same run under "alt".
  "f16x2"  every operand held as two scaled fp16 planes — 22-bit operands, NARROWER than the reference's fp32, so never the headline:
           reported in the same line as "speed_mode" with its measured error (<= the fp32-MFMA path's on every test of
           tests/test_gpu_f16x2.py, scales bounded from the weights so fp16 cannot overflow);
  "bf16"   one bf16 plane: reduced precision, BASELINE config C2 — its error is reported, not gated.
Every fp32-level mode is GATED: one step against the CPU oracle must agree to 1e-4 or the run exits non-zero.
The f16x2 engine runs the cond / null CFG halves as two kernel chains on two HIP streams by default (bit-identical results,
--split-streams 0 turns it off); every per-kernel roofline pass runs single-stream, where a kernel's duration is its own.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch

PEAK_F32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
PEAK_HBM_GBS = 8000.0
# split-operand matmul: one product costs TERMS bf16 MFMAs, so the dense bf16 peak (2,516.6 TFLOP/s) prices ALGORITHMIC fp32
# flops at 2516.6 / TERMS (the chip holds well under 2.4 GHz on this load; that is not priced in)
PEAK_BF16_MATRIX_TFLOPS = 2516.6
# per-clock rates behind those two figures (MI355X_MICROARCH.md, Matrix cores): FLOP per clock per SIMD, four SIMDs per CU
F32_MATRIX_FLOP_PER_CLK_SIMD, BF16_MATRIX_FLOP_PER_CLK_SIMD = 64, 1024
PEAK_SOURCE = "MI355X_MICROARCH.md constants (256 CUs x 4 SIMDs x FLOP/clk/SIMD x 2.4 GHz)"


def peaks_from_device(device_index: int) -> None:
    """Re-derive the matrix peaks from what THIS box reports (peaks from the box, not hard-coded): CU count and peak
    shader clock from the HIP runtime (hipDeviceGetAttribute: MultiprocessorCount, ClockRate — the values rocminfo prints),
    per-clock rates from the microarchitecture guide.  On an MI355X this reproduces 157.3 / 2,516.6 TFLOP/s; the constants above
    stay in force when the query fails or returns something implausible.  (sysfs pp_dpm_sclk is NOT a source for this: on this
    ASIC its level 1 is the CURRENT clock.)"""
    global PEAK_F32_MATRIX_TFLOPS, PEAK_BF16_MATRIX_TFLOPS, PEAK_SOURCE
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")           # the runtime torch has already loaded
        HIP_ATTR_CLOCK_RATE_KHZ, HIP_ATTR_MULTIPROCESSOR_COUNT = 5, 63      # hipDeviceAttribute_t, hip_runtime_api.h of ROCm 7.x
        khz, cus = ctypes.c_int(0), ctypes.c_int(0)
        if hip.hipDeviceGetAttribute(ctypes.byref(khz), HIP_ATTR_CLOCK_RATE_KHZ, device_index) != 0:
            return
        if hip.hipDeviceGetAttribute(ctypes.byref(cus), HIP_ATTR_MULTIPROCESSOR_COUNT, device_index) != 0:
            return
        if cus.value != torch.cuda.get_device_properties(device_index).multi_processor_count or not 5e5 < khz.value < 5e6:
            return                                    # enum drift or an implausible clock: keep the guide's constants
        per_clk = cus.value * 4 * khz.value * 1e3 / 1e12
        PEAK_F32_MATRIX_TFLOPS = per_clk * F32_MATRIX_FLOP_PER_CLK_SIMD
        PEAK_BF16_MATRIX_TFLOPS = per_clk * BF16_MATRIX_FLOP_PER_CLK_SIMD
        PEAK_SOURCE = (f"this box: {cus.value} CUs x 4 SIMDs x {khz.value / 1e3:.0f} MHz peak shader clock (hipDeviceGetAttribute) x "
                       f"{F32_MATRIX_FLOP_PER_CLK_SIMD} / {BF16_MATRIX_FLOP_PER_CLK_SIMD} FLOP/clk/SIMD (MI355X_MICROARCH.md); HBM {PEAK_HBM_GBS:.0f} GB/s is the spec figure")
    except Exception:      # no runtime handle, no symbol: keep the guide's constants
        return
MODE_TERMS = {"f32": 0, "bf16x3": 6, "bf16x3_strict": 9, "bf16": 1, "f16x2": 3}
MODE_DTYPE = {
    "f32": "f32",
    "bf16x3": "f32 via 3xbf16 split operands (6-term products, f32 accumulate; fp32-level error)",
    "bf16x3_strict": "f32 via 3xbf16 split operands (all 9 product terms, f32 accumulate)",
    "bf16": "bf16 operands, f32 accumulate (reduced precision; error reported, not a parity path)",
    "f16x2": "f32 via 2xfp16 split operands (22-bit operands, 3-term products, f32 accumulate)",
}
PARITY_TOL = 1e-4                      # one step vs the CPU oracle, max|d| / max(1, max|ref|)  
GATED_MODES = ("f32", "bf16x3", "bf16x3_strict", "f16x2")      # fp32-level claims: a violation fails the run


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


def csrc_hash() -> str:
    """Identity of the kernel sources a profile was taken at (sha256 over csrc/*.hip, *.h and the C header, first 16 hex)."""
    h = hashlib.sha256()
    files = sorted((ROOT / "multimodal_diffusion_amd" / "csrc").glob("*.hip")) + \
        sorted((ROOT / "multimodal_diffusion_amd" / "csrc").glob("*.h")) + [ROOT / "include" / "avdiff_hip.h"]
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_tag: str, matmul: str):
    """(bytes beyond L2 per launch, source) of `kernel_tag` from the newest committed PMC pass (tools/pmc_traffic.py) taken at
    THESE kernel sources, else (None, reason).  The PMC passes cannot run inside this process (rocprofv3 wraps the command),
    so the figure is read back — and dropped as soon as csrc/ differs from what was profiled."""
    want = csrc_hash()
    pat = "r*_traffic.json" if matmul == "f32" else f"r*_traffic_{matmul}.json"
    for f in sorted((ROOT / "profiles").glob(pat), reverse=True):      # newest round first
        try:
            doc = json.loads(f.read_text())
            ks = doc["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        if doc.get("csrc_sha16") != want:
            return None, f"stale: {f.name} was taken at csrc {doc.get('csrc_sha16', 'unknown')}, tree is {want}"
        for name, v in ks.items():
            if kernel_tag.replace(" ", "") in name.replace(" ", ""):
                return v["traffic_bytes_per_launch"], f"profiles/{f.name}@csrc:{want}"
        return None, f"kernel not in profiles/{f.name}"
    return None, "no PMC profile committed for this mode"


class PowerClockSampler:
    """Core clock and socket power of THIS GPU while a pass runs, read from sysfs by a sampling thread (no child process: a process
    that has initialised HIP must not fork + exec on the GPU boxes).  The builder's explanation of every roofline fraction in this
    file is "the chip holds its clock down under the power cap" (DESIGN.md 4.5-4.8): the driver's record carries the evidence."""

    def __init__(self, device_index: int):
        self.dir = self._find(device_index)
        self.samples = []          # (t, sclk_mhz | None, power_w | None)
        self._stop = False
        self._thread = None

    @staticmethod
    def _find(device_index: int):
        try:
            pr = torch.cuda.get_device_properties(device_index)
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            d = Path("/sys/bus/pci/devices") / bdf
            if (d / "pp_dpm_sclk").exists():
                return d
        except Exception:
            pass
        cards = sorted(Path("/sys/class/drm").glob("card[0-9]*/device/pp_dpm_sclk"))
        return cards[0].parent if len(cards) == 1 else None      # several cards and no PCI match: do not guess

    def _read(self):
        mhz = watts = None
        try:
            for line in (self.dir / "pp_dpm_sclk").read_text().splitlines():
                if line.rstrip().endswith("*"):
                    mhz = float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
        except (OSError, ValueError, IndexError):
            pass
        for name in ("power1_average", "power1_input"):
            for f in self.dir.glob(f"hwmon/hwmon*/{name}"):
                try:
                    watts = float(f.read_text()) / 1e6
                    break
                except (OSError, ValueError):
                    continue
            if watts is not None:
                break
        return mhz, watts

    def cap_watts(self):
        if self.dir is None:
            return None
        for f in self.dir.glob("hwmon/hwmon*/power1_cap"):
            try:
                return float(f.read_text()) / 1e6
            except (OSError, ValueError):
                pass
        return None

    def start(self):
        import threading
        if self.dir is None:
            return
        self._t0 = time.perf_counter()

        def loop():
            while not self._stop:
                mhz, watts = self._read()
                self.samples.append((time.perf_counter() - self._t0, mhz, watts))
                time.sleep(0.05)
        self._thread = threading.Thread(target=loop, daemon=True)
        self._thread.start()

    def stop(self, settle_s: float):
        """median clock / power over the samples taken after `settle_s` seconds of load"""
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=2.0)
        if self.dir is None:
            return {"source": None, "reason": "no sysfs node found for this device (pp_dpm_sclk / hwmon power1_average)"}
        late = [x for x in self.samples if x[0] >= settle_s] or self.samples
        med = lambda v: (sorted(v)[len(v) // 2] if v else None)
        return {"source": f"sysfs {self.dir}", "samples": len(late), "settle_s": settle_s,
                "sclk_mhz_median": med([x[1] for x in late if x[1] is not None]),
                "socket_power_w_median": med([x[2] for x in late if x[2] is not None]),
                "socket_power_cap_w": self.cap_watts(),
                "note": "pp_dpm_sclk reads up to ~10 % above the in-kernel clock of an MFMA-dense loop (MI355X_MICROARCH.md, DVFS give-back 6)"}


def workload_name(size: int, B: int) -> str:
    """Which BASELINE.json configuration (SURVEY 8d's mapping) a geometry is: the default line is C3; the other geometries are
    parity / record runs and must not carry C3's name."""
    if size == 256:
        return "C3" if B == 32 else f"C3 geometry at batch {B}"
    if size == 32:
        return "C1" if B == 4 else f"C1 geometry at batch {B}"
    if size == 64:
        return "C2" if B == 32 else f"C2 geometry at batch {B}"
    if size == 512:
        return "C5 per-GPU shard" if B == 8 else f"C5 geometry at batch {B}"
    if size == 128:
        return f"shipped mvp.yaml geometry (not a BASELINE config) at batch {B}"
    return f"{size}x{size} at batch {B} (not a BASELINE config)"


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=256, help="video size (square); 256 = BASELINE C3")
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--sampler-steps", type=int, default=50)
    ap.add_argument("--guidance", type=float, default=3.5)
    ap.add_argument("--graph", action="store_true", help="replay a captured 2-step HIP graph instead of eager launches")
    ap.add_argument("--split-streams", type=int, default=-1,
                    help="run the cond/null halves on two HIP streams: 1 / 0; -1 (default) = the engine's rule (on for f16x2 at >= 6144 rows per half)")
    ap.add_argument("--matmul", default="bf16x3", choices=sorted(MODE_TERMS),
                    help="bf16x3 (default): three bf16 planes per operand (exact fp32 operand split), six product terms; f32: fp32 MFMA "
                         "everywhere; bf16x3_strict: all nine product terms; f16x2: two scaled fp16 planes (22-bit operands; reported as "
                         "speed_mode, never the headline); bf16: plain bf16 operands (reduced precision, reported error)")
    ap.add_argument("--attn", default="default", choices=["default", "fp8"],
                    help="fp8: e4m3 QK^T / PV in the attention (reduced precision, BASELINE C5; needs a split matmul mode — f16x2, bf16x3 or bf16; never the default)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra (untimed-region) measurements of the other matmul modes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse N > 1 on a 1-GPU box)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (with --backend gloo)")
    ap.add_argument("--verify-ranks", action="store_true",
                    help="N > 1: also report, per rank, a checksum of the broadcast conditioning and of the final latents "
                         "(two tiny all-gathers outside the timed region; tests/test_gpu_parity.py::test_bench_two_ranks_share_device)")
    return ap.parse_args(argv)


def setup_run(args, need_gpu: bool = True):
    """Everything bench.py does BEFORE its first GPU call, as one function: rank / world / device from torchrun's env,
    process-group init, geometry, the ONE broadcast of the conditioning latents and this rank's shard, the per-rank initial
    latent and the schedule tables.  `need_gpu=False` lets the CPU test suite run exactly this path under gloo."""
    from multimodal_diffusion_amd import dist as D, schedule_utils as su
    multi = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if need_gpu and multi and args.backend == "nccl" and not args.share_device:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local = D.init_from_env(args.backend if multi else None)
    if args.share_device:
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if need_gpu:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
    else:
        dev = torch.device("cpu")
    comm_dev = dev if (need_gpu and args.backend == "nccl") else torch.device("cpu")
    B, size, S = args.batch, args.size, args.sampler_steps
    lat = (B, 8, 12, size // 8, size // 8)
    nv, na = 6 * (size // 32) ** 2, 37
    abar = su.alphas_cumprod_from_betas(su.make_beta_schedule(1000, "cosine", 1e-4, 0.02))[1]
    sched = su.make_sampling_schedule(1000, S)
    # conditioning: root draws the global batch of prompt latents, ONE broadcast, each rank keeps its shard
    gshape = (B * world, 8, 150)
    cond = torch.randn(gshape, generator=torch.Generator().manual_seed(2)) if rank == 0 else None
    cond_all = D.broadcast_conditioning(cond, gshape, comm_dev).to(dev)
    z_a0 = D.local_conditioning(cond_all, rank, world)
    z0 = torch.randn(lat, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    return dict(rank=rank, world=world, local=local, dev=dev, comm_dev=comm_dev, B=B, size=size, S=S, lat=lat, nv=nv, na=na,
                abar=abar, sched=sched, z_a0=z_a0, z0=z0, global_batch=B * world,
                cond_checksum=float(cond_all.double().sum().item()))


def roofline_of(rep, matmul: str, n_steps: int):
    """dominant-kernel roofline object + per-kernel table from the library's per-launch HIP-event records"""
    dom = max((k for k in rep if k.startswith("gemm")), key=lambda k: rep[k][1])   # most time => dominant
    n, ms, work = rep[dom]
    achieved = work / (ms * 1e-3) / 1e12

    def peak_of(name):
        if name.startswith(("gemm_bf16x3_m16", "gemm_bf16x3_w128")):      # <EPI, WAVES, RT> / <EPI, RT>: always the six bf16x3 terms (two per 16x16x32 MFMA)
            return PEAK_BF16_MATRIX_TFLOPS / 6
        if name.startswith("gemm_bf16x3"):            # <EPI, TERMS, ...>
            return PEAK_BF16_MATRIX_TFLOPS / int(name.split("<")[1].split(",")[1].strip(" >"))
        if name.startswith("attn_bf16x3"):            # <TERMS>
            return PEAK_BF16_MATRIX_TFLOPS / int(name.split("<")[1].strip(" >"))
        return PEAK_F32_MATRIX_TFLOPS

    traffic, source = pmc_traffic(dom, matmul)
    roof = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": peak_of(dom), "unit": "TFLOP/s",
            "frac": achieved / peak_of(dom), "peak_source": PEAK_SOURCE, "traffic": traffic, "traffic_source": source,
            "launches": n, "avg_launch_us": 1e3 * ms / max(n, 1), "flops_per_launch": work / max(n, 1)}
    tot_ms = sum(v[1] for v in rep.values())
    kern = {}
    for k, (cnt, kms, w) in rep.items():
        if cnt == 0:
            continue
        e = {"launches_per_step": cnt / n_steps, "ms_per_step": kms / n_steps, "share": kms / tot_ms}
        if k.startswith(("gemm", "attn")):
            e["tflops"] = w / (kms * 1e-3) / 1e12
            e["frac_of_mfma_peak_for_its_dtype"] = e["tflops"] / peak_of(k)
        else:
            e["gbs"] = w / (kms * 1e-3) / 1e9
            e["frac_of_hbm_peak"] = e["gbs"] / PEAK_HBM_GBS
        kern[k] = e
    return roof, kern


def main():
    args = parse_args()
    from multimodal_diffusion_amd import dist as D, _lib as L
    import multimodal_diffusion_amd as A

    ctx = setup_run(args)
    rank, world, dev = ctx["rank"], ctx["world"], ctx["dev"]
    B, size, S, lat, nv, na = ctx["B"], ctx["size"], ctx["S"], ctx["lat"], ctx["nv"], ctx["na"]
    abar, sched, z_a0, z0 = ctx["abar"], ctx["sched"], ctx["z_a0"], ctx["z0"]
    mods, tdim = build_modules(dev)
    av, aa, core, head = mods
    peaks_from_device(dev.index if dev.index is not None else 0)

    def make_engine(mode, split="arg"):
        if split == "arg":
            split = None if args.split_streams < 0 else bool(args.split_streams)
        e = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=tdim, target="video",
                            latent_shape=lat, prompt_tokens=na, alpha_bar=abar, guidance=args.guidance,
                            split_streams=split, matmul=mode,
                            attn=args.attn if mode != "f32" else "default")
        e.set_prompt(z_a0)
        e.begin(sched)
        return e

    eng = make_engine(args.matmul)
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
    dt = D.max_over_ranks(time.perf_counter() - t0, ctx["comm_dev"])
    assert torch.isfinite(za).all(), "non-finite latent after the timed region"
    verify = None
    if args.verify_ranks:
        verify = {"conditioning_checksum_per_rank": D.gather_scalars(ctx["cond_checksum"], ctx["comm_dev"]),
                  "latent_abs_sum_per_rank": D.gather_scalars(float(za.double().abs().sum().item()), ctx["comm_dev"])}

    out = None
    if rank == 0:
        fl = step_flops_per_sample(nv, na) * B
        per_gpu = args.steps / dt
        out = {
            "metric": "denoising steps/sec @256x256 multimodal-cond batch=32",
            # whole-job aggregate under weak scaling: every rank steps its own batch of 32, so N ranks complete N batch-32 steps
            # per step time.  The global-batch (32 N samples) step rate is global_batch_steps_per_s.
            "value": world * per_gpu,
            "unit": "steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": MODE_DTYPE[args.matmul] + ("; attention in fp8 e4m3 (reduced precision)" if args.attn == "fp8" else ""),
            "data": "synthetic",
            "config": {"workload": f"{workload_name(size, B)}: {size}x{size} audio->video CFG denoising step (cond+null MMDiT d512 L8 H8 + noise head + "
                                   f"DDIM), {nv}+{na} tokens, batch {B} per GPU, DDIM {S} steps, guidance {args.guidance}",
                       "global_batch": B * world, "tokens": nv + na, "sampler_steps": S,
                       "parallelism": f"dp{world}", "launch": "hipgraph" if args.graph else "eager", "matmul": args.matmul, "attn": args.attn,
                       "cfg_halves": "two HIP streams" if eng._split_streams else "one stacked 2B batch, one stream"},
            "per_gpu_steps_per_s": per_gpu,
            "global_batch_steps_per_s": per_gpu,      # steps/s of the GLOBAL batch (32 x n_gpus samples advance together): SURVEY 8e's rate
            "value_is": "sum over ranks of batch-32 steps/s (weak scaling: n_gpus x global_batch_steps_per_s)",
            "sample_steps_per_s": world * B * per_gpu,
            "algorithmic_tflops_per_gpu": fl * per_gpu / 1e12,
            "algorithmic_tflops_total": fl * world * per_gpu / 1e12,
        }
        if verify is not None:
            out["verify"] = verify

    def instrumented(mode, base_engine):
        """(roofline, kernels, single_stream_steps_per_s | None) of `mode`: HIP events around every launch, on its stream, over 5
        single-stream steps — with the CFG halves on two streams an event pair around one launch would time two kernels sharing
        the chip, so a two-stream engine gets a one-stream twin for this pass (same kernels, stacked 2B batch)."""
        eng_r = make_engine(mode, split=False) if base_engine._split_streams else base_engine
        zr, zr2 = z0.clone(), torch.empty_like(z0)

        def run_r(k):
            nonlocal zr, zr2
            for i in range(k):
                if i % S == 0:
                    eng_r.rewind()
                    zr.copy_(z0)
                eng_r.advance(zr, zr2)
                zr, zr2 = zr2, zr

        run_r(2)
        torch.cuda.synchronize()
        L.prof_enable(True)
        run_r(5)
        torch.cuda.synchronize()
        L.prof_enable(False)
        roof, kern = roofline_of(L.prof_report(), mode, 5)
        roof["pass"] = "5 instrumented single-stream steps after the timed region (HIP events around every launch, on its stream)"
        single = None
        if eng_r is not base_engine:
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            run_r(args.steps)
            torch.cuda.synchronize()
            single = args.steps / (time.perf_counter() - r0)
        return roof, kern, single

    # ---- roofline of the dominant kernel: per-launch HIP events on the launch stream, separate instrumented pass
    if rank == 0 and not args.no_roofline:
        out["roofline"], out["kernels"], single = instrumented(args.matmul, eng)
        if single is not None:
            out["single_stream_steps_per_s"] = single

    # ---- clock and socket power the chip holds on this workload: >= 2.5 s of back-to-back steps of the timed engine, sysfs sampled
    # every 50 ms, medians over what was read after the first second (outside the timed region)
    if rank == 0 and not args.no_roofline:
        smp = PowerClockSampler(dev.index if dev.index is not None else 0)
        smp.start()
        p0 = time.perf_counter()
        n_pc = 0
        while time.perf_counter() - p0 < 2.5:
            run_steps(10)
            torch.cuda.synchronize()
            n_pc += 10
        out["power_clock"] = smp.stop(settle_s=1.0)
        out["power_clock"]["steps_per_s_during_pass"] = n_pc / (time.perf_counter() - p0)

    # ---- CPU side (rank 0, N = 1): the oracle's step is the parity reference; the TIMED baseline is the port of the same step on
    # the fused ATen kernels the reference's modules dispatch to (oracle/ref_cpu_fast.py; tools/cpu_port_speed.py holds it to the
    # reference's own MMDiT.forward time in the build container: profiles/r03_cpu_port_speed.json)
    tn = tp = cpu_ref = None
    gate = []          # (mode, err) of every fp32-level mode checked against the oracle
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_cpu as R, ref_cpu_fast as RF
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
            cpu_ref = R.denoise_step_a2v(zc, zac, tn, tp, abar, **kw)            # the adjudicator (pinned by tests/golden)
            port = RF.denoise_step_a2v(zc, zac, tn, tp, abar, **kw)              # warm-up of the timed port
            c0 = time.perf_counter()
            for _ in range(args.cpu_steps):
                RF.denoise_step_a2v(zc, zac, tn, tp, abar, **kw)
            cdt = (time.perf_counter() - c0) / args.cpu_steps
        denom = max(1.0, float(cpu_ref.abs().max()))
        got = eng.step(z0, tn.to(dev), tp.to(dev)).cpu()
        err = float((got - cpu_ref).abs().max() / denom)
        out["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": f"{args.cpu_steps} timed steps (+1 warm-up) of the same batch-{B} {size}x{size} step: fp32 torch CPU "
                                         f"port on the reference's fused ATen kernels (F.linear / scaled_dot_product_attention / F.gelu), "
                                         f"{cores} threads",
                               "ms_per_step": 1e3 * cdt,
                               "port_rel_err_vs_oracle": float((port - cpu_ref).abs().max() / denom)}
        out["parity_rel_err_vs_cpu_oracle"] = err
        out["parity_tolerance"] = PARITY_TOL
        gate.append((args.matmul, err))

    # ---- the other matmul modes, measured the same way right after (N=1 only; never part of `value`)
    if rank == 0 and world == 1 and not args.no_alt:
        alts = []
        for other in ("f32", "bf16x3", "bf16x3_strict", "f16x2"):      # each with its engine's default stream layout
            if other == args.matmul:
                continue
            eng2 = make_engine(other, split=None)
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
                   "algorithmic_tflops": step_flops_per_sample(nv, na) * B * args.steps / adt / 1e12, "dtype": MODE_DTYPE[other],
                   "cfg_halves": "two HIP streams" if eng2._split_streams else "one stacked 2B batch, one stream"}
            if not args.no_roofline and other != "bf16x3_strict":
                alt["roofline"], _, single = instrumented(other, eng2)
                if single is not None:
                    alt["single_stream_steps_per_s"] = single
            if cpu_ref is not None:
                got2 = eng2.step(z0, tn.to(dev), tp.to(dev)).cpu()
                alt["parity_rel_err_vs_cpu_oracle"] = float((got2 - cpu_ref).abs().max() / max(1.0, float(cpu_ref.abs().max())))
                gate.append((other, alt["parity_rel_err_vs_cpu_oracle"]))
            if other == "f16x2":
                alt["operands"] = "22-bit (two fp16 planes): narrower than the reference's fp32, reported beside the headline, never as it"
                out["speed_mode"] = alt
            else:
                alts.append(alt)
            del eng2
        out["alt"] = alts

    if rank == 0:
        print(json.dumps(out), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    # --attn fp8 is an explicit reduced-precision request (BASELINE C5): its error is reported in the line, never gated as fp32-level
    bad = [(m, e) for m, e in gate if m in GATED_MODES and args.attn == "default" and not e < PARITY_TOL]
    if bad:
        raise SystemExit("parity gate failed (one step vs the CPU oracle, tolerance %g): %s" % (PARITY_TOL, bad))


if __name__ == "__main__":
    main()
