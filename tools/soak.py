import sys, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import multimodal_diffusion_amd as A
from multimodal_diffusion_amd import schedule_utils as su
import bench
dev = torch.device('cuda:0')
mods, tdim = bench.build_modules(dev)
av, aa, core, head = mods
abar = su.alphas_cumprod_from_betas(su.make_beta_schedule(1000, "cosine", 1e-4, 0.02))[1]
sched = su.make_sampling_schedule(1000, 50)
B = 32
z0 = torch.randn(B, 8, 12, 32, 32, generator=torch.Generator().manual_seed(1)).to(dev)
za = torch.randn(B, 8, 150, generator=torch.Generator().manual_seed(2)).to(dev)
for mode in ("f32", "bf16x3", "f16x2"):
    eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=tdim, target="video", latent_shape=tuple(z0.shape),
                          prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode)
    eng.set_prompt(za)
    ref = eng.run(z0, sched)
    ok = bool(torch.isfinite(ref).all())
    same = True
    for rep in range(8):
        out = eng.run(z0, sched, graph=(rep % 2 == 1))
        same = same and torch.equal(out, ref)
    print(mode, "finite", ok, "8 repeated 50-step trajectories bit-identical (eager and graph):", same, "max|z|", float(ref.abs().max()), flush=True)
