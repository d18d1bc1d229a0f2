import sys, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")
import multimodal_diffusion_amd as A
from multimodal_diffusion_amd import functional as Fn, _lib as L, schedule_utils as su
from oracle import ref_cpu as R
dev = torch.device("cuda:0")
ws = R.synth_weights(seed=0)
def mods():
    core = A.MMDiT(d_model=512, n_layers=8, n_heads=8, mlp_ratio=4.0).eval(); core.load_state_dict(ws["core"])
    head = A.MultiModalNoiseHead({"video": 512, "audio": 512}, {"video": 256, "audio": 32}, hidden_dim=512).eval(); head.load_state_dict(ws["head"])
    av, aa = A.LinearAdapter(256, 256), A.LinearAdapter(32, 256)
    av.load_state_dict(ws["adapt_v"]); aa.load_state_dict(ws["adapt_a"])
    return [m.to(dev) for m in (core, head, av, aa)]
B = 16
g = torch.Generator().manual_seed(77)
z = torch.randn(B, 8, 12, 32, 32, generator=g).to(dev)
za = torch.randn(B, 8, 150, generator=g).to(dev)
abar = R.alpha_bar_table(R.beta_table(1000))
tn = torch.full((B,), 999, device=dev); tp = torch.full((B,), 749, device=dev)
sched = su.make_sampling_schedule(1000, 4)
import os
for rep in range(6):
  for mode in ("bf16x3", "f16x2"):
    outs = {}
    junk = torch.randn(64, 1024, 1024, device=dev) * float("nan") if rep % 2 else None      # poison freed memory on odd reps
    del junk
    for split in (False, True):
        core, head, av, aa = mods()
        eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z.shape),
                              prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode, split_streams=split)
        eng.set_prompt(za)
        outs[split] = (eng.run(z, sched, graph=False), eng.run(z, sched, graph=True))
    torch.cuda.synchronize()
    names = {(False, 0): "single/eager", (False, 1): "single/graph", (True, 0): "two/eager", (True, 1): "two/graph"}
    ref = outs[False][0]
    msg = []
    for k, nm in names.items():
        d = (outs[k[0]][k[1]] - ref).abs()
        if not torch.equal(outs[k[0]][k[1]], ref):
            bad = ((d > 0) | torch.isnan(d)).nonzero()
            msg.append(f"{nm}: max {float(d.max()):.3e} n {len(bad)} samples {bad[:,0].unique().tolist()}")
    print(rep, mode, "finite", bool(torch.isfinite(ref).all()), "OK" if not msg else msg)
