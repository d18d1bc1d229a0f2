"""debug: one-layer core forward, fused MLP vs two launches: where do they differ?"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import multimodal_diffusion_amd as A
from multimodal_diffusion_amd import _lib as L
from oracle import ref_cpu as R
dev = torch.device("cuda:0")
ws = R.synth_weights(seed=0, n_layers=1)
core = A.MMDiT(d_model=512, n_layers=1, n_heads=8).eval()
core.load_state_dict(ws["core"], strict=True)
core = core.to(dev); core.matmul = "bf16x3"
B, N = 16, 421
x = torch.randn(B, N, 512, generator=torch.Generator().manual_seed(3)).to(dev)
def tune(k, v): L.check(L.lib().avd_tune_set(k.encode(), v))
tune("s3_splitk", 0)
outs = {}
for fused in (0, 1, 0, 1):
    tune("mlp_fused", fused)
    outs.setdefault(fused, []).append(core(x).cpu())
print("unfused repeat equal", torch.equal(outs[0][0], outs[0][1]), "fused repeat equal", torch.equal(outs[1][0], outs[1][1]))
a, b = outs[0][0].view(-1, 512), outs[1][0].view(-1, 512)
d = (a - b).abs()
print("max abs diff", float(d.max()), "rel", float(d.max() / a.abs().max()), "nonzero frac", float((d > 0).float().mean()))
rows = (d > 0).any(1).nonzero().flatten()
cols = (d > 0).any(0).nonzero().flatten()
print("rows differing", rows.numel(), "of", a.shape[0], rows[:20].tolist(), "cols differing", cols.numel())
ref = R.mmdit_forward(x[:1].cpu(), ws["core"], 1, 8)
print("unfused vs oracle", float((outs[0][0][:1] - ref).abs().max()), "fused vs oracle", float((outs[1][0][:1] - ref).abs().max()))
