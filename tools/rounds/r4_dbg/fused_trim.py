"""debug: which of (fused, trim, s3_rt4, s3_deep4) breaks the step at 256x256 B=8"""
import sys, torch
sys.path.insert(0, ".")
import multimodal_diffusion_amd as A
from multimodal_diffusion_amd import _lib as L
from oracle import ref_cpu as R
dev = torch.device("cuda:0")
def tune(k, v): L.check(L.lib().avd_tune_set(k.encode(), v))
B = 8
ws = R.synth_weights(seed=0)
core = A.MMDiT(d_model=512, n_layers=8, n_heads=8, mlp_ratio=4.0).eval(); core.load_state_dict(ws["core"], strict=True)
head = A.MultiModalNoiseHead({"video": 512, "audio": 512}, {"video": 256, "audio": 32}, hidden_dim=512).eval(); head.load_state_dict(ws["head"], strict=True)
av, aa = A.LinearAdapter(256, 256), A.LinearAdapter(32, 256)
av.load_state_dict(ws["adapt_v"], strict=True); aa.load_state_dict(ws["adapt_a"], strict=True)
core, head, av, aa = core.to(dev), head.to(dev), av.to(dev), aa.to(dev)
g = torch.Generator().manual_seed(900 + B)
z_v = torch.randn(B, 8, 12, 32, 32, generator=g); z_a = torch.randn(B, 8, 150, generator=g)
abar = R.alpha_bar_table(R.beta_table(1000))
tn = torch.tensor(([982, 500, 16, 999] * B)[:B]); tp = torch.tensor(([966, 480, -1, 979] * B)[:B])
ref = R.denoise_step_a2v(z_v[:1], z_a[:1], tn[:1], tp[:1], abar, adapt_v=ws["adapt_v"], adapt_a=ws["adapt_a"], core=ws["core"], head=ws["head"], n_layers=8, n_heads=8, guidance=3.5)
def rel(a, b): return float((a - b).norm() / b.norm())
tune("s3_splitk", 0)
base = None
for fused in (0, 1):
    for trim in (1, 0):
        for rt4, deep in ((8, 0), (0, 0), (0, 1), (5, 1), (8, 1)):
            tune("mlp_fused", fused); tune("core_trim", trim); tune("s3_rt4", rt4); tune("s3_deep4", deep)
            eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z_v.shape),
                                  prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul="bf16x3")
            eng.set_prompt(z_a.to(dev))
            outs = [eng.step(z_v.to(dev), tn.to(dev), tp.to(dev)).cpu() for _ in range(3)]
            if base is None: base = outs[0]
            print(f"fused {fused} trim {trim} rt4 {rt4} deep {deep}: vs oracle {rel(outs[0][:1], ref):.2e}; repeats equal {torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])}; "
                  f"equal to first config {torch.equal(outs[0], base)}; max diff {float((outs[0] - base).abs().max()):.2e}", flush=True)
