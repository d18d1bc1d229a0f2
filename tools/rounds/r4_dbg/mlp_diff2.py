"""debug: localise the last-bit difference between the fused MLP and the two launches: zero fc1.weight (H = GELU(b1): only the fc2 phase
sees data), then random fc1 with fc2 rows that pick single hidden columns (y shows H itself)."""
import sys, torch
sys.path.insert(0, ".")
import multimodal_diffusion_amd as A
from multimodal_diffusion_amd import _lib as L
from oracle import ref_cpu as R
dev = torch.device("cuda:0")
def tune(k, v): L.check(L.lib().avd_tune_set(k.encode(), v))
tune("s3_splitk", 0)
B, N = 16, 421
x = torch.randn(B, N, 512, generator=torch.Generator().manual_seed(3)).to(dev)
def run(ws, label):
    core = A.MMDiT(d_model=512, n_layers=1, n_heads=8).eval()
    core.load_state_dict(ws["core"], strict=True)
    core = core.to(dev); core.matmul = "bf16x3"
    outs = []
    for fused in (0, 1):
        tune("mlp_fused", fused)
        outs.append(core(x).cpu().view(-1, 512))
    d = (outs[0] - outs[1]).abs()
    print(f"{label}: max abs diff {float(d.max()):.3e}, nonzero frac {float((d > 0).float().mean()):.4f}, cols differing {(d > 0).any(0).sum().item()}")
ws = R.synth_weights(seed=0, n_layers=1)
run(ws, "baseline")
w2 = {k: {n: t.clone() for n, t in v.items()} for k, v in ws.items()}
w2["core"]["blocks.0.mlp.fc1.weight"].zero_()
run(w2, "fc1.weight = 0 (fc2 phase only sees GELU(b1))")
w3 = {k: {n: t.clone() for n, t in v.items()} for k, v in ws.items()}
sel = torch.zeros(512, 2048); sel[torch.arange(512), torch.arange(512) * 4] = 1.0      # y[:, n] = x + H[:, 4 n] + b2
w3["core"]["blocks.0.mlp.fc2.weight"].copy_(sel)
run(w3, "fc2 = column selector (y shows H)")
w4 = {k: {n: t.clone() for n, t in v.items()} for k, v in ws.items()}
w4["core"]["blocks.0.mlp.fc2.weight"].copy_(sel); w4["core"]["blocks.0.mlp.fc1.bias"].fill_(-30.0)          # GELU(-30 + small) = 0: H = 0
run(w4, "fc2 selector, fc1 bias -30 (H ~ 0)")
w5 = {k: {n: t.clone() for n, t in v.items()} for k, v in ws.items()}
w5["core"]["blocks.0.mlp.fc1.weight"].zero_(); w5["core"]["blocks.0.mlp.fc2.weight"].copy_(sel)
run(w5, "fc1.weight = 0, fc2 selector (GELU(b1) passes through exactly)")
w6 = {k: {n: t.clone() for n, t in v.items()} for k, v in ws.items()}
w6["core"]["blocks.0.mlp.fc1.weight"].zero_(); w6["core"]["blocks.0.mlp.fc1.bias"].fill_(30.0)
run(w6, "fc1.weight = 0, b1 = 30 (H = 30 exactly in one plane; fc2 accumulates 30 W2)")
w7 = {k: {n: t.clone() for n, t in v.items()} for k, v in ws.items()}
w7["core"]["blocks.0.mlp.fc1.weight"].zero_(); w7["core"]["blocks.0.mlp.fc1.bias"].fill_(30.0)
w7["core"]["blocks.0.mlp.fc2.weight"].copy_(w7["core"]["blocks.0.mlp.fc2.weight"].bfloat16().float())
run(w7, "... and W2 rounded to bf16 (one plane each side: only the h.h term is non-zero)")
