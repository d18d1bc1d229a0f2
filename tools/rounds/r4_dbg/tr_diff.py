"""debug: does the transposed-accumulator (image) epilogue kernel give the same fp32 sums as the plain (fp32 out) kernel?"""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from multimodal_diffusion_amd import functional as Fn, _lib as L
from test_gpu_parity import _split3_decode
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (M, N, K) in ((6736, 512, 2048), (6736, 2048, 512), (26944, 512, 2048)):
    x = torch.randn(M, K, generator=g).to(dev); w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev); b = torch.randn(N, generator=g).to(dev)
    x3, w3 = Fn.split3(x), Fn.split3(w)
    y = Fn.linear_bf16x3(x3, M, w3, N, K, bias=b)                       # EPI_BIAS, plain accumulators
    img = Fn.linear_bf16x3(x3, M, w3, N, K, bias=b, out_split3=True)    # EPI_SPLIT, transposed accumulators
    pl = _split3_decode(img.cpu().numpy(), M, N).astype(np.float64).sum(0)
    d = np.abs(pl - y.cpu().numpy().astype(np.float64))
    print((M, N, K), "plain vs transposed: max abs diff", d.max(), "nonzero frac", (d > 0).mean())
