"""GPU tests of the "f16x2" matmul mode (two scaled fp16 planes per operand, three product terms — include/avdiff_hip.h,
csrc/gemm_bf16x3.hip, csrc/attn_bf16x3.hip): accuracy against fp64 / the CPU oracle next to the fp32-MFMA path, the image
format, the scale bounds under adversarial weights, and the behaviour at the edges of the fp16 range."""
import math

import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import ref_cpu as R
from test_gpu_parity import TOL, _full_modules, dev, full  # noqa: F401  (dev / full are fixtures)

pytestmark = pytest.mark.gpu


def _h2_decode(img: np.ndarray, rows: int, K: int) -> np.ndarray:
    """Independent reading of the f16x2 image (split3 geometry, planes 0 and 1 hold fp16): -> [2, rows, K] float64."""
    r = np.arange(rows)[:, None]
    k = np.arange(K)[None, :]
    f = ((r & 127) >> 4) & 1
    half = (k >> 3) & 1
    base = ((r >> 7) * (K // 16) + (k >> 4)) * (128 * 96) + (r & 127) * 32 + ((half ^ f) * 16) + (k & 7) * 2
    f16 = img.view(np.float16)
    return np.stack([f16[(base + p * 4096) // 2].astype(np.float64) for p in range(2)])


def _gemm(dev, x, w, b=None, r=None, sx=None, sw=None):
    from multimodal_diffusion_amd import functional as Fn
    M, K = x.shape
    N = w.shape[0]
    x2, sx = Fn.split_f16x2(x.to(dev), sx)
    w2, sw = Fn.split_f16x2(w.to(dev), sw)
    y2 = Fn.linear_f16x2(x2, M, w2, N, K, sx * sw, bias=None if b is None else b.to(dev),
                         residual=None if r is None else r.to(dev)).cpu().double()
    y32 = Fn.linear(x.to(dev), w.to(dev), None if b is None else b.to(dev), residual=None if r is None else r.to(dev)).cpu().double()
    ref = x.double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    if r is not None:
        ref = ref + r.double()
    mag = x.double().abs() @ w.double().abs().t()
    return y2, y32, ref, mag


def test_f16x2_scale_rule():
    from multimodal_diffusion_amd import functional as Fn
    for bound in (1e-6, 0.04, 1.0, 22.6, 32768.0, 1e9):
        s = Fn.f16x2_scale(bound)
        assert math.log2(s) == int(math.log2(s)) and s * bound <= 2.0 ** 15 and 2 * s * bound * 1.01 > 2.0 ** 15
    assert Fn.f16x2_scale(0.0) == 1.0


@pytest.mark.parametrize("rows,K", [(5, 16), (300, 512), (1000, 2048)])
def test_split_f16x2_image(dev, rows, K):
    """(h + l) / s reproduces x to 22 bits (2^-22 relative, or 2^-25 / s absolute for elements far below the top of the
    range); l is at most half an ulp of h."""
    from multimodal_diffusion_amd import functional as Fn
    g = torch.Generator().manual_seed(rows + K)
    x = torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, 1, generator=g) * 2.0)
    img, s = Fn.split_f16x2(x.to(dev))
    pl = _h2_decode(img.cpu().numpy(), rows, K)
    assert np.isfinite(pl).all() and np.abs(pl[0]).max() <= 2.0 ** 15
    xd = x.numpy().astype(np.float64)
    err = np.abs(pl.sum(0) / s - xd)
    assert (err <= 2.0 ** -22 * np.abs(xd) + 2.0 ** -25 / s).all()
    assert (np.abs(pl[1]) <= np.abs(pl[0]) * 2.0 ** -11 + 2.0 ** -25).all()


@pytest.mark.parametrize("M,N,K", [(700, 512, 256), (333, 768, 2048), (5000, 1536, 512), (130, 256, 16), (26944, 512, 512)])
@pytest.mark.parametrize("mode", ["plain", "res", "gelu_split"])
def test_gemm_f16x2_accuracy(dev, M, N, K, mode):
    """Against fp64, next to the fp32-MFMA GEMM: the three-term product on 22-bit operands stays within a small multiple of the fp32
    chain's error (both share the fp32 accumulation rounding, which dominates at these K)."""
    from multimodal_diffusion_amd import functional as Fn, _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * 3.0
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    if mode == "gelu_split":
        ref = R.gelu_erf(R.linear(x.double(), w.double(), b.double()))
        x2, sx = Fn.split_f16x2(x.to(dev))
        w2, sw = Fn.split_f16x2(w.to(dev))
        so = Fn.f16x2_scale(float(ref.abs().max()))
        img = Fn.linear_f16x2(x2, M, w2, N, K, sx * sw, bias=b.to(dev), act=L.ACT_GELU, out_scale=so).cpu().numpy()
        y = torch.from_numpy(_h2_decode(img, M, N).sum(0) / so)
        y32 = Fn.linear(x.to(dev), w.to(dev), b.to(dev), act=L.ACT_GELU).cpu().double()
    else:
        y, y32, ref, _ = _gemm(dev, x, w, b, r if mode == "res" else None)
    e2 = (y.double() - ref).abs().max().item()
    e32 = (y32 - ref).abs().max().item()
    print(f"f16x2 gemm {M}x{N}x{K} {mode}: err {e2:.3e} (fp32-MFMA {e32:.3e}), |ref| {ref.abs().max().item():.3g}")
    assert e2 <= 4e-6 * ref.abs().max().item()
    assert e2 <= 2.5 * e32 + 1e-7, (e2, e32)


@pytest.mark.parametrize("B,N,H", [(2, 133, 4), (1, 421, 4), (3, 64, 8), (2, 37, 4)])
def test_attention_f16x2(dev, B, N, H):
    """in_proj epilogue -> f16x2 q|k|v image -> attention, against softmax(q k^T / 8) v in fp64 and the fp32-MFMA kernel."""
    from multimodal_diffusion_amd import functional as Fn, _lib as L
    d = H * 64
    g = torch.Generator().manual_seed(B * 1000 + N + H)
    qkv = torch.randn(B * N, 3 * d, generator=g) * 1.5
    bias = torch.randn(3 * d, generator=g) * 0.1
    lib = L.lib()
    img = torch.empty(lib.avd_qkv3_bytes(B, N, H), dtype=torch.uint8, device=dev)
    x2, sx = Fn.split_f16x2(qkv.to(dev))
    w2, sw = Fn.split_f16x2(torch.eye(3 * d).to(dev))
    sq = Fn.f16x2_scale(float((qkv + bias).abs().max()))
    bd = bias.to(dev)
    L.check(lib.avd_gemm_f16x2_qkv_f32(x2.data_ptr(), w2.data_ptr(), bd.data_ptr(), img.data_ptr(), B * N, N, H, 3 * d,
                                       0.125 * 1.4426950408889634, sx * sw, sq, L.stream_ptr(dev)))
    out = torch.empty(B, N, d, device=dev)
    L.check(lib.avd_attn_fwd_qkv_f16x2_f32(img.data_ptr(), out.data_ptr(), None, B, N, H, N, sq, 1.0, L.stream_ptr(dev)))
    full_ = (qkv + bias).double().view(B, N, 3, H, 64)
    q, k, v = (full_[:, :, i].transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) @ v).transpose(1, 2).reshape(B, N, d)
    e2 = rel_err(out.cpu(), ref)
    e32 = rel_err(Fn.attention((qkv + bias).view(B, N, 3 * d).to(dev), H).cpu(), ref)
    print(f"f16x2 attention B{B} N{N} H{H}: err {e2:.3e} (fp32-MFMA {e32:.3e})")
    assert e2 < 3e-6 and e2 < 3.0 * e32 + 2e-7, (e2, e32)
    # image output (the A operand of out_proj) carries the same values to 22 bits; rows >= n_query stay untouched
    o2 = torch.zeros(lib.avd_split3_bytes(B * N, d), dtype=torch.uint8, device=dev)
    nq = max(1, N - 5)
    L.check(lib.avd_attn_fwd_qkv_f16x2_f32(img.data_ptr(), None, o2.data_ptr(), B, N, H, nq, sq, sq, L.stream_ptr(dev)))
    got = (_h2_decode(o2.cpu().numpy(), B * N, d).sum(0) / sq).reshape(B, N, d)
    want = out.cpu().double().numpy()
    assert np.abs(got[:, :nq] - want[:, :nq]).max() <= 2.0 ** -21 * np.abs(want).max()
    assert not got[:, nq:].any()


def test_core_f16x2_vs_oracle_and_f32(dev, full):
    """MMDiT.forward at 16,840 rows: inside the fp32 parity tolerance, and within a small multiple of the fp32-MFMA path's own
    distance from the fp64 oracle."""
    ws, _ = full
    core2, _, _, _ = _full_modules(dev, ws)
    core32, _, _, _ = _full_modules(dev, ws)
    core2.matmul, core32.matmul = "f16x2", "f32"
    x = torch.randn(40, 421, 512, generator=torch.Generator().manual_seed(12))
    y2 = core2(x.to(dev)).cpu()
    y32 = core32(x.to(dev)).cpu()
    assert not torch.equal(y2, y32), "f16x2 path did not run"
    assert torch.isfinite(y2).all()
    sub = slice(0, 3)
    ref = R.mmdit_forward(x[sub].double(), {k: v.double() for k, v in ws["core"].items()}, 8, 8)
    e2, e32 = rel_err(y2[sub], ref), rel_err(y32[sub], ref)
    print(f"f16x2 core: err {e2:.3e} (fp32-MFMA {e32:.3e})")
    assert e2 < TOL and e32 < TOL
    assert e2 < 3.0 * e32 + 1e-7, (e2, e32)
    assert rel_err(y2, y32) < 3e-5


def test_full_step_f16x2_vs_oracle(dev, full):
    """BASELINE C3 shape, B=20 (16,840 rows), one CFG step against the CPU oracle; deterministic."""
    import multimodal_diffusion_amd as A
    ws, _ = full
    core, head, av, aa = _full_modules(dev, ws)
    B = 20
    g = torch.Generator().manual_seed(256)
    z_v = torch.randn(B, 8, 12, 32, 32, generator=g)
    z_a = torch.randn(B, 8, 150, generator=g)
    abar = R.alpha_bar_table(R.beta_table(1000))
    tn = torch.tensor([982, 500, 16, 999] * 5)
    tp = torch.tensor([966, 480, -1, 979] * 5)
    nb = 4
    ref = R.denoise_step_a2v(z_v[:nb], z_a[:nb], tn[:nb], tp[:nb], abar, adapt_v=ws["adapt_v"], adapt_a=ws["adapt_a"],
                             core=ws["core"], head=ws["head"], n_layers=8, n_heads=8, guidance=3.5)
    outs = {}
    for mode in ("f16x2", "f32"):
        eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video",
                              latent_shape=tuple(z_v.shape), prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode)
        eng.set_prompt(z_a.to(dev))
        outs[mode] = eng.step(z_v.to(dev), tn.to(dev), tp.to(dev))
        assert torch.equal(outs[mode], eng.step(z_v.to(dev), tn.to(dev), tp.to(dev)))
    e2, e32 = rel_err(outs["f16x2"][:nb].cpu(), ref), rel_err(outs["f32"][:nb].cpu(), ref)
    print(f"f16x2 step: err {e2:.3e} (fp32-MFMA {e32:.3e})")
    assert e2 < TOL and e2 < 3.0 * e32 + 1e-7, (e2, e32)


def test_chain_f16x2_tracks_f32(dev, full):
    """A 10-step DDIM + CFG trajectory at the C3 shape (B=20): within the chained tolerance of the fp32-MFMA mode (rel L2 <= 1e-3,
    SURVEY 8c), eager and as a replayed HIP graph — late steps carry |x| ~ 1e4..1e5 in the residual stream."""
    import multimodal_diffusion_amd as A
    from multimodal_diffusion_amd import schedule_utils as su
    ws, _ = full
    B = 20
    g = torch.Generator().manual_seed(99)
    z = torch.randn(B, 8, 12, 32, 32, generator=g).to(dev)
    za = torch.randn(B, 8, 150, generator=g).to(dev)
    abar = R.alpha_bar_table(R.beta_table(1000))
    sched = su.make_sampling_schedule(1000, 10)
    outs = {}
    for mode in ("f32", "bf16x3", "f16x2"):
        core, head, av, aa = _full_modules(dev, ws)
        eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video",
                              latent_shape=tuple(z.shape), prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode)
        eng.set_prompt(za)
        outs[mode] = eng.run(z, sched)
        if mode == "f16x2":
            outs["f16x2_graph"] = eng.run(z, sched, graph=True)
    ref = outs["f32"].double()
    assert torch.isfinite(outs["f16x2"]).all()
    l2 = {k: float((outs[k].double() - ref).norm() / ref.norm()) for k in ("bf16x3", "f16x2", "f16x2_graph")}
    print(f"10-step chain vs fp32-MFMA, rel L2: {l2}, |x| max {ref.abs().max().item():.3g}")
    assert l2["f16x2"] < 1e-3
    assert torch.equal(outs["f16x2"], outs["f16x2_graph"])


# ---------------------------------------------------------------------------------------------------- edges of the range
@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1.0, 1e12, 1e18])
def test_f16x2_dynamic_range(dev, scale):
    """The image scale absorbs the operand's magnitude: from 1e-30 to 1e18 the error stays a small multiple of the fp32 chain's."""
    g = torch.Generator().manual_seed(int(abs(math.log10(scale))) + 3)
    M, N, K = 300, 256, 512
    x = torch.randn(M, K, generator=g) * scale
    w = torch.randn(N, K, generator=g) * (1.0 / math.sqrt(K)) * (scale if scale > 1 else 1.0)
    y2, y32, ref, mag = _gemm(dev, x, w)
    assert torch.isfinite(y2).all()
    assert ((y2 - ref).abs() <= 2.0 * K * 2.0 ** -24 * mag).all()
    e2, e32 = (y2 - ref).abs().max().item(), (y32 - ref).abs().max().item()
    assert e2 <= 2.5 * e32 + 1e-30, (e2, e32)


def test_f16x2_wide_spread_inside_one_image(dev):
    """One image, magnitudes spread over 2^24: elements far below the top of the range keep only the ABSOLUTE accuracy
    2^-25 / s, which is what the documented bound promises — error <= 2^-21 sum|x w| + K 2^-24 max|x| max|w|."""
    g = torch.Generator().manual_seed(21)
    M, N, K = 260, 256, 512
    x = torch.randn(M, K, generator=g) * torch.exp2(-torch.randint(0, 24, (M, K), generator=g).float())
    w = torch.randn(N, K, generator=g) * torch.exp2(-torch.randint(0, 24, (N, K), generator=g).float())
    y2, y32, ref, mag = _gemm(dev, x, w)
    bound = 2.0 ** -21 * mag + K * 2.0 ** -24 * x.abs().max().item() * w.abs().max().item()
    assert ((y2 - ref).abs() <= bound).all()
    # fp16 subnormals in the planes are honoured by the matrix pipe: rows made ONLY of small elements are still accurate to
    # the absolute floor of the image (not flushed to a relative 2^-11)
    xs = torch.zeros(4, K)
    xs[0, 0] = 1.0                                              # fixes the scale: s = 2^14
    xs[1:] = torch.randn(3, K, generator=g) * 2.0 ** -20       # scaled to ~2^-6: the l plane is subnormal
    ws = torch.randn(N, K, generator=g)
    y2, _, ref, _ = _gemm(dev, xs, ws)
    floor = K * 2.0 ** -25 / 2.0 ** 14 * ws.abs().max().item() * 2
    print(f"small-element rows: err {(y2[1:] - ref[1:]).abs().max().item():.3e}, floor {floor:.3e}, "
          f"flushed-l would be {2.0 ** -31 * math.sqrt(K):.3e}")
    assert ((y2[1:] - ref[1:]).abs() <= floor + 2.0 ** -22 * (xs[1:].double().abs() @ ws.double().abs().t())).all()


def test_f16x2_cancellation(dev):
    """K-long catastrophic cancellation (products in +a, -a pairs): the error is bounded by the operands' 22 bits,
    3 * 2^-22 per product plus the fp32 accumulation both paths share."""
    g = torch.Generator().manual_seed(9)
    M, N, K = 260, 256, 2048
    a = torch.randn(M, K // 2, generator=g) * 100.0
    x = torch.empty(M, K)
    x[:, 0::2], x[:, 1::2] = a, -a
    x[:, 1::2] += torch.randn(M, K // 2, generator=g) * 1e-5
    w = torch.empty(N, K)
    wv = torch.randn(N, K // 2, generator=g)
    w[:, 0::2], w[:, 1::2] = wv, wv
    y2, y32, ref, mag = _gemm(dev, x, w)
    assert ((y2 - ref).abs() <= (K * 2.0 ** -24 + 3 * 2.0 ** -22) * mag).all()
    print(f"cancellation: f16x2 {(y2 - ref).abs().max().item():.3e}, fp32-MFMA {(y32 - ref).abs().max().item():.3e}")


def test_f16x2_out_of_range_is_loud(dev):
    """A scale that does not cover the data (or inf / NaN in it) must not saturate silently: the rows it touches come out NaN
    and every other row is untouched."""
    g = torch.Generator().manual_seed(13)
    x = torch.randn(300, 256, generator=g)
    w = torch.randn(256, 256, generator=g) / 16
    x[5, 17] = 1.0e3                       # far past the scale derived for |x| <= 8 below
    x[77, 3] = float("nan")
    x[200, 0] = float("inf")
    from multimodal_diffusion_amd import functional as Fn
    y2, _, ref, _ = _gemm(dev, x, w, sx=Fn.f16x2_scale(8.0))
    bad = torch.tensor([5, 77, 200])
    good = torch.ones(300, dtype=torch.bool)
    good[bad] = False
    assert torch.isnan(y2[bad]).all()
    assert torch.isfinite(y2[good]).all() and rel_err(y2[good], ref[good]) < 3e-6


def test_f16x2_scales_hold_for_adversarial_weights(dev, full):
    """The core's image scales come from bounds that hold for every input: blow up the norm gains, weights and biases, feed a
    residual stream of magnitude 1e6 and rows of zeros — the result stays finite and tracks the fp32-MFMA path."""
    ws, _ = full
    cores = []
    for mode in ("f16x2", "f32"):
        core, _, _, _ = _full_modules(dev, ws)
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for i, b in enumerate(core.blocks):
                b.norm1.scale.mul_(40.0 if i % 2 else 0.02)
                b.norm2.scale.mul_(0.03 if i % 2 else 25.0)
                b.attn.mha.in_proj_weight.mul_(0.05 if i % 2 else 3.0)
                b.attn.mha.in_proj_bias.add_(torch.randn(1536, generator=g).to(dev) * 5.0)
                b.mlp.fc1.weight.mul_(0.1 if i % 2 else 2.0)
                b.mlp.fc1.bias.add_(torch.randn(2048, generator=g).to(dev) * 3.0)
                b.mlp.fc2.weight.mul_(1e-3 if i % 3 else 1.0)
                b.attn.mha.out_proj.weight.mul_(1e-2 if i % 3 == 1 else 1.0)
        core.matmul = mode
        cores.append(core)
    x = torch.randn(16, 421, 512, generator=torch.Generator().manual_seed(3))
    x[0] *= 1e6
    x[1] *= 1e-6
    x[2, :100] = 0.0
    x[3, 7, 5] = 3e7                                  # one huge outlier channel
    y2 = cores[0](x.to(dev)).cpu()
    y32 = cores[1](x.to(dev)).cpu()
    assert not torch.equal(y2, y32)
    assert torch.isfinite(y2).all() and torch.isfinite(y32).all()
    for s in range(16):
        assert rel_err(y2[s], y32[s].double()) < 1e-4, s


def test_f16x2_engine_follows_weight_updates(dev, full):
    """The scales are cached per parameter version: an in-place weight update (x64 on a norm gain with in_proj / 64, so the
    attention logits keep their conditioning; x32 on fc1 with fc2 / 32) is picked up — a stale scale would overflow fp16."""
    ws, _ = full
    core, _, _, _ = _full_modules(dev, ws)
    ref_core, _, _, _ = _full_modules(dev, ws)
    core.matmul, ref_core.matmul = "f16x2", "f32"
    x = torch.randn(16, 421, 512, generator=torch.Generator().manual_seed(8)).to(dev)
    y0 = core(x)
    for c in (core, ref_core):
        with torch.no_grad():
            c.blocks[0].norm1.scale.mul_(64.0)
            c.blocks[0].attn.mha.in_proj_weight.mul_(1.0 / 64.0)
            c.blocks[3].mlp.fc1.weight.mul_(32.0)
            c.blocks[3].mlp.fc2.weight.mul_(1.0 / 32.0)
    y1, r1 = core(x), ref_core(x)
    assert torch.isfinite(y1).all() and not torch.equal(y0, y1)
    assert rel_err(y1.cpu(), r1.cpu().double()) < 1e-4


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("act", ["gelu", "leaky_relu"])
def test_head_split_modes_vs_oracle(dev, full, mode, act):
    """MultiModalNoiseHead with its Linears on the split-operand kernels (input rows -> image, bias-only image epilogue, LayerNorm +
    activation -> image): same tolerance as the fp32 path, 6,736 rows; the audio path (d_out = 32) stays on the fp32 kernels."""
    import multimodal_diffusion_amd as A
    ws, _ = full
    heads = {}
    for m in (mode, "f32"):
        head = A.MultiModalNoiseHead({"video": 512, "audio": 512}, {"video": 256, "audio": 32}, hidden_dim=512, activation=act).eval()
        head.load_state_dict(ws["head"], strict=True)
        head.to(dev)
        head.matmul = m
        heads[m] = head
    g = torch.Generator().manual_seed(31)
    x = torch.randn(16, 421, 512, generator=g) * 1.7
    x[3] *= 30.0
    out = heads[mode]({"video": x.to(dev), "audio": x[:2].to(dev)})
    out32 = heads["f32"]({"video": x.to(dev), "audio": x[:2].to(dev)})
    assert not torch.equal(out["video"], out32["video"]), "split path did not run"
    assert torch.equal(out["audio"], out32["audio"])
    hw = {k: v.double() for k, v in ws["head"].items()}
    ref = R.noise_head(x[:4].double(), hw, "video", activation=act)
    e, e32 = rel_err(out["video"][:4].cpu(), ref), rel_err(out32["video"][:4].cpu(), ref)
    print(f"head {mode} {act}: err {e:.3e} (fp32-MFMA {e32:.3e})")
    assert e < TOL and e < 3.0 * e32 + 1e-7, (e, e32)


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
def test_two_stream_cfg_halves_are_bit_identical(dev, full, mode):
    """split_streams runs the cond / null halves as two kernel chains on two HIP streams: every output element is the same sum in
    the same order, so a step, a trajectory and a replayed graph must equal the single-stream results bit for bit.  (The f16x2
    engine turns the two-stream layout on by itself at >= 6144 rows per half.)"""
    import multimodal_diffusion_amd as A
    from multimodal_diffusion_amd import schedule_utils as su
    ws, _ = full
    B = 16
    g = torch.Generator().manual_seed(77)
    z = torch.randn(B, 8, 12, 32, 32, generator=g).to(dev)
    za = torch.randn(B, 8, 150, generator=g).to(dev)
    abar = R.alpha_bar_table(R.beta_table(1000))
    sched = su.make_sampling_schedule(1000, 4)
    outs = {}
    for split in (False, True):
        core, head, av, aa = _full_modules(dev, ws)
        eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z.shape),
                              prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode, split_streams=split)
        eng.set_prompt(za)
        outs[split] = (eng.run(z, sched, graph=False), eng.run(z, sched, graph=True))
    assert torch.isfinite(outs[True][0]).all()

    def same(a, b, what):
        d = (a - b).abs()
        bad = ((d > 0) | torch.isnan(d)).nonzero()
        assert torch.equal(a, b), f"{what}: max |diff| {float(d.max()):.3e} in {len(bad)} elements of samples {bad[:, 0].unique().tolist()}"

    same(outs[False][0], outs[True][0], "two streams vs one, eager")
    same(outs[False][1], outs[True][1], "two streams vs one, graph replay")
    same(outs[True][0], outs[True][1], "two streams: eager vs graph replay")
    core, head, av, aa = _full_modules(dev, ws)
    auto = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z.shape),
                           prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode)
    assert auto._split_streams == (mode == "f16x2")          # 16 x 421 = 6736 rows per half


@pytest.mark.parametrize("mode", ["bf16x3", "auto"])
def test_two_stream_halves_take_the_step_kernel_family(dev, full, mode):
    """in the six-term mode the split kernels engage at 2,048 rows of the stacked 2B x N batch.  At B = 3 (2,526 rows; one CFG
    half alone: 1,263) a forced two-stream step must take the SAME kernels as the one-stream layout — the family is chosen once per step —
    so the two layouts stay bit-identical in that band too (and the step really runs on the split kernels: kernel tags)."""
    import multimodal_diffusion_amd as A
    from multimodal_diffusion_amd import _lib as L
    ws, _ = full
    B = 3
    g = torch.Generator().manual_seed(303)
    z = torch.randn(B, 8, 12, 32, 32, generator=g).to(dev)
    za = torch.randn(B, 8, 150, generator=g).to(dev)
    abar = R.alpha_bar_table(R.beta_table(1000))
    tn, tp = torch.tensor([982, 500, 16]).to(dev), torch.tensor([966, 480, -1]).to(dev)
    outs, tags = {}, {}
    for split in (False, True):
        core, head, av, aa = _full_modules(dev, ws)
        eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z.shape),
                              prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode, split_streams=split)
        eng.set_prompt(za)
        eng.step(z, tn, tp)
        L.check(L.lib().avd_prof_enable(1))
        try:
            outs[split] = eng.step(z, tn, tp)
            torch.cuda.synchronize()
        finally:
            L.check(L.lib().avd_prof_enable(0))
        tags[split] = {k for k, v in L.prof_report().items() if v[0] > 0}
    assert any(k.startswith("gemm_bf16x3") for k in tags[False]), tags[False]
    # the same split-operand kernels in both layouts (the fp32 launches of the noise head may pick another tile / ring depth for the
    # half-size call: those variants are bit-identical by construction and are checked by the equality below)
    fam = lambda t: {k for k in t if k.startswith(("gemm_bf16x3", "attn_bf16x3"))}
    assert fam(tags[True]) == fam(tags[False]), (tags[False], tags[True])
    assert torch.equal(outs[False], outs[True])


def test_f16x2_below_and_at_the_row_threshold(dev, full):
    """Below 6144 rows the f16x2 request keeps the fp32-MFMA kernels (the 256-row tiles would not fill the chip); at the threshold
    (128x128 at B=32: 8512 core rows, exactly 6144 head rows) the split kernels take over — the step workspace is sized for
    whichever path runs, and both agree with the f32 mode."""
    import multimodal_diffusion_amd as A
    ws, _ = full
    abar = R.alpha_bar_table(R.beta_table(1000))
    for B, hw in ((4, 32), (32, 16)):
        g = torch.Generator().manual_seed(B)
        z = torch.randn(B, 8, 12, hw, hw, generator=g).to(dev)
        za = torch.randn(B, 8, 150, generator=g).to(dev)
        tn, tp = torch.full((B,), 500, device=dev), torch.full((B,), 480, device=dev)
        outs = {}
        for mode in ("f32", "f16x2"):
            core, head, av, aa = _full_modules(dev, ws)
            eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z.shape),
                                  prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode)
            eng.set_prompt(za)
            outs[mode] = eng.step(z, tn, tp)
        assert torch.isfinite(outs["f16x2"]).all()
        # (not bit-equal below the threshold either: the f32 mode folds the RMSNorms into the GEMM epilogues, the fallback does not)
        assert float((outs["f32"] - outs["f16x2"]).abs().max()) < 2e-5 * max(1.0, float(outs["f32"].abs().max()))


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
def test_full_width_step_vs_reference_golden(dev, full, mode):
    """Fixture g17: one CFG step at the bench's full model width, batch 8 (6,736 rows — every matrix-pipe mode engages, the head's
    split path too), computed by the reference's own modules.  Every mode must reproduce the reference's next latents."""
    import multimodal_diffusion_amd as A
    from conftest import load_golden
    from test_oracle_golden import g17_inputs
    ws, _ = full
    g = load_golden("g17_full_step_c3.npz")
    z_v, z_a = g17_inputs()
    abar = R.alpha_bar_table(R.beta_table(1000))
    core, head, av, aa = _full_modules(dev, ws)
    eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z_v.shape),
                          prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=mode)
    eng.set_prompt(z_a.to(dev))
    out = eng.step(z_v.to(dev), torch.from_numpy(g["t_now"]).to(dev), torch.from_numpy(g["t_prev"]).to(dev)).cpu()
    e = rel_err(out[:2], g["z_next01"])
    print(f"full-width step vs the reference, {mode}: {e:.3e}")
    assert torch.isfinite(out).all() and e < TOL


def test_full_step_v2a_f16x2(dev, full):
    """Video -> audio direction in f16x2: 37 target + 384 prompt tokens at 256x256 (target rows first, the 32-wide audio head stays
    on fp32 MFMA), B=10 -> 8,420 rows; oracle on the first two samples; two streams off (4,210 rows per half)."""
    import multimodal_diffusion_amd as A
    ws, _ = full
    core, head, av, aa = _full_modules(dev, ws)
    g = torch.Generator().manual_seed(78)
    B = 10
    z_v = torch.randn(B, 8, 12, 32, 32, generator=g)
    z_a = torch.randn(B, 8, 150, generator=g)
    abar = R.alpha_bar_table(R.beta_table(1000))
    tn = torch.tensor([982, 19] * 5)
    tp = torch.tensor([966, -1] * 5)
    ref = R.denoise_step_v2a(z_a[:2], z_v[:2], tn[:2], tp[:2], abar, adapt_v=ws["adapt_v"], adapt_a=ws["adapt_a"], core=ws["core"],
                             head=ws["head"], n_layers=8, n_heads=8, guidance=3.0)
    outs = {}
    for mode in ("f16x2", "f32"):
        eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=_full_modules(dev, ws)[0], head=head, tstep_dim=256, target="audio",
                              latent_shape=tuple(z_a.shape), prompt_tokens=384, alpha_bar=abar, guidance=3.0, matmul=mode)
        eng.set_prompt(z_v.to(dev))
        outs[mode] = eng.step(z_a.to(dev), tn.to(dev), tp.to(dev))
        assert not eng._split_streams
    e2, e32 = rel_err(outs["f16x2"][:2].cpu(), ref), rel_err(outs["f32"][:2].cpu(), ref)
    print(f"V->A step: f16x2 {e2:.3e}, fp32-MFMA {e32:.3e}")
    assert e2 < TOL and not torch.equal(outs["f16x2"], outs["f32"])


def test_full_step_f16x2_512(dev, full):
    """BASELINE C5 geometry (512x512: 1536+37 tokens, ragged 1573 -> 1600 padded keys), B=8 -> 25,168 rows, the f16x2 engine's default
    layout (two streams, 12,584 rows per half); sample 0 against the CPU oracle."""
    import multimodal_diffusion_amd as A
    ws, _ = full
    core, head, av, aa = _full_modules(dev, ws)
    B = 8
    g = torch.Generator().manual_seed(512)
    z_v = torch.randn(B, 8, 12, 64, 64, generator=g)
    z_a = torch.randn(B, 8, 150, generator=g)
    abar = R.alpha_bar_table(R.beta_table(1000))
    tn = torch.tensor([982, 500, 16, 999, 700, 300, 64, 5])
    tp = torch.tensor([966, 480, -1, 979, 680, 280, 48, -1])
    ref = R.denoise_step_a2v(z_v[:1], z_a[:1], tn[:1], tp[:1], abar, adapt_v=ws["adapt_v"], adapt_a=ws["adapt_a"],
                             core=ws["core"], head=ws["head"], n_layers=8, n_heads=8, guidance=3.5)
    eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video",
                          latent_shape=tuple(z_v.shape), prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul="f16x2")
    assert eng._split_streams
    eng.set_prompt(z_a.to(dev))
    out = eng.step(z_v.to(dev), tn.to(dev), tp.to(dev))
    e = rel_err(out[:1].cpu(), ref)
    print(f"512x512 step, f16x2: {e:.3e}")
    assert torch.isfinite(out).all() and e < TOL


def test_sample_one_direction_with_f16x2_modules(dev, full):
    """The reference-signature sampler with the core / head switched to "f16x2" on a small clip (B=1: far below 6144 rows, so the
    request must be harmless — the fp32 kernels run) produces the frames of the f32 run to within 1 LSB."""
    import multimodal_diffusion_amd as A
    from multimodal_diffusion_amd.audio_codec import AudioCodec
    ws, _ = full
    torch.manual_seed(3)
    vae = A.VideoVAE.from_config({"latent": {"channels": 8, "t_down": 4, "s_down": 8}}).eval().to(dev)
    codec = AudioCodec.from_config({"sr": 16000, "latent": {"channels": 8, "frames_per_clip": 150},
                                    "codec": {"hop_samples": 320}}).eval().to(dev)
    cfg = {"tokenizer": {"width": 512, "video": {"tube": {"t": 2, "h": 4, "w": 4}}, "audio": {"chunk": {"length": 4, "stride": 4}}},
           "video": {"fps": 16, "size": [32, 32], "latent": {"channels": 8, "t_down": 4, "s_down": 8}},
           "audio": {"sr": 16000, "latent": {"channels": 8, "frames_per_clip": 150}},
           "data": {"clip_seconds": 0.5},
           "diffusion": {m: {"steps": 1000, "sampler_steps": 3, "schedule": "cosine", "min_beta": 1e-4, "max_beta": 0.02}
                         for m in ("video", "audio")},
           "sampling": {"ddim_eta": 0.0, "guidance_scale": {"video": 2.0, "audio": 2.0}}}
    wav = (0.1 * torch.randn(48000, generator=torch.Generator().manual_seed(5))).numpy()
    frames = {}
    for mode in ("f32", "f16x2"):
        core, head, av, aa = _full_modules(dev, ws)
        core.matmul = head.matmul = mode
        torch.manual_seed(77)
        res = A.sample_one_direction(cfg=cfg, vid_vae=vae, aud_codec=codec, adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256,
                                     device=dev, prompt_modality="audio", prompt_video=None, prompt_audio=wav)
        frames[mode] = res["video"].astype(np.int32)
    assert np.abs(frames["f32"] - frames["f16x2"]).max() <= 1


def test_weight_bounds_kernel(dev):
    """avd_weight_bounds_f32 (what the f16x2 scales are derived from): max |w| and the largest row 2-norm, exact against fp64;
    NaN anywhere is reported as NaN, and f16x2_scale refuses a non-finite bound."""
    from multimodal_diffusion_amd import functional as Fn, _lib as L
    g = torch.Generator().manual_seed(3)
    ws = [torch.randn(1536, 512, generator=g) * 0.05, torch.randn(7, 13, generator=g), torch.randn(2048, generator=g), torch.zeros(4, 16)]
    ws[0][77, 5] = -9.5
    got = Fn.weight_bounds([w.to(dev) for w in ws])
    for w, (amax, nrm) in zip(ws, got):
        w2 = w.double().reshape(-1, w.shape[-1])
        assert amax == float(w.abs().max())
        assert abs(nrm - float(w2.norm(dim=1).max())) <= 1e-6 * max(1.0, nrm)
    bad = torch.randn(8, 32, generator=g)
    bad[3, 4] = float("nan")
    a, n = Fn.weight_bounds([bad.to(dev)])[0]
    assert math.isnan(a) and math.isnan(n)
    with pytest.raises(L.AvdError):
        Fn.f16x2_scale(a)


# ------------------------------------------------------------------------------------------------- VideoVAE convolutions on f16x2
def test_vae_decode_f16x2(dev):
    """Decoder convolutions with two scaled fp16 planes per operand (the first image's scale derived on the device from
    max |from_lat(z)|, the later ones from the GroupNorm bound): golden fixture, ragged tiles, chunked batch, and the fp64 oracle at
    128x128 next to the fp32-MFMA decoder; latents of magnitude 1e4 and 1e-4 (the scale follows the data); NaN stays loud."""
    import multimodal_diffusion_amd as A
    from conftest import load_golden, split_weights
    from test_gpu_parity import G, _vae_from
    g = load_golden("g11_vae_decode.npz")
    vae = _vae_from(split_weights(g)["w"], dev)
    vae.matmul = "f16x2"
    x = vae.decode(G(g["z"], dev)).cpu()
    assert rel_err(x, g["x"]) < TOL
    assert rel_err(vae.decode(G(g["z"][:1], dev), out_size=(6, 24, 40)).cpu(), g["x_odd"]) < TOL
    assert torch.equal(vae.decode(G(g["z"], dev), max_workspace_bytes=1).cpu(), x)
    W = R.synth_vae_decoder(seed=3, n_blocks=3)
    z = torch.randn(1, 8, 3, 16, 16, generator=torch.Generator().manual_seed(4))
    errs = {}
    for scale in (1.0, 1e4, 1e-4):
        ref = R.vae_decode((z * scale).double(), {k: v.double() for k, v in W.items()}, n_blocks=3, out_act="tanh")
        for mode in ("f32", "f16x2"):
            v = A.VideoVAE(A.VideoVAEConfig(dec_blocks=3, out_activation="tanh")).eval()
            v.load_state_dict(W, strict=False)
            v.matmul = mode
            out = v.to(dev).decode((z * scale).to(dev)).cpu()
            assert torch.isfinite(out).all()
            errs[(mode, scale)] = rel_err(out, ref)
        print(f"vae decode |z| x {scale:g}: f16x2 {errs[('f16x2', scale)]:.3e}, fp32-MFMA {errs[('f32', scale)]:.3e}")
        assert errs[("f16x2", scale)] < TOL and errs[("f16x2", scale)] < 3.0 * errs[("f32", scale)] + 1e-7, errs
    zn = z.clone()
    zn[0, 3, 1, 5, 5] = float("nan")
    v.matmul = "f16x2"
    assert torch.isnan(v.decode(zn.to(dev))).any()


def test_vae_encode_f16x2(dev):
    """Encoder with its 64 -> 64 convolution(s) on f16x2: golden fixture at the fp32 encoder's tolerance."""
    import warnings
    import multimodal_diffusion_amd as A
    from conftest import load_golden, split_weights
    from test_gpu_parity import G
    g = load_golden("g12_vae_encode.npz")
    vae = A.VideoVAE.from_config({"latent": {"channels": 8, "t_down": 4, "s_down": 8}}).eval()
    vae.load_state_dict(split_weights(g)["w"], strict=False)
    vae = vae.to(dev)
    vae.matmul = "f32"
    z32 = vae.encode(G(g["x"], dev)).cpu()
    vae.matmul = "f16x2"
    z2 = vae.encode(G(g["x"], dev)).cpu()
    assert rel_err(z2, g["z"]) < TOL and rel_err(z2, z32) < 2e-5
    assert not torch.equal(z2, z32) or len(vae.enc_net) == 1, "f16x2 encoder path did not run"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert rel_err(vae.encode(G(g["x_crop"], dev)).cpu(), g["z_crop"]) < TOL


def test_vae_decode_f16x2_256(dev):
    """256x256 (the bench's clip size), one sample: f16x2 against the fp32-MFMA decoder (which test_vae_decode_256_vs_oracle pins
    to the oracle) — the GroupNorm bound at n = 25 M elements per group leaves typical values 12 binades below the fp16 top."""
    import multimodal_diffusion_amd as A
    torch.manual_seed(0)
    vae = A.VideoVAE.from_config({"latent": {"channels": 8, "t_down": 4, "s_down": 8}}).eval().to(dev)
    with torch.no_grad():
        for p in vae.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    z = torch.randn(1, 8, 12, 32, 32, generator=torch.Generator().manual_seed(6)).to(dev)
    vae.matmul = "f32"
    x32 = vae.decode(z)
    vae.matmul = "f16x2"
    x2 = vae.decode(z)
    assert torch.isfinite(x2).all() and not torch.equal(x2, x32)
    e = rel_err(x2.cpu(), x32.cpu().double())
    print(f"vae decode 256x256, f16x2 vs fp32-MFMA: {e:.3e}")
    assert e < 2e-5


def test_block_stagger_changes_timing_only(dev, full):
    """The start-up stagger of the co-resident blocks of the 256x128 split GEMM (automatic on single-stream launches; `s3_stagger`
    forces it) is a scheduling device: results are bit-identical with it off, automatic and forced large."""
    from multimodal_diffusion_amd import _lib as L
    ws, _ = full
    core, _, _, _ = _full_modules(dev, ws)
    core.matmul = "f16x2"
    x = torch.randn(64, 421, 512, generator=torch.Generator().manual_seed(21)).to(dev)      # 26,944 rows: two generations of blocks
    outs = []
    for v in (0, -1, 40):
        L.check(L.lib().avd_tune_set(b"s3_stagger", v))
        try:
            outs.append(core(x))
        finally:
            L.check(L.lib().avd_tune_set(b"s3_stagger", -1))
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_fp8_attention_on_f16x2_images(dev, full):
    """BASELINE C5 as named (512x512, "fp8 MFMA attention"), with f16x2 projections: the e4m3 attention reads the f16x2 q|k|v image
    and writes its result as an f16x2 image.  Reduced precision — the error against the fp32 oracle is reported and bounded at 2.2x the
    measured 1.5e-3 of max|z|, and must be of the size the bf16x3 + fp8 combination has."""
    import multimodal_diffusion_amd as A
    ws, mods = full
    core, head, av, aa = mods
    g = torch.Generator().manual_seed(512)
    B = 8
    z_v = torch.randn(B, 8, 12, 64, 64, generator=g)
    z_a = torch.randn(B, 8, 150, generator=g)
    abar = R.alpha_bar_table(R.beta_table(1000))
    tn = torch.tensor([982, 500, 16, 999] * 2)
    tp = torch.tensor([966, 480, -1, 979] * 2)
    ref = R.denoise_step_a2v(z_v[:1], z_a[:1], tn[:1], tp[:1], abar, adapt_v=ws["adapt_v"], adapt_a=ws["adapt_a"], core=ws["core"],
                             head=ws["head"], n_layers=8, n_heads=8, guidance=3.5)
    errs = {}
    for matmul, attn in (("f16x2", "default"), ("f16x2", "fp8"), ("bf16x3", "fp8")):
        eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z_v.shape),
                              prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul=matmul, attn=attn)
        eng.set_prompt(z_a.to(dev))
        out = eng.step(z_v.to(dev), tn.to(dev), tp.to(dev)).cpu()
        assert torch.isfinite(out).all()
        errs[(matmul, attn)] = rel_err(out[:1], ref)
    print(f"C5 step: {errs}")
    assert errs[("f16x2", "default")] < TOL
    assert 1e-5 < errs[("f16x2", "fp8")] < 3.3e-3         # measured 1.5e-3 (round 4)
    assert errs[("f16x2", "fp8")] < 2.0 * errs[("bf16x3", "fp8")]


@pytest.mark.parametrize("seed", range(10))
def test_f16x2_fuzz_weight_and_input_scales(dev, seed):
    """Seeded fuzz of the scale derivation: a 2-layer core whose every parameter tensor is multiplied by an independent log-uniform
    factor in [1e-2, 1e2] (gains, weights, biases), inputs of magnitude 10^U(-3, 4) — the f16x2 forward must stay finite and no
    further from the fp64 oracle than 3x the fp32-MFMA path's own error (or the parity tolerance, whichever is larger)."""
    import multimodal_diffusion_amd as A
    g = torch.Generator().manual_seed(1000 + seed)
    ws = R.synth_weights(seed=seed, n_layers=2)["core"]
    for k in ws:
        if k.endswith("in_proj_weight"):
            f = 10.0 ** float(torch.empty(1).uniform_(-1.0, 0.5, generator=g))     # keeps the attention logits conditioned
        else:
            f = 10.0 ** float(torch.empty(1).uniform_(-2.0, 2.0, generator=g))
        ws[k] = ws[k] * f
    x = torch.randn(16, 421, 512, generator=g) * 10.0 ** float(torch.empty(1).uniform_(-3.0, 4.0, generator=g))
    ref = R.mmdit_forward(x[:1].double(), {k: v.double() for k, v in ws.items()}, 2, 8)
    errs = {}
    for mode in ("f32", "f16x2"):
        core = A.MMDiT(d_model=512, n_layers=2, n_heads=8, mlp_ratio=4.0).eval()
        core.load_state_dict(ws, strict=True)
        core = core.to(dev)
        core.matmul = mode
        y = core(x.to(dev)).cpu()
        assert torch.isfinite(y).all(), mode
        errs[mode] = rel_err(y[:1], ref)
    assert errs["f16x2"] < max(TOL, 3.0 * errs["f32"]), errs


def test_captured_graph_refuses_stale_scales(dev, full):
    """a captured f16x2 step holds the image scales BY VALUE (kernel arguments).  An in-place update that moves max|w|
    across a power of two (here a norm gain x64 with in_proj / 64) leaves every pointer unchanged — the replay must raise instead
    of dividing by the old scales; an update that keeps the scales replays and follows the new weights; capturing again works."""
    import multimodal_diffusion_amd as A
    from multimodal_diffusion_amd import schedule_utils as su, _lib as L
    ws, _ = full
    B = 4
    g = torch.Generator().manual_seed(9)
    z = torch.randn(B, 8, 12, 32, 32, generator=g).to(dev)
    za = torch.randn(B, 8, 150, generator=g).to(dev)
    abar = R.alpha_bar_table(R.beta_table(1000))
    sched = su.make_sampling_schedule(1000, 6)
    core, head, av, aa = _full_modules(dev, ws)
    eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z.shape),
                          prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul="f16x2")
    eng.set_prompt(za)
    eng.begin(sched)
    a, b = z.clone(), torch.empty_like(z)
    eng.advance(a, b)                       # lazy init outside capture
    eng.rewind()
    a.copy_(z)
    graph = eng.capture_pair(a, b)
    graph.replay()
    torch.cuda.synchronize()
    first = a.clone()
    # an update that keeps every scale: a bias moves (biases are far below the bounds' power-of-two steps)
    with torch.no_grad():
        core.blocks[2].mlp.fc2.bias.add_(1e-3)
    eng.rewind()
    a.copy_(z)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and not torch.equal(a, first), "the replay did not follow an in-place update behind unchanged pointers"
    # an update that moves scales by 2^6
    with torch.no_grad():
        core.blocks[0].norm1.scale.mul_(64.0)
        core.blocks[0].attn.mha.in_proj_weight.mul_(1.0 / 64.0)
    eng.rewind()
    a.copy_(z)
    with pytest.raises(L.AvdError, match="capture again"):
        graph.replay()
    with pytest.raises(L.AvdError, match="capture again"):
        graph.replay()                      # stays refused
    graph2 = eng.capture_pair(a, b)
    eng.rewind()
    a.copy_(z)
    graph2.replay()
    torch.cuda.synchronize()
    ref_eng = A.DenoiseEngine(adapt_v=av, adapt_a=aa, core=core, head=head, tstep_dim=256, target="video", latent_shape=tuple(z.shape),
                              prompt_tokens=37, alpha_bar=abar, guidance=3.5, matmul="f16x2")
    ref_eng.set_prompt(za)
    ref_eng.begin(sched)
    c, d = z.clone(), torch.empty_like(z)
    ref_eng.advance(c, d)
    ref_eng.advance(d, c)
    assert torch.equal(a, c)
    # the pair invalidated above stays refused although the engine has captured again since
    with pytest.raises(L.AvdError, match="capture again"):
        graph.replay()
    # a RE-ALLOCATED parameter (new storage behind the same module attribute): the graph holds the old device pointer
    with torch.no_grad():
        core.blocks[1].mlp.fc1.bias = torch.nn.Parameter(core.blocks[1].mlp.fc1.bias.detach().clone() + 1e-3)
    eng.rewind()
    a.copy_(z)
    with pytest.raises(L.AvdError, match="capture again"):
        graph2.replay()
    with pytest.raises(L.AvdError, match="capture again"):
        graph2.replay()                     # stays refused: a second replay must not slip through on the refreshed tables
    # ... while eager steps and a fresh capture follow the new tables, and a later in-place update does not raise spuriously
    eng.rewind()
    a.copy_(z)
    eng.advance(a, b)
    eng.advance(b, a)
    torch.cuda.synchronize()
    eager = a.clone()
    graph3 = eng.capture_pair(a, b)
    eng.rewind()
    a.copy_(z)
    graph3.replay()
    torch.cuda.synchronize()
    assert torch.equal(a, eager)
    with torch.no_grad():
        core.blocks[2].mlp.fc2.bias.add_(1e-3)
    eng.rewind()
    a.copy_(z)
    graph3.replay()                         # pointers and scales unchanged: still valid, follows the update
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and not torch.equal(a, eager)
    out = eng.run(z, sched, graph=True)     # run() captures its own pair; its local graph leaves no state behind
    with torch.no_grad():
        core.blocks[2].mlp.fc2.bias.add_(1e-3)
    assert torch.isfinite(eng.run(z, sched, graph=False)).all() and torch.isfinite(out).all()
