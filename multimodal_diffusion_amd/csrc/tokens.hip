// Token plumbing and the DDIM update — the HBM-bound ends of a denoising step.
//   timestep embedding ........ avdiff/utils/schedule_utils.py:64-86
//   tube patch / unpatch ...... avdiff/utils/ops.py:100-119, 122-144
//   audio chunk / overlap-add . avdiff/models/infer/sample_clip.py:184-188, 191-215 (ops.py:17-45, 48-93)
//   ddim_step ................. avdiff/utils/schedule_utils.py:146-200
//   CFG + unpatch + DDIM ...... avdiff/models/infer/sample_clip.py:381-389 (video), :342-348 (audio)
//   sequence assembly ......... avdiff/models/infer/sample_clip.py:367-371,377 / 328-333,338
// All kernels move 16 bytes per lane, coalesced along the innermost latent axis (W or the feature axis),
// and touch every byte exactly once: fused CFG+unpatch+DDIM reads 2 eps + z and writes z' = 16 B per latent
// element, which is its HBM roofline.
#include "avd_common.h"

#include <stdlib.h>

namespace avd {

// ------------------------------------------------------------------ timestep embedding
__global__ void temb_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs, float* __restrict__ out,
                            int B, int dim, float neg_log_mp) {
    const int half = dim >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, c = i % dim;
    float v = 0.f;
    if (c < 2 * half) {
        const int k = c < half ? c : c - half;
        // freqs = exp(-ln(max_period) * k / half), all in fp32 like the reference
        // (the caller may pass the host-built table: cos/sin at t ~ 1000 amplify a 1-ulp difference in exp 1000x)
        const float f = freqs ? freqs[k] : expf(neg_log_mp * (float)k / (float)half);
        const float a = (float)t[b] * f;
        v = c < half ? cosf(a) : sinf(a);
    }
    out[i] = v;
}

int temb_f32(const int64_t* t, const float* freqs, float* out, int B, int dim, float max_period, hipStream_t st) {
    AVD_REQUIRE(t && out, AVD_EINVAL, "timestep_embedding: null pointer");
    AVD_REQUIRE(B > 0 && dim > 0 && max_period > 0.f, AVD_EINVAL, "timestep_embedding: bad dims");
    const int n = B * dim;
    hipLaunchKernelGGL(temb_kernel, dim3((n + 255) / 256), dim3(256), 0, st, t, freqs, out, B, dim,
                       -(float)log((double)max_period));
    AVD_CHECK_LAUNCH("timestep_embedding");
    return AVD_OK;
}

// ------------------------------------------------------------------ tube geometry
struct Tube {
    int C, T, H, W, t, h, w;
    int Ht, Wt;        // H/h, W/w
    int D;             // C*t*h*w
    int64_t per;       // C*T*H*W
};

static int make_tube(Tube& g, int C, int T, int H, int W, int t, int h, int w) {
    AVD_REQUIRE(C > 0 && T > 0 && H > 0 && W > 0 && t > 0 && h > 0 && w > 0, AVD_EINVAL, "tube: bad dims");
    AVD_REQUIRE(T % t == 0 && H % h == 0 && W % w == 0, AVD_EINVAL, "tube sizes must divide latent dims");
    AVD_REQUIRE(w % 4 == 0, AVD_EUNSUPPORTED, "tube: w=%d must be a multiple of 4 (16-byte runs)", w);
    g = Tube{C, T, H, W, t, h, w, H / h, W / w, C * t * h * w, (int64_t)C * T * H * W};
    return AVD_OK;
}

// latent float4 index e4 (within one sample, NCDHW order) -> offset of the same 4 floats inside the
// sample's token matrix [Nv, D]
__device__ __forceinline__ int64_t tube_tok_off(const Tube& g, int64_t e4) {
    const int W4 = g.W >> 2;
    const int w0 = (int)(e4 % W4) * 4;
    int64_t r = e4 / W4;
    const int y = (int)(r % g.H);
    r /= g.H;
    const int tt = (int)(r % g.T);
    const int c = (int)(r / g.T);
    const int n = ((tt / g.t) * g.Ht + y / g.h) * g.Wt + w0 / g.w;
    const int k = ((c * g.t + tt % g.t) * g.h + y % g.h) * g.w + w0 % g.w;
    return (int64_t)n * g.D + k;
}

template <bool TO_TOKENS>
__global__ __launch_bounds__(256) void tube_kernel(const float* __restrict__ src, float* __restrict__ dst, Tube g,
                                                   int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int64_t per4 = g.per >> 2;
    const int64_t b = i / per4, e4 = i % per4;
    const int64_t lat = b * g.per + e4 * 4;
    const int64_t tok = b * g.per + tube_tok_off(g, e4);
    if (TO_TOKENS) *reinterpret_cast<f32x4*>(dst + tok) = *reinterpret_cast<const f32x4*>(src + lat);
    else *reinterpret_cast<f32x4*>(dst + lat) = *reinterpret_cast<const f32x4*>(src + tok);
}

int tube_patch_f32(const float* z, float* tok, int B, int C, int T, int H, int W, int t, int h, int w, hipStream_t st) {
    AVD_REQUIRE(z && tok && B > 0, AVD_EINVAL, "tube_patch: bad arguments");
    Tube g;
    if (int rc = make_tube(g, C, T, H, W, t, h, w)) return rc;
    const int64_t total4 = (int64_t)B * (g.per >> 2);
    static const int tag = prof_tag_id("tube_kernel<true>");
    ProfScope prof(tag, 8.0 * (double)B * g.per, st);
    hipLaunchKernelGGL(tube_kernel<true>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, z, tok, g, total4);
    AVD_CHECK_LAUNCH("tube_patch");
    return AVD_OK;
}

int tube_unpatch_f32(const float* tok, float* z, int B, int C, int T, int H, int W, int t, int h, int w,
                     hipStream_t st) {
    AVD_REQUIRE(z && tok && B > 0, AVD_EINVAL, "tube_unpatch: bad arguments");
    Tube g;
    if (int rc = make_tube(g, C, T, H, W, t, h, w)) return rc;
    const int64_t total4 = (int64_t)B * (g.per >> 2);
    hipLaunchKernelGGL(tube_kernel<false>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, tok, z, g, total4);
    AVD_CHECK_LAUNCH("tube_unpatch");
    return AVD_OK;
}

// ------------------------------------------------------------------ audio chunk tokens
__global__ void audio_tok_kernel(const float* __restrict__ z, float* __restrict__ tok, int B, int Ca, int F, int len,
                                 int stride, int Na) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int D = Ca * len;
    if (i >= (int64_t)B * Na * D) return;
    const int k = (int)(i % D);
    const int n = (int)((i / D) % Na);
    const int b = (int)(i / ((int64_t)D * Na));
    const int c = k / len, j = k % len;
    tok[i] = z[((int64_t)b * Ca + c) * F + n * stride + j];
}

// overlap-add value for frame f (f < L): windows summed in increasing window order, divided by the summed window weights
// (win == nullptr: rectangular window, i.e. the overlap count; otherwise ops.py:76-93 with apply_hann — multiply and add are
// rounded separately like the reference's `y += windows * win`)
__device__ __forceinline__ float ola_gather(const float* __restrict__ tokb, int D, int c, int len, int stride, int Na,
                                            int f, const float* __restrict__ win = nullptr) {
    int n_hi = f / stride;
    if (n_hi > Na - 1) n_hi = Na - 1;
    int n_lo = (f - len + stride) / stride;   // ceil((f-len+1)/stride)
    if (f - len + 1 <= 0) n_lo = 0;
    float acc = 0.f, cnt = 0.f;
    for (int n = n_lo; n <= n_hi; ++n) {
        const int j = f - n * stride;
        const float v = tokb[(int64_t)n * D + c * len + j];
        if (win) {
            acc = __fadd_rn(acc, __fmul_rn(v, win[j]));
            cnt += win[j];
        } else {
            acc += v;
            cnt += 1.f;
        }
    }
    return acc / fmaxf(cnt, 1e-8f);
}

__global__ void audio_untok_kernel(const float* __restrict__ tok, float* __restrict__ z, int B, int Ca, int F, int len,
                                   int stride, int Na, const float* __restrict__ win) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * Ca * F) return;
    const int f = (int)(i % F);
    const int c = (int)((i / F) % Ca);
    const int b = (int)(i / ((int64_t)F * Ca));
    const int L = (Na - 1) * stride + len;
    const int D = Ca * len;
    z[i] = f < L ? ola_gather(tok + (int64_t)b * Na * D, D, c, len, stride, Na, f, win) : 0.f;
}

static int audio_na(int F, int len, int stride) { return (F - len) / stride + 1; }

int audio_tokens_f32(const float* z, float* tok, int B, int Ca, int F, int len, int stride, hipStream_t st) {
    AVD_REQUIRE(z && tok && B > 0 && Ca > 0, AVD_EINVAL, "audio_tokens: bad arguments");
    AVD_REQUIRE(len > 0 && stride > 0 && F >= len, AVD_EUNSUPPORTED,
                "audio_tokens: need 0 < len <= F and stride > 0 (got F=%d len=%d stride=%d)", F, len, stride);
    const int Na = audio_na(F, len, stride);
    const int64_t n = (int64_t)B * Na * Ca * len;
    hipLaunchKernelGGL(audio_tok_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, tok, B, Ca, F, len,
                       stride, Na);
    AVD_CHECK_LAUNCH("audio_tokens");
    return AVD_OK;
}

int audio_untokens_f32(const float* tok, const float* win, float* z, int B, int Ca, int F, int len, int stride, hipStream_t st) {
    AVD_REQUIRE(z && tok && B > 0 && Ca > 0, AVD_EINVAL, "audio_untokens: bad arguments");
    AVD_REQUIRE(len > 0 && stride > 0 && F >= len, AVD_EUNSUPPORTED, "audio_untokens: bad chunking");
    const int Na = audio_na(F, len, stride);
    const int64_t n = (int64_t)B * Ca * F;
    hipLaunchKernelGGL(audio_untok_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, tok, z, B, Ca, F, len,
                       stride, Na, win);
    AVD_CHECK_LAUNCH("audio_untokens");
    return AVD_OK;
}

// ------------------------------------------------------------------ DDIM coefficients (per sample)
struct Ddim {
    float sqrt_omb_t, den, sqrt_a_prev, coeff_eps, sigma;
};

__device__ __forceinline__ Ddim ddim_coef(const int64_t* t_now, const int64_t* t_prev, const float* abar, int T_train,
                                          float eta, int b) {
    long long tn = t_now[b], tp = t_prev[b];
    if (tn < 0) tn = 0;
    if (tn > T_train - 1) tn = T_train - 1;           // the reference would raise IndexError; stay in bounds
    const float a_t = abar[tn];
    float a_p = 1.0f;                                   // abar_{-1} := 1
    if (tp >= 0) a_p = abar[tp > T_train - 1 ? T_train - 1 : tp];
    Ddim c;
    c.sqrt_omb_t = sqrtf(fmaxf(1.0f - a_t, 0.f));
    c.den = fmaxf(sqrtf(a_t), 1e-8f);
    c.sqrt_a_prev = sqrtf(a_p);
    c.sigma = 0.f;
    if (eta > 0.f) {
        const float frac = fmaxf((1.0f - a_p) / fmaxf(1.0f - a_t, 1e-8f), 0.f);
        const float omr = fmaxf(1.0f - a_t / fmaxf(a_p, 1e-8f), 0.f);
        c.sigma = eta * sqrtf(frac * omr);
    }
    c.coeff_eps = sqrtf(fmaxf(1.0f - a_p - c.sigma * c.sigma, 0.f));
    return c;
}

// No fp contraction in these two: the reference evaluates them as separate torch ops, one rounding each (schedule_utils.py:186-199,
// sample_clip.py:381), and hipcc's default -ffp-contract=fast picks its fused multiply-adds per CALL SITE — the same expression came out one
// ulp apart in two kernels of this file (round 5: the whole-line CFG kernel against the gather form; round 4 met the same in split8).
__device__ __forceinline__ float ddim_apply(const Ddim& c, float x, float e, float zn) {
#pragma clang fp contract(off)
    const float x0 = (x - c.sqrt_omb_t * e) / c.den;
    return c.sqrt_a_prev * x0 + c.coeff_eps * e + c.sigma * zn;
}
__device__ __forceinline__ float cfg_combine(float e_cond, float e_null, float guidance) {
#pragma clang fp contract(off)
    return e_null + guidance * (e_cond - e_null);
}

__global__ __launch_bounds__(256) void ddim_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                   const int64_t* __restrict__ t_now, const int64_t* __restrict__ t_prev,
                                                   const float* __restrict__ abar, int T_train, float eta,
                                                   const float* __restrict__ noise, float* __restrict__ out,
                                                   int64_t per, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / per);
    const Ddim c = ddim_coef(t_now, t_prev, abar, T_train, eta, b);
    out[i] = ddim_apply(c, x[i], eps[i], eta > 0.f ? noise[i] : 0.f);
}

int ddim_step_f32(const float* x_t, const float* eps, const int64_t* t_now, const int64_t* t_prev, const float* abar,
                  int T_train, float eta, const float* noise, float* x_prev, int B, int64_t per, hipStream_t st) {
    AVD_REQUIRE(x_t && eps && t_now && t_prev && abar && x_prev, AVD_EINVAL, "ddim_step: null pointer");
    AVD_REQUIRE(B > 0 && per > 0 && T_train > 0, AVD_EINVAL, "ddim_step: bad dims");
    AVD_REQUIRE(eta >= 0.f, AVD_EINVAL, "ddim_step: eta must be >= 0");
    AVD_REQUIRE(eta == 0.f || noise != nullptr, AVD_EINVAL, "ddim_step: eta > 0 needs a noise tensor");
    const int64_t total = (int64_t)B * per;
    hipLaunchKernelGGL(ddim_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x_t, eps, t_now, t_prev,
                       abar, T_train, eta, noise, x_prev, per, total);
    AVD_CHECK_LAUNCH("ddim_step");
    return AVD_OK;
}

// ------------------------------------------------------------------ fused CFG + unpatch + DDIM (video target)
int g_cfg_rows = getenv("AVD_CFG_ROWS") ? atoi(getenv("AVD_CFG_ROWS")) : 1;      // avd_tune_set "cfg_rows": 0 = the 16-bytes-per-lane gather form
__global__ __launch_bounds__(256) void cfg_unpatch_ddim_kernel(
    const float* __restrict__ eps2, const float* __restrict__ z, const int64_t* __restrict__ t_now,
    const int64_t* __restrict__ t_prev, const float* __restrict__ abar, int T_train, float guidance, float eta,
    const float* __restrict__ noise, float* __restrict__ z_out, Tube g, int B, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int64_t per4 = g.per >> 2;
    const int b = (int)(i / per4);
    const int64_t e4 = i % per4;
    const int64_t lat = (int64_t)b * g.per + e4 * 4;
    const int64_t toff = tube_tok_off(g, e4);
    const f32x4 ec = *reinterpret_cast<const f32x4*>(eps2 + (int64_t)b * g.per + toff);
    const f32x4 en = *reinterpret_cast<const f32x4*>(eps2 + ((int64_t)B + b) * g.per + toff);
    const f32x4 x = *reinterpret_cast<const f32x4*>(z + lat);
    f32x4 zn = {0.f, 0.f, 0.f, 0.f};
    if (eta > 0.f) zn = *reinterpret_cast<const f32x4*>(noise + lat);
    const Ddim c = ddim_coef(t_now, t_prev, abar, T_train, eta, b);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float e = cfg_combine(ec[k], en[k], guidance);
        o[k] = ddim_apply(c, x[k], e, zn[k]);
    }
    *reinterpret_cast<f32x4*>(z_out + lat) = o;
}

// Coalesced form (round 5).  In the kernel above a lane's 16 bytes of latent (4 consecutive w) are one 16-byte piece of a token row, and
// the NEXT 16 bytes of latent belong to the next token, 1 KiB further on: every wave-wide token read touches 64 cache lines for 16 bytes
// each (FETCH_SIZE 63 MB per launch at C3 against 37.7 MB algorithmic).  Here a block owns GT tokens that are neighbours along w' — GT x w
// = one 128-byte line of latent per (c, t, h) — reads their cond / null rows as whole contiguous rows (GT x D floats each, 16 B per lane),
// combines (CFG) in registers, parks the combined eps in LDS as [token][feature] and reads it back as [(c, t, h)][GT x w]: eight lanes then
// cover one full 128-byte line of z / z_out.  Same arithmetic per element, in the same order: bit-identical to the kernel above.
template <int GT>      // tokens per block
__global__ __launch_bounds__(256) void cfg_unpatch_ddim_rows_kernel(
    const float* __restrict__ eps2, const float* __restrict__ z, const int64_t* __restrict__ t_now,
    const int64_t* __restrict__ t_prev, const float* __restrict__ abar, int T_train, float guidance, float eta,
    const float* __restrict__ noise, float* __restrict__ z_out, Tube g, int B, int groups_per_sample) {
    extern __shared__ __attribute__((aligned(16))) float ebuf[];       // [GT][D + 4]: the pad keeps the transposed 16-byte reads off one bank group
    const int LD = g.D + 4;
    const int b = blockIdx.x / groups_per_sample, grp = blockIdx.x % groups_per_sample;
    const int n0 = grp * GT;                                            // first token of the group (GT divides W / w: one (t', h') row)
    const float* tc = eps2 + ((int64_t)b * (g.per / g.D) + n0) * g.D;
    const float* tn = eps2 + (((int64_t)B + b) * (g.per / g.D) + n0) * g.D;
    const int nf4 = GT * g.D / 4;
    for (int i = threadIdx.x; i < nf4; i += 256) {
        const f32x4 ec = *reinterpret_cast<const f32x4*>(tc + (int64_t)i * 4), en = *reinterpret_cast<const f32x4*>(tn + (int64_t)i * 4);
        f32x4 e;
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = cfg_combine(ec[k], en[k], guidance);
        const int tok = (i * 4) / g.D, k0 = (i * 4) % g.D;
        *reinterpret_cast<f32x4*>(ebuf + tok * LD + k0) = e;
    }
    __syncthreads();
    const Ddim c = ddim_coef(t_now, t_prev, abar, T_train, eta, b);
    // token coordinates of the group: n = (t' Ht + h') Wt + w'
    const int wq = n0 % g.Wt, hq = (n0 / g.Wt) % g.Ht, tq = n0 / (g.Wt * g.Ht);
    const int segs = g.D / g.w;                                          // (c, t, h) combinations of a token
    const int per_seg4 = GT * g.w / 4;                                   // float4s of one latent line piece
    for (int i = threadIdx.x; i < segs * per_seg4; i += 256) {
        const int sg = i / per_seg4, j = i % per_seg4;                   // piece j of line sg: token j * 4 / w, offset (j * 4) % w
        const int tok = (j * 4) / g.w, wo = (j * 4) % g.w;
        const int hh = sg % g.h, tt = (sg / g.h) % g.t, cc = sg / (g.h * g.t);
        const f32x4 e = *reinterpret_cast<const f32x4*>(ebuf + tok * LD + sg * g.w + wo);
        const int64_t lat = (int64_t)b * g.per + (((int64_t)cc * g.T + (tq * g.t + tt)) * g.H + (hq * g.h + hh)) * g.W + (wq + tok) * g.w + wo;
        const f32x4 x = *reinterpret_cast<const f32x4*>(z + lat);
        f32x4 zn = {0.f, 0.f, 0.f, 0.f};
        if (eta > 0.f) zn = *reinterpret_cast<const f32x4*>(noise + lat);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ddim_apply(c, x[k], e[k], zn[k]);
        *reinterpret_cast<f32x4*>(z_out + lat) = o;
    }
}

int cfg_unpatch_ddim_f32(const float* eps2, const float* z, const int64_t* t_now, const int64_t* t_prev,
                         const float* abar, int T_train, float guidance, float eta, const float* noise, float* z_out,
                         int B, int C, int T, int H, int W, int t, int h, int w, hipStream_t st) {
    AVD_REQUIRE(eps2 && z && t_now && t_prev && abar && z_out, AVD_EINVAL, "cfg_unpatch_ddim: null pointer");
    AVD_REQUIRE(B > 0 && T_train > 0, AVD_EINVAL, "cfg_unpatch_ddim: bad dims");
    AVD_REQUIRE(eta >= 0.f && (eta == 0.f || noise), AVD_EINVAL, "cfg_unpatch_ddim: eta > 0 needs a noise tensor");
    AVD_REQUIRE(z != z_out, AVD_EINVAL, "cfg_unpatch_ddim: z_out must not alias z");
    Tube g;
    if (int rc = make_tube(g, C, T, H, W, t, h, w)) return rc;
    const int64_t total4 = (int64_t)B * (g.per >> 2);
    static const int tag = prof_tag_id("cfg_unpatch_ddim_kernel");
    ProfScope prof(tag, 16.0 * (double)B * g.per, st);
    // whole-line form: groups of tokens along w' that make up 128 bytes (or the whole row when W is shorter) of latent per (c, t, h)
    const int gt = (g.W < 32 ? g.W : 32) / g.w;
    if (g_cfg_rows && (gt == 8 || gt == 4) && g.Wt % gt == 0 && g.D % 4 == 0 && (int64_t)gt * (g.D + 4) * 4 <= 64 * 1024) {
        const int groups = (int)(g.per / g.D) / gt;
        const size_t lds = (size_t)gt * (g.D + 4) * 4;
        if (gt == 8)
            hipLaunchKernelGGL(cfg_unpatch_ddim_rows_kernel<8>, dim3((unsigned)(B * groups)), dim3(256), lds, st, eps2, z, t_now, t_prev, abar,
                               T_train, guidance, eta, noise, z_out, g, B, groups);
        else
            hipLaunchKernelGGL(cfg_unpatch_ddim_rows_kernel<4>, dim3((unsigned)(B * groups)), dim3(256), lds, st, eps2, z, t_now, t_prev, abar,
                               T_train, guidance, eta, noise, z_out, g, B, groups);
        AVD_CHECK_LAUNCH("cfg_unpatch_ddim (rows)");
        return AVD_OK;
    }
    hipLaunchKernelGGL(cfg_unpatch_ddim_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, eps2, z, t_now,
                       t_prev, abar, T_train, guidance, eta, noise, z_out, g, B, total4);
    AVD_CHECK_LAUNCH("cfg_unpatch_ddim");
    return AVD_OK;
}

// ------------------------------------------------------------------ fused CFG + overlap-add + DDIM (audio target)
__global__ void cfg_untoken_ddim_audio_kernel(const float* __restrict__ eps2, const float* __restrict__ z,
                                              const int64_t* __restrict__ t_now, const int64_t* __restrict__ t_prev,
                                              const float* __restrict__ abar, int T_train, float guidance, float eta,
                                              const float* __restrict__ noise, float* __restrict__ z_out, int B, int Ca,
                                              int F, int len, int stride, int Na) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * Ca * F) return;
    const int f = (int)(i % F);
    const int c = (int)((i / F) % Ca);
    const int b = (int)(i / ((int64_t)F * Ca));
    const int L = (Na - 1) * stride + len;
    const int D = Ca * len;
    float e = 0.f;
    if (f < L) {
        // CFG combine is linear, but the reference combines per token first and overlap-adds after:
        // gather both halves with the same window order, combine per window
        int n_hi = f / stride;
        if (n_hi > Na - 1) n_hi = Na - 1;
        int n_lo = (f - len + 1 <= 0) ? 0 : (f - len + stride) / stride;
        float acc = 0.f, cnt = 0.f;
        const float* tc = eps2 + (int64_t)b * Na * D;
        const float* tn = eps2 + ((int64_t)B + b) * Na * D;
        for (int n = n_lo; n <= n_hi; ++n) {
            const int64_t o = (int64_t)n * D + c * len + (f - n * stride);
            const float vn = tn[o];
            acc += cfg_combine(tc[o], vn, guidance);
            cnt += 1.f;
        }
        e = acc / fmaxf(cnt, 1e-8f);
    }
    const Ddim cf = ddim_coef(t_now, t_prev, abar, T_train, eta, b);
    z_out[i] = ddim_apply(cf, z[i], e, eta > 0.f ? noise[i] : 0.f);
}

int cfg_untoken_ddim_audio_f32(const float* eps2, const float* z, const int64_t* t_now, const int64_t* t_prev,
                               const float* abar, int T_train, float guidance, float eta, const float* noise,
                               float* z_out, int B, int Ca, int F, int len, int stride, hipStream_t st) {
    AVD_REQUIRE(eps2 && z && t_now && t_prev && abar && z_out, AVD_EINVAL, "cfg_untoken_ddim_audio: null pointer");
    AVD_REQUIRE(B > 0 && Ca > 0 && T_train > 0, AVD_EINVAL, "cfg_untoken_ddim_audio: bad dims");
    AVD_REQUIRE(len > 0 && stride > 0 && F >= len, AVD_EUNSUPPORTED, "cfg_untoken_ddim_audio: bad chunking");
    AVD_REQUIRE(eta >= 0.f && (eta == 0.f || noise), AVD_EINVAL, "cfg_untoken_ddim_audio: eta > 0 needs noise");
    AVD_REQUIRE(z != z_out, AVD_EINVAL, "cfg_untoken_ddim_audio: z_out must not alias z");
    const int Na = audio_na(F, len, stride);
    const int64_t n = (int64_t)B * Ca * F;
    hipLaunchKernelGGL(cfg_untoken_ddim_audio_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, eps2, z, t_now,
                       t_prev, abar, T_train, guidance, eta, noise, z_out, B, Ca, F, len, stride, Na);
    AVD_CHECK_LAUNCH("cfg_untoken_ddim_audio");
    return AVD_OK;
}

// ------------------------------------------------------------------ CFG-stacked sequence assembly
// X2[2B, N, d]: the GEMM has already written adapter(target tokens) into the cond half's target rows,
// columns [0, d-tdim).  This pass fills everything else in one sweep.
__global__ __launch_bounds__(256) void assemble_kernel(float* __restrict__ X2, const float* __restrict__ temb,
                                                       const float* __restrict__ Xp, int B, int N, int d, int tdim,
                                                       int Nt, int Np, int target_first, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int d4 = d >> 2;
    const int col = (int)(i % d4) * 4;
    const int64_t row = i / d4;
    const int n = (int)(row % N);
    const int bb = (int)(row / N);          // 0..2B-1
    const int half = bb >= B, b = half ? bb - B : bb;
    const int t0 = target_first ? 0 : Np;   // first target row
    const bool is_t = n >= t0 && n < t0 + Nt;
    f32x4 v;
    if (is_t) {
        if (col >= d - tdim) {
            v = *reinterpret_cast<const f32x4*>(temb + (int64_t)b * tdim + (col - (d - tdim)));
        } else {
            if (!half) return;               // already in place
            v = *reinterpret_cast<const f32x4*>(X2 + ((int64_t)b * N + n) * d + col);
        }
    } else {
        const int np = target_first ? n - Nt : n;
        v = half ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(Xp + ((int64_t)b * Np + np) * d + col);
    }
    *reinterpret_cast<f32x4*>(X2 + row * d + col) = v;
}

int assemble_f32(float* X2, const float* temb, const float* Xp, int B, int N, int d, int tdim, int Nt, int Np,
                 int target_first, hipStream_t st) {
    const int64_t total4 = (int64_t)2 * B * N * (d >> 2);
    static const int tag = prof_tag_id("assemble_kernel");
    ProfScope prof(tag, 4.0 * (2.0 * B * N * d + (double)B * Nt * (d - tdim) + (double)B * Np * d), st);
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, X2, temb, Xp, B, N, d,
                       tdim, Nt, Np, target_first, total4);
    AVD_CHECK_LAUNCH("assemble");
    return AVD_OK;
}

// Fused form for the sampler's concat embedding (sample_clip.py:59-70, 371, 377): one wave per row of X2 fills everything the
// adapter GEMM did not write — the sinusoidal timestep columns are computed in place (no [B, tdim] buffer, no separate
// kernel), the null half's target rows are copied from the cond half, prompt rows come from Xp / zeros — and, while the row
// is in registers, its sum of squares goes to ss[row] (the table the first folded RMSNorm reads: no rowss pass).
__global__ __launch_bounds__(256) void assemble_rows_kernel(float* __restrict__ X2, const int64_t* __restrict__ t_now,
                                                            const float* __restrict__ freqs, const float* __restrict__ Xp,
                                                            float* __restrict__ ss, int B, int N, int d, int tdim, int Nt, int Np,
                                                            int target_first, float neg_log_mp) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)2 * B * N) return;
    const int n = (int)(row % N);
    const int bb = (int)(row / N);
    const int half = bb >= B, b = half ? bb - B : bb;
    const int t0 = target_first ? 0 : Np;
    const bool is_t = n >= t0 && n < t0 + Nt;
    const int da = d - tdim, th = tdim >> 1;
    const float tf = is_t ? (float)t_now[b] : 0.f;
    float acc = 0.f;
    for (int col = lane * 4; col < d; col += 256) {
        f32x4 v;
        bool store = true;
        if (is_t) {
            if (col >= da) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = col - da + e;
                    float x = 0.f;
                    if (c < 2 * th) {
                        const int k = c < th ? c : c - th;
                        const float f = freqs ? freqs[k] : expf(neg_log_mp * (float)k / (float)th);
                        const float a = tf * f;
                        x = c < th ? cosf(a) : sinf(a);
                    }
                    v[e] = x;
                }
            } else {
                v = *reinterpret_cast<const f32x4*>(X2 + ((int64_t)b * N + n) * d + col);      // the adapter GEMM's output (cond half)
                store = half != 0;
            }
        } else {
            const int np = target_first ? n - Nt : n;
            v = half ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(Xp + ((int64_t)b * Np + np) * d + col);
        }
        if (store) *reinterpret_cast<f32x4*>(X2 + row * d + col) = v;
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    acc = wave_sum(acc);
    if (ss && lane == 0) ss[row] = acc;
}

int assemble_rows_f32(float* X2, const int64_t* t_now, const float* freqs, const float* Xp, float* ss, int B, int N, int d, int tdim,
                      int Nt, int Np, int target_first, float max_period, hipStream_t st) {
    const int64_t rows = (int64_t)2 * B * N;
    static const int tag = prof_tag_id("assemble_rows_kernel");
    ProfScope prof(tag, 4.0 * (2.0 * B * N * d + (double)B * Nt * (d - tdim) + (double)B * Np * d), st);
    hipLaunchKernelGGL(assemble_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, X2, t_now, freqs, Xp, ss, B, N, d, tdim,
                       Nt, Np, target_first, -(float)log((double)max_period));
    AVD_CHECK_LAUNCH("assemble_rows");
    return AVD_OK;
}

// ------------------------------------------------------------------ device-side schedule cursor
__global__ void sched_advance_kernel(const int64_t* __restrict__ sched, int n_sched, int32_t* cursor,
                                     int64_t* __restrict__ t_now, int64_t* __restrict__ t_prev, int B) {
    __shared__ int cur;
    if (threadIdx.x == 0) cur = *cursor;
    __syncthreads();
    int i = cur;
    if (i < 0) i = 0;
    if (i > n_sched - 2) i = n_sched - 2;
    const int64_t a = sched[i], p = sched[i + 1];
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        t_now[b] = a;
        t_prev[b] = p;
    }
    __syncthreads();
    if (threadIdx.x == 0) *cursor = cur + 1;
}

}  // namespace avd

using namespace avd;

extern "C" int avd_timestep_embedding_f32(const int64_t* t, const float* freqs, float* out, int B, int dim,
                                          float max_period, avd_stream_t stream) {
    return temb_f32(t, freqs, out, B, dim, max_period, static_cast<hipStream_t>(stream));
}
extern "C" int avd_tube_patch_f32(const float* z, float* tok, int B, int C, int T, int H, int W, int t, int h, int w,
                                  avd_stream_t stream) {
    AVD_REQUIRE(aligned16(z) && aligned16(tok), AVD_EUNSUPPORTED, "tube_patch: pointers must be 16-byte aligned");
    return tube_patch_f32(z, tok, B, C, T, H, W, t, h, w, static_cast<hipStream_t>(stream));
}
extern "C" int avd_tube_unpatch_f32(const float* tok, float* z, int B, int C, int T, int H, int W, int t, int h, int w,
                                    avd_stream_t stream) {
    AVD_REQUIRE(aligned16(z) && aligned16(tok), AVD_EUNSUPPORTED, "tube_unpatch: pointers must be 16-byte aligned");
    return tube_unpatch_f32(tok, z, B, C, T, H, W, t, h, w, static_cast<hipStream_t>(stream));
}
extern "C" int avd_audio_tokens_f32(const float* z, float* tok, int B, int Ca, int F, int len, int stride,
                                    avd_stream_t stream) {
    return audio_tokens_f32(z, tok, B, Ca, F, len, stride, static_cast<hipStream_t>(stream));
}
extern "C" int avd_audio_untokens_f32(const float* tok, const float* window, float* z, int B, int Ca, int F, int len, int stride,
                                      avd_stream_t stream) {
    return audio_untokens_f32(tok, window, z, B, Ca, F, len, stride, static_cast<hipStream_t>(stream));
}
extern "C" int avd_ddim_step_f32(const float* x_t, const float* eps_hat, const int64_t* t_now, const int64_t* t_prev,
                                 const float* alpha_bar, int T_train, float eta, const float* noise, float* x_prev,
                                 int B, int64_t per_sample, avd_stream_t stream) {
    return ddim_step_f32(x_t, eps_hat, t_now, t_prev, alpha_bar, T_train, eta, noise, x_prev, B, per_sample,
                         static_cast<hipStream_t>(stream));
}
extern "C" int avd_cfg_unpatch_ddim_f32(const float* eps2, const float* z, const int64_t* t_now, const int64_t* t_prev,
                                        const float* alpha_bar, int T_train, float guidance, float eta,
                                        const float* noise, float* z_out, int B, int C, int T, int H, int W, int t,
                                        int h, int w, avd_stream_t stream) {
    AVD_REQUIRE(aligned16(eps2) && aligned16(z) && aligned16(z_out) && (!noise || aligned16(noise)), AVD_EUNSUPPORTED,
                "cfg_unpatch_ddim: pointers must be 16-byte aligned");
    return cfg_unpatch_ddim_f32(eps2, z, t_now, t_prev, alpha_bar, T_train, guidance, eta, noise, z_out, B, C, T, H, W, t,
                                h, w, static_cast<hipStream_t>(stream));
}
extern "C" int avd_cfg_untoken_ddim_audio_f32(const float* eps2, const float* z, const int64_t* t_now,
                                              const int64_t* t_prev, const float* alpha_bar, int T_train, float guidance,
                                              float eta, const float* noise, float* z_out, int B, int Ca, int F, int len,
                                              int stride, avd_stream_t stream) {
    return cfg_untoken_ddim_audio_f32(eps2, z, t_now, t_prev, alpha_bar, T_train, guidance, eta, noise, z_out, B, Ca, F,
                                      len, stride, static_cast<hipStream_t>(stream));
}
extern "C" int avd_sched_advance(const int64_t* sched, int n_sched, int32_t* cursor, int64_t* t_now, int64_t* t_prev,
                                 int B, avd_stream_t stream) {
    AVD_REQUIRE(sched && cursor && t_now && t_prev && n_sched >= 2 && B > 0, AVD_EINVAL, "sched_advance: bad arguments");
    hipLaunchKernelGGL(sched_advance_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), sched, n_sched,
                       cursor, t_now, t_prev, B);
    AVD_CHECK_LAUNCH("sched_advance");
    return AVD_OK;
}
