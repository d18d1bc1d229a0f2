// fp8 (OCP e4m3) attention — the reduced-precision variant BASELINE config C5 names ("512x512 latent-space diffusion, fp8 MFMA
// attention").  Same reference op as attn_f32.hip / attn_bf16x3.hip (nn.MultiheadAttention's scaled-dot-product step,
// avdiff/models/mmdt.py:51-61), but BOTH contractions take e4m3 operands on v_mfma_f32_32x32x16_fp8_fp8 with fp32 accumulation.
// The reference has no reduced-precision inference (infer/sample_clip.py:399-411 never reads mixed_precision), so this is never a
// parity path and never the default: its error against the fp32 result is REPORTED (tests/test_gpu_parity.py).
//
// Two kernels:
//  quant_fp8_kernel   reads the qkv3 image the in_proj epilogue wrote (three bf16 planes per value), rebuilds the fp32 value
//                     (h + m + l, exact), rounds it ONCE to e4m3 and lays it out for the attention kernel:
//                       Q8 [b][h][Npad][64 d]                      (q already carries softmax scale * log2 e; an extra x8 keeps
//                                                                   small components in e4m3's normal range)
//                       K8 [b][h][Npad/64 tiles][64 keys][64 d]     8-byte d chunks XOR-swizzled with (key >> 2) & 7
//                       V8 [b][h][Npad/64 tiles][64 d][64 keys]     TRANSPOSED through LDS, keys inside every 16-group in the order
//                                                                   the score accumulators hold them (bits 2 and 3 swapped), 8-byte key
//                                                                   chunks swizzled with (d >> 2) & 7
//                     so that a K / V tile is one contiguous, conflict-free 4 KiB LDS image (linear LDS-DMA copy) and every MFMA
//                     operand is a single ds_read_b64.
//  attn_fp8_kernel    flash-style loop of attn_bf16x3.hip with one plane: S^T = K Q^T (8 MFMAs per 64-key tile), online softmax in
//                     registers in the exp2 domain, P scaled by 256 before its e4m3 rounding (a uniform row over 1,573 keys is
//                     6e-4 per key — below e4m3's smallest subnormal unscaled), O^T = V^T P^T (8 MFMAs), fp32 normalisation.
#include "avd_common.h"

namespace avd {

constexpr int F8_DH = 64, F8_KT = 64, F8_NW = 4;
constexpr float F8_NEG = -1.0e30f;
constexpr float F8_QSCALE = 8.0f, F8_PSCALE = 256.0f;

#define AVD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AVD_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

typedef long long i64;

// 8 floats -> 8 e4m3 bytes (round to nearest even, saturating), element e in byte e
__device__ __forceinline__ i64 pack_fp8x8(const float* v) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return (i64)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// position of key offset k (0..15) inside its 16-key group of the V image: bits 2 and 3 swapped, so that the 8 keys a lane half
// multiplies in one MFMA (4 hi + (j & 3) + 8 (j >> 2), j = 0..7) are the 8 contiguous bytes at 8 hi
__device__ __forceinline__ int f8_vpos(int k) { return (k & 3) | ((k & 4) << 1) | ((k & 8) >> 1); }

// one block per (part, sample, head, 64-row tile) of the qkv3 image.  F16: the image holds two fp16 planes at scale 1 / inv_s (f16x2)
template <bool F16>
__global__ __launch_bounds__(256) void quant_fp8_kernel(const unsigned char* __restrict__ img, unsigned char* __restrict__ q8,
                                                        unsigned char* __restrict__ k8, unsigned char* __restrict__ v8, int Bt, int N,
                                                        int Npad, int H, float inv_s) {
    __shared__ __attribute__((aligned(16))) unsigned char vt[64 * 64];
    const int ntile = Npad / 64;
    int w = blockIdx.x;
    const int tile = w % ntile; w /= ntile;
    const int h = w % H; w /= H;
    const int b = w % Bt;
    const int part = w / Bt;
    const int tid = threadIdx.x;
    const int row = tid >> 2, c0 = (tid & 3) * 2;          // row of the tile, first of two 8-d chunks
    const int n = tile * 64 + row;
    const unsigned char* src = img + ((((int64_t)part * Bt + b) * H + h) * (int64_t)Npad + (n < N ? n : N - 1)) * QKV3_ROWB;
    i64 packed[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
        const int c = c0 + cc;
        const int slot = (c ^ qkv3_swizzle(part, n < N ? n : N - 1)) << 4;
        float v[8];
        const u32x4 ph = *reinterpret_cast<const u32x4*>(src + slot), pm = *reinterpret_cast<const u32x4*>(src + 128 + slot);
        if constexpr (F16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const f32x2 a = unpk_f16(ph[e]), c = unpk_f16(pm[e]);
                v[2 * e] = (a[0] + c[0]) * inv_s;
                v[2 * e + 1] = (a[1] + c[1]) * inv_s;
            }
        } else {
            const u32x4 pl = *reinterpret_cast<const u32x4*>(src + 256 + slot);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = (bf16_lo(ph[e]) + bf16_lo(pm[e])) + bf16_lo(pl[e]);
                v[2 * e + 1] = (bf16_hi(ph[e]) + bf16_hi(pm[e])) + bf16_hi(pl[e]);
            }
        }
        if (n >= N) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;          // padding rows: finite zeros (their scores are masked)
        }
        if (part == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= F8_QSCALE;
        }
        packed[cc] = pack_fp8x8(v);
    }
    const int64_t bh = (int64_t)b * H + h;
    if (part == 0) {
        unsigned char* dst = q8 + (bh * Npad + n) * 64 + c0 * 8;
        *reinterpret_cast<i64*>(dst) = packed[0];
        *reinterpret_cast<i64*>(dst + 8) = packed[1];
    } else if (part == 1) {
        unsigned char* dst = k8 + (bh * ntile + tile) * 4096 + row * 64;
        const int sw = (row >> 2) & 7;
        *reinterpret_cast<i64*>(dst + (((c0) ^ sw) << 3)) = packed[0];
        *reinterpret_cast<i64*>(dst + (((c0 + 1) ^ sw) << 3)) = packed[1];
    } else {
        // transpose through LDS: byte (d, position of this key)
        const int kpos = (row & ~15) | f8_vpos(row & 15);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e) vt[((c0 + cc) * 8 + e) * 64 + kpos] = (unsigned char)(packed[cc] >> (8 * e));
        __syncthreads();
        const int d = tid >> 2, seg = (tid & 3) * 2;      // d row, first of two 8-key chunks
        unsigned char* dst = v8 + (bh * ntile + tile) * 4096 + d * 64;
        const int sw = (d >> 2) & 7;
        *reinterpret_cast<i64*>(dst + (((seg) ^ sw) << 3)) = *reinterpret_cast<const i64*>(vt + d * 64 + seg * 8);
        *reinterpret_cast<i64*>(dst + (((seg + 1) ^ sw) << 3)) = *reinterpret_cast<const i64*>(vt + d * 64 + seg * 8 + 8);
    }
}

template <int SPLIT_OUT>      // 0: fp32 out; 1: bf16x3 (split3) image; 2: f16x2 image at scale o_scale
__global__ __launch_bounds__(F8_NW * 64, 2) void attn_fp8_kernel(const unsigned char* __restrict__ q8, const unsigned char* __restrict__ k8,
                                                                 const unsigned char* __restrict__ v8, float* __restrict__ out, int N,
                                                                 int Npad, int H, int n_query, int nqb, float o_scale) {
    constexpr int NW = F8_NW;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[4096];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[4096];

    int qb, h, b;
    {
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        const int w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
        qb = w % nqb;
        h = (w / nqb) % H;
        b = w / (nqb * H);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int ntile = Npad / 64;
    const int64_t bh = (int64_t)b * H + h;
    const unsigned char* Kb = k8 + bh * ntile * 4096;
    const unsigned char* Vb = v8 + bh * ntile * 4096;

    // Q fragments: lane (query column l31, half hi), d step s: the 8 bytes d = 16 s + 8 hi .. + 7
    const int q_row = qb * (NW * 32) + wave * 32 + l31;
    i64 qf[4];
    {
        const unsigned char* src = q8 + (bh * Npad + (q_row < N ? q_row : N - 1)) * 64 + hi * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const i64*>(src + s * 16);
    }
    auto dma = [&](const unsigned char* gsrc, unsigned char* ldst, int kt) {
        __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(gsrc + (int64_t)kt * 4096 + wave * 1024 + lane * 16), AVD_LDS_PTR(ldst + wave * 1024), 16, 0, 0);
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = F8_NEG, l_run = 0.f;

    const int nkt = (N + F8_KT - 1) / F8_KT;
    dma(Kb, Ks, 0);
    dma(Vb, Vs, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();

    const int ksw = (l31 >> 2) & 7;       // same for key rows l31 and 32 + l31, and for d rows l31 and 32 + l31
    const bool active = qb * (NW * 32) + wave * 32 < n_query;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (!active) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (more) dma(Kb, Ks, kt + 1);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (more) dma(Vb, Vs, kt + 1);
            continue;
        }
        // ---- S^T = K Q^T for keys [0,32) and [32,64) of the tile ----
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int off = ((2 * s + hi) ^ ksw) << 3;
            const i64 ka = *reinterpret_cast<const i64*>(Ks + l31 * 64 + off);
            const i64 kb2 = *reinterpret_cast<const i64*>(Ks + (32 + l31) * 64 + off);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(ka, qf[s], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(kb2, qf[s], s1, 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma(Kb, Ks, kt + 1);

        if (!more && (N & (F8_KT - 1))) {
            const int kbase = kt * F8_KT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + mfma32_row(r, hi);
                if (key >= N) s0[r] = F8_NEG;
                if (key + 32 >= N) s1[r] = F8_NEG;
            }
        }
        // ---- online softmax (scores carry the extra Q scale: undo it on the way into exp2) ----
        constexpr float inv_q = 1.0f / F8_QSCALE;
        float mt = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s1[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * inv_q;
        const float m_new = fmaxf(m_run, mt);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], inv_q, -m_new));
            s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], inv_q, -m_new));
            ps += s0[r] + s1[r];
        }
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        l_run += ps;
        m_run = m_new;

        // ---- O^T += V^T P^T: k-step (kb, t) covers keys 32 kb + 16 t + 4 hi + (j & 3) + 8 (j >> 2) — registers 8t..8t+7 of S^T ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = (kb ? s1[8 * t + j] : s0[8 * t + j]) * F8_PSCALE;
                const i64 pf = pack_fp8x8(pv);
                const int off = ((4 * kb + 2 * t + hi) ^ ksw) << 3;
                const i64 va = *reinterpret_cast<const i64*>(Vs + l31 * 64 + off);
                const i64 vb = *reinterpret_cast<const i64*>(Vs + (32 + l31) * 64 + off);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(va, pf, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(vb, pf, o1, 0, 0, 0);
            }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma(Vb, Vs, kt + 1);
    }
    if (!active) return;

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / (l_tot * F8_PSCALE);
    const int d = H * F8_DH;
    if constexpr (SPLIT_OUT != 0) {
        float ch[8][4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ch[g4][e] = o0[4 * g4 + e] * inv;
                ch[4 + g4][e] = o1[4 * g4 + e] * inv;
            }
        unsigned char* o3 = reinterpret_cast<unsigned char*>(out);
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float send = hi ? ch[c][e] : ch[c + 1][e];
                const float recv = __shfl_xor(send, 32, 64);
                v[e] = hi ? recv : ch[c][e];
                v[4 + e] = hi ? ch[c + 1][e] : recv;
            }
            if (q_row < n_query) {
                if constexpr (SPLIT_OUT == 2) store_split8_h2(o3, (int64_t)b * N + q_row, h * F8_DH + 8 * (c + hi), d, v, o_scale);
                else store_split8(o3, (int64_t)b * N + q_row, h * F8_DH + 8 * (c + hi), d, v);
            }
        }
    } else if (q_row < n_query) {
        float* dst = out + ((int64_t)b * N + q_row) * d + h * F8_DH + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<f32x4*>(dst + 8 * g4) = a;
            *reinterpret_cast<f32x4*>(dst + 32 + 8 * g4) = c;
        }
    }
}

int64_t attn_fp8_ws_bytes(int B, int N, int H) { return (int64_t)3 * B * H * qkv3_npad(N) * 64; }

// qkv3: the image avd_gemm_bf16x3_qkv3_f32 (img_terms != 3) or avd_gemm_f16x2_qkv_f32 (img_terms 3, at scale img_scale) wrote;
// ws: attn_fp8_ws_bytes scratch; out3 != null: the result as an operand image of the same kind (f16x2: at scale out_scale)
int attn_fp8(const void* qkv3, void* ws, int64_t ws_bytes, float* out, void* out3, int B, int N, int H, int n_query, hipStream_t st,
             int img_terms, float img_scale, float out_scale) {
    AVD_REQUIRE(qkv3 && ws && (out || out3), AVD_EINVAL, "attn_fp8: null pointer");
    AVD_REQUIRE(img_scale > 0.f && img_scale < __builtin_inff() && out_scale > 0.f && out_scale < __builtin_inff(), AVD_EINVAL,
                "attn_fp8: image scales must be positive and finite");
    const bool h2 = img_terms == 3;
    AVD_REQUIRE(B > 0 && N > 0 && H > 0, AVD_EINVAL, "attn_fp8: bad dims B=%d N=%d H=%d", B, N, H);
    AVD_REQUIRE(n_query >= 0 && n_query <= N, AVD_EINVAL, "attn_fp8: n_query=%d outside [0,%d]", n_query, N);
    AVD_REQUIRE(ws_bytes >= attn_fp8_ws_bytes(B, N, H), AVD_EWORKSPACE, "attn_fp8: workspace %lld < %lld bytes", (long long)ws_bytes,
                (long long)attn_fp8_ws_bytes(B, N, H));
    AVD_REQUIRE(aligned16(qkv3) && aligned16(ws) && aligned16(out) && aligned16(out3), AVD_EUNSUPPORTED, "attn_fp8: pointers must be 16-byte aligned");
    if (n_query == 0) return AVD_OK;
    const int Npad = qkv3_npad(N), ntile = Npad / 64;
    const int64_t per = (int64_t)B * H * Npad * 64;
    AVD_REQUIRE((int64_t)3 * B * H * ntile < (1ll << 31), AVD_EUNSUPPORTED, "attn_fp8: grid too large");
    unsigned char* q8 = static_cast<unsigned char*>(ws);
    unsigned char* k8 = q8 + per;
    unsigned char* v8 = k8 + per;
    {
        static const int tag = prof_tag_id("quant_fp8_kernel");
        ProfScope prof(tag, (double)per * 3 * 7.0, st);
        if (h2)
            hipLaunchKernelGGL(quant_fp8_kernel<true>, dim3((unsigned)(3 * B * H * ntile)), dim3(256), 0, st, static_cast<const unsigned char*>(qkv3),
                               q8, k8, v8, B, N, Npad, H, 1.0f / img_scale);
        else
            hipLaunchKernelGGL(quant_fp8_kernel<false>, dim3((unsigned)(3 * B * H * ntile)), dim3(256), 0, st, static_cast<const unsigned char*>(qkv3),
                               q8, k8, v8, B, N, Npad, H, 1.0f);
        AVD_CHECK_LAUNCH("quant_fp8");
    }
    const int nqb = (n_query + 32 * F8_NW - 1) / (32 * F8_NW);
    static const int tag = prof_tag_id("attn_fp8_kernel");
    ProfScope prof(tag, 4.0 * (double)B * H * (double)n_query * N * F8_DH, st);
    if (out3 && h2)
        hipLaunchKernelGGL(attn_fp8_kernel<2>, dim3(nqb * H * B), dim3(F8_NW * 64), 0, st, q8, k8, v8, static_cast<float*>(out3), N, Npad, H,
                           n_query, nqb, out_scale);
    else if (out3)
        hipLaunchKernelGGL(attn_fp8_kernel<1>, dim3(nqb * H * B), dim3(F8_NW * 64), 0, st, q8, k8, v8, static_cast<float*>(out3), N, Npad, H,
                           n_query, nqb, 1.f);
    else
        hipLaunchKernelGGL(attn_fp8_kernel<0>, dim3(nqb * H * B), dim3(F8_NW * 64), 0, st, q8, k8, v8, out, N, Npad, H, n_query, nqb, 1.f);
    AVD_CHECK_LAUNCH("attn_fp8");
    return AVD_OK;
}

}  // namespace avd

extern "C" int64_t avd_attn_fp8_workspace_bytes(int B, int N, int H) {
    if (B <= 0 || N <= 0 || H <= 0) return -1;
    return avd::attn_fp8_ws_bytes(B, N, H);
}
extern "C" int avd_attn_fwd_fp8_f32(const void* qkv3, void* workspace, int64_t workspace_bytes, float* out, void* out3, int B, int N, int H,
                                    int n_query, avd_stream_t stream) {
    return avd::attn_fp8(qkv3, workspace, workspace_bytes, out, out3, B, N, H, n_query, static_cast<hipStream_t>(stream), 6, 1.f, 1.f);
}
extern "C" int avd_attn_fwd_fp8_f16x2_f32(const void* qkv, void* workspace, int64_t workspace_bytes, float* out, void* out2, int B, int N, int H,
                                          int n_query, float qkv_scale, float out_scale, avd_stream_t stream) {
    return avd::attn_fp8(qkv, workspace, workspace_bytes, out, out2, B, N, H, n_query, static_cast<hipStream_t>(stream), 3, qkv_scale, out_scale);
}
