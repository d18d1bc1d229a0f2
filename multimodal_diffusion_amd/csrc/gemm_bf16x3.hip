// Linear layers on the 16-bit matrix pipes with split operands — an alternative to gemm_f32.hip for the projections of a Block
// (avdiff/models/mmdt.py:60,77-83) and of the noise head (heads/noise_heads.py:141-147, 206-223) when the batch is large.
//
// Why: on gfx950 v_mfma_f32_32x32x2_f32 runs at 1/16 of the 16-bit MFMA rate (157 vs 2,516 TFLOP/s).  Modes (`terms`):
//   6  "bf16x3": every fp32 value splits EXACTLY into three bf16 planes, x = h + m + l (8 significant bits each, bf16 has fp32's
//      exponent range, so no scaling is needed).  A product keeps the six terms down to 2^-16 — hh, hm, mh, hl, lh, mm — and drops
//      ml, lm, ll (<= 2^-24 relative), each term accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  Against fp64 the error is that
//      of the fp32 FMA chain (3.5e-6 vs 3.8e-6 max at K=512, tools/micro/split_lab2.hip check).  Six bf16 MFMAs of 32 cycles
//      replace eight fp32 MFMAs of 64 cycles per k=16: 2.67x fewer matrix-pipe cycles.
//   9  "bf16x3_strict": all nine terms, nothing dropped.      1  "bf16": the h plane only — reduced precision, reported error.
//   3  "f16x2": two fp16 planes of a value scaled by a caller-chosen power of two (22 significant bits, avd_common.h), three terms
//      hl, lh, hh on v_mfma_f32_32x32x16_f16: half the matrix-pipe work of bf16x3, measured error at or below the fp32 kernels'.
//
// Operand image ("split3"): X[rows][K] fp32 -> T[ceil(rows/128)][K/16][3 planes][128 rows][32 B]; inside a 12 KiB chunk
// the 16 bytes of (plane p, row r, half = (k%16)/8) sit at p*4096 + r*32 + (half ^ ((r>>4)&1))*16 (conflict-free for the
// ds_read_b128 fragment reads of BOTH MFMA shapes: 32 rows x k-half, and 16 rows x 4 k-groups).  One block's K-tile of
// an operand is then contiguous, already bank-swizzled pieces — the global->LDS DMA copies whole 1 KiB pieces (only the planes the
// mode uses), the ds_read_b128 of fragment rows is conflict free — and a producer's store of one plane for consecutive rows is
// contiguous too.  Producers write the image directly (rmsnorm_split3_kernel and layernorm_act_split3_kernel here, the attention
// epilogue in attn_bf16x3.hip, the bias / bias+GELU epilogues below), so no fp32 copy of those activations exists in these modes;
// the in_proj epilogue writes the "qkv3" image the attention kernel reads (layout in avd_common.h).
//
// Kernel: gemm_bf16x3_kernel<EPI, TERMS, WAVES> (wave tile 128x64 = 4x2 accumulators, K-tile 16, XCD-contiguous super-tiles, LDS stage
// = the planes moved, compact), a software pipeline with the MFMAs of a step issued first and the fragment reads / DMA pieces of
// the following tiles spread between them (see the kernel):
//   WAVES = 8   256x256 block, one block per CU, ring of 3 / 4 / 6 stages (3 / 2 / 1 planes);
//   WAVES = 4   256x128 block, two blocks per CU, 2 / 3 / 4 stages — for the heavy epilogues (image outputs) and for batches
//               that do not fill 256-row tiles;
//   EPI_RES_NORM (f16x2, N = 512): 128x512 row-owner block, 8 waves side by side — the RMSNorm that follows the residual add is
//               finished in the epilogue.
// bf16x3 (six terms) runs by default on gemm_bf16x3_m16_kernel<EPI, WAVES, RT> below: v_mfma_f32_16x16x32_bf16 with TWO product terms
// per instruction (same matrix-pipe cycles, higher sustained clock), 224-row blocks (RT = 7) where that keeps a launch in one generation.
// What caps the rate is power: bf16x3 holds 2.07 GHz at 1.33 kW, f16x2 1.91 GHz at the 1.4 kW cap.
#include "avd_common.h"
#include "s3_common.h"

#include <stdlib.h>
#include <type_traits>

namespace avd {


// ---------------------------------------------------------------------------------------------------------
// producers
// ---------------------------------------------------------------------------------------------------------
// x [rows][K] (row stride ld) -> split3 image; rows in [rows, rows_pad) are written as zeros.  ss != null (K % 64 == 0): also the
// rows' sums of squares per 64-column chunk, ss[row][K / 64] — the table a norm-folded GEMM reads (S3Args::ss_in)
template <bool F16>   // F16: f16x2 image with scale s (avd_common.h)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, RowMap xm, unsigned char* __restrict__ out,
                                                     int64_t rows, int64_t rows_pad, int K, float s, float* __restrict__ ss) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = K >> 3;
    if (i >= rows_pad * per_row) return;
    const int64_t r = i / per_row;
    const int k = (int)(i % per_row) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
        const float* xr = x + xm.off(r) + k;
        *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(xr);
        *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(xr + 4);
    }
    if constexpr (F16) store_split8_h2(out, r, k, K, v, s);
    else store_split8(out, r, k, K, v);
    if (ss) {       // 8 consecutive threads hold one 64-column chunk of one row (K % 64 == 0 keeps the groups inside a row)
        float q = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        q += __shfl_xor(q, 4, 64);
        if ((threadIdx.x & 7) == 0 && r < rows) ss[r * (K >> 6) + (k >> 6)] = q;
    }
}

// RMSNorm (mmdt.py:39-42, eps outside the sqrt) writing the split3 image of its output; one wave per row
template <int NC, bool F16>   // 8-element chunks per lane: d <= 512 * NC
__global__ __launch_bounds__(256) void rmsnorm_split3_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             unsigned char* __restrict__ out, int64_t rows, int d, float eps,
                                                             float sqrt_d, float s) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * d;
    float v[NC][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
            *reinterpret_cast<f32x4*>(v[i]) = *reinterpret_cast<const f32x4*>(xr + c);
            *reinterpret_cast<f32x4*>(v[i] + 4) = *reinterpret_cast<const f32x4*>(xr + c + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += v[i][e] * v[i][e];
        }
    }
    ss = wave_sum(ss);
    const float den = sqrtf(ss) / sqrt_d + eps;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = scale[c + e] * v[i][e] / den;
            if constexpr (F16) store_split8_h2(out, row, c, d, o, s);
            else store_split8(out, row, c, d, o);
        }
    }
}

// LayerNorm + activation (noise_heads.py:141-147: Linear -> LayerNorm -> act) writing the operand image of its output; one wave
// per row, same arithmetic as layernorm_act_kernel (rowops.hip)
template <int NC, bool F16, bool GELU>   // GELU: the activation is known to be GELU (the shipped heads); otherwise read from `act`
__global__ __launch_bounds__(256) void layernorm_act_split3_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, unsigned char* __restrict__ out,
                                                                   int64_t rows, int d, float eps, int act, float s) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * d;
    float v[NC][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
            *reinterpret_cast<f32x4*>(v[i]) = *reinterpret_cast<const f32x4*>(xr + c);
            *reinterpret_cast<f32x4*>(v[i] + 4) = *reinterpret_cast<const f32x4*>(xr + c + 4);
            sum += (v[i][0] + v[i][1] + v[i][2] + v[i][3]) + (v[i][4] + v[i][5] + v[i][6] + v[i][7]);
        }
    }
    const float mean = wave_sum(sum) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);      // biased variance, as torch.nn.LayerNorm
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < d) {
            float o[8], gm[8], bt[8];
            *reinterpret_cast<f32x4*>(gm) = *reinterpret_cast<const f32x4*>(gamma + c);
            *reinterpret_cast<f32x4*>(gm + 4) = *reinterpret_cast<const f32x4*>(gamma + c + 4);
            *reinterpret_cast<f32x4*>(bt) = *reinterpret_cast<const f32x4*>(beta + c);
            *reinterpret_cast<f32x4*>(bt + 4) = *reinterpret_cast<const f32x4*>(beta + c + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = (v[i][e] - mean) * rstd * gm[e] + bt[e];
                if (GELU || act == AVD_ACT_GELU) t = gelu_erf(t);
                else if (act == AVD_ACT_SILU) t = silu(t);
                else if (act == AVD_ACT_RELU) t = fmaxf(t, 0.f);
                else if (act == AVD_ACT_LEAKY_RELU) t = t > 0.f ? t : 0.1f * t;
                o[e] = t;
            }
            if constexpr (F16) store_split8_h2(out, row, c, d, o, s);
            else store_split8(out, row, c, d, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------

// epilogue: the wave's 128 x 64 accumulator tile goes through a private LDS slab in two 64-row passes and is streamed out as
// whole 16-byte segments (mwave0 = first output row of the wave, nbase = first column)
template <int EPI, bool F16>
__device__ __forceinline__ void s3_epilogue(const S3Args& g, f32x16 (&acc)[4][2], float* slab, int64_t mwave0, int nbase, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int CLD = 64 + 4, TN = 2;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[ps * 2 + i][j][r];
        const int64_t m0 = mwave0 + ps * 64;
        if constexpr (EPI == S3_EPI_GELU_SPLIT || EPI == S3_EPI_SPLIT) {      // (kept for reference builds; the kernel takes the register epilogue)
            // 8 lanes per row (8 columns each), 8 rows per wave instruction
            const int cr = lane >> 3, cc = (lane & 7) * 8;
            const int n = nbase + cc;
            float bv[8];
            *reinterpret_cast<f32x4*>(bv) = *reinterpret_cast<const f32x4*>(g.bias + n);
            *reinterpret_cast<f32x4*>(bv + 4) = *reinterpret_cast<const f32x4*>(g.bias + n + 4);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int64_t m = m0 + cr + it * 8;
                float v[8];
                *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(slab + (cr + it * 8) * CLD + cc);
                *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(slab + (cr + it * 8) * CLD + cc + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = (F16 ? v[e] * g.ab_inv : v[e]) + bv[e];
                    v[e] = EPI == S3_EPI_GELU_SPLIT ? gelu_erf(t) : t;
                }
                if (m < g.M) {
                    if constexpr (F16) store_split8_h2(g.C3, m, n, g.N, v, g.c_scale);
                    else store_split8(g.C3, m, n, g.N, v);
                }
            }
        } else {
            // 16 lanes per row (4 columns each), 4 rows per wave instruction
            const int cr = lane >> 4, cc = (lane & 15) * 4;
            const int n = nbase + cc;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
            float* cptr = g.C + (m0 + cr) * g.N + n;
            const float* rptr = EPI == S3_EPI_RES ? g.R + (m0 + cr) * g.N + n : nullptr;
#pragma unroll
            for (int c0 = 0; c0 < 16; c0 += 8) {
                f32x4 rv[8];
                if constexpr (EPI == S3_EPI_RES) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        rv[u] = m0 + cr + (c0 + u) * 4 < g.M ? *reinterpret_cast<const f32x4*>(rptr + (int64_t)(c0 + u) * 4 * g.N)
                                                            : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int it = c0 + u;
                    f32x4 v = *reinterpret_cast<const f32x4*>(slab + (cr + it * 4) * CLD + cc);
                    if constexpr (F16) v *= g.ab_inv;
                    v += bv;
                    if constexpr (EPI == S3_EPI_RES) v += rv[u];
                    if (m0 + cr + it * 4 < g.M) *reinterpret_cast<f32x4*>(cptr + (int64_t)it * 4 * g.N) = v;
                }
            }
        }
    }
}

// Image epilogues (fc1 + GELU -> operand image, head input_proj -> image, in_proj -> q|k|v image) work on TRANSPOSED accumulator
// tiles: the kernel multiplies W fragments as the MFMA's A operand and X fragments as its B operand, so lane (l31, hi) of tile (i, j)
// holds output row m = 32 i + l31 and columns n = 32 j + 8 (r >> 2) + 4 hi + (r & 3) in register r.  One v_permlane32_swap per
// register pair then gives every lane 8 CONSECUTIVE columns of its row — lanes hi = 0 the 16-byte chunks 0 and 2 of the 32 columns,
// lanes hi = 1 chunks 1 and 3 — which is exactly what an image stores per (row, 8-k chunk): bias / GELU / split run on registers and
// every wave store writes 1 KiB of contiguous image (32 rows x 32 B).  No LDS slab, no barrier, no transposing ds_write / ds_read
// (rounds 1-2 went through the slab: ~6,000 instructions per wave, a third of them exec-mask bookkeeping of the per-pair range
// check; the GELU -> image epilogue was 39-46 % of a block's life and paced the launch, tools/micro/s3_stamps.py).
// `big` (wave-uniform; bf16 planes only): some value of this wave's VALID rows may come within reach of the top of the bf16 range
// (|v| > largest finite bf16, or inf) — only then is split8's edge handling executed.  The test is made once per wave tile, and it
// selects nothing but the (exact) split: bias, scaling and GELU are one piece of code on both sides, so a wave's results do not
// depend on which side it took, i.e. not on what its neighbouring rows — or the never-written rows past M — hold.
template <int EPI, bool F16>
__device__ __forceinline__ void s3_epilogue_img_t(const S3Args& g, f32x16 (&acc)[4][2], int64_t mwave0, int nbase, int lane, bool big) {
    const int l31 = lane & 31, hi = lane >> 5;
    // bias of this lane's chunks: tile column j, chunk group cg -> columns nbase + 32 j + 8 (2 cg + hi) + (0..7)
    float bv[2][2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const float* bp = g.bias + nbase + 32 * j + 8 * (2 * cg + hi);
            *reinterpret_cast<f32x4*>(bv[j][cg]) = *reinterpret_cast<const f32x4*>(bp);
            *reinterpret_cast<f32x4*>(bv[j][cg] + 4) = *reinterpret_cast<const f32x4*>(bp + 4);
        }
    [[maybe_unused]] float mul = 1.0f;
    [[maybe_unused]] float ssq[4] = {0.f, 0.f, 0.f, 0.f};      // EPI_RES_IMG: this lane's part of its rows' sums of squares
    [[maybe_unused]] int64_t qbase = 0;       // EPI_QKV3: byte offset of (part, sample 0, head, token 0)
    if constexpr (EPI == S3_EPI_QKV3) {
        const int dmodel = g.heads * 64;
        const int part = nbase / dmodel, head = (nbase % dmodel) >> 6;       // a wave's 64 columns are one (part, head)
        mul = part == 0 ? g.qscale : 1.0f;
        qbase = (((int64_t)part * (g.M / g.tokN)) * g.heads + head) * (int64_t)g.tokNpad * QKV3_ROWB;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = mwave0 + 32 * i + l31;
        // folded RMSNorm: factor of this lane's row from the partial sums of squares (fixed summation order)
        float rinv = 1.0f;
        if (EPI != S3_EPI_RES_IMG && g.ss_in != nullptr && m < g.M) {
            const int nc = g.K >> 6;
            const float* sp = g.ss_in + m * nc;
            float ssum = 0.f;
            if (nc == 8) {
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(sp), p1 = *reinterpret_cast<const f32x4*>(sp + 4);
                const f32x4 t = p0 + p1;
                ssum = (t[0] + t[1]) + (t[2] + t[3]);
            } else {
                for (int c = 0; c < nc; ++c) ssum += sp[c];
            }
            rinv = 1.0f / (sqrtf(ssum) / g.ss_sqrt_d + g.ss_eps);
        }
        [[maybe_unused]] unsigned char* qrow = nullptr;
        [[maybe_unused]] int qsw = 0;
        if constexpr (EPI == S3_EPI_QKV3) {
            const unsigned b = (unsigned)m / (unsigned)g.tokN, tok = (unsigned)m - b * (unsigned)g.tokN;      // M < 2^31 (checked by the host)
            const int dmodel = g.heads * 64;
            qrow = g.C3 + qbase + ((int64_t)b * g.heads * g.tokNpad + tok) * QKV3_ROWB;
            qsw = qkv3_swizzle(nbase / dmodel, (int)tok);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                // registers 8 cg .. 8 cg + 3 (columns 4 hi + ..) and 8 cg + 4 .. + 7 (columns 8 + 4 hi + ..): swap across the lane halves
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * cg + e]), __float_as_uint(acc[i][j][8 * cg + 4 + e]),
                                                                     false, false);
                    v[e] = __uint_as_float(sw[0]);
                    v[4 + e] = __uint_as_float(sw[1]);
                }
                if constexpr (EPI == S3_EPI_QKV3) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(F16 ? v[e] * g.ab_inv : v[e], rinv, bv[j][cg][e]) * mul;
                    if (S3_ROW_OK(m)) {
                        const int c = 4 * j + 2 * cg + hi;                                  // 8-d chunk of the head
                        unsigned char* dst = qrow + ((c ^ qsw) << 4);
                        if constexpr (F16) {
                            u32x4 Hh, Lo;
                            split8_h2(v, g.c_scale, Hh, Lo);
                            *reinterpret_cast<u32x4*>(dst) = Hh;
                            *reinterpret_cast<u32x4*>(dst + 128) = Lo;
                        } else {
                            u32x4 Hh, Mi, Lo;
                            if (big) split8<true>(v, Hh, Mi, Lo);
                            else split8<false>(v, Hh, Mi, Lo);
                            *reinterpret_cast<u32x4*>(dst) = Hh;
                            *reinterpret_cast<u32x4*>(dst + 128) = Mi;
                            *reinterpret_cast<u32x4*>(dst + 256) = Lo;
                        }
                    }
                } else if constexpr (EPI == S3_EPI_RES_IMG) {
                    // new residual stream: fp32 [M][N] (8 consecutive floats of the lane's row), its operand image, its sum of squares
                    const int n = nbase + 32 * j + 8 * (2 * cg + hi);
                    if (m < g.M) {
                        const float* rp = g.R + m * g.N + n;
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = ((F16 ? v[e] * g.ab_inv : v[e]) + bv[j][cg][e]) + (e < 4 ? r0[e] : r1[e - 4]);
                        float* cp = g.C + m * g.N + n;
                        *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                        ssq[i] += ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
                        if constexpr (F16) store_split8_h2(g.C3, m, n, g.N, v, g.c_scale);
                        else if (big) store_split8<true>(g.C3, m, n, g.N, v);
                        else store_split8<false>(g.C3, m, n, g.N, v);
                    }
                } else {
                    const int n = nbase + 32 * j + 8 * (2 * cg + hi);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = fmaf(F16 ? v[e] * g.ab_inv : v[e], rinv, bv[j][cg][e]);
                        v[e] = EPI == S3_EPI_GELU_SPLIT ? gelu_erf(t) : t;
                    }
                    if (S3_ROW_OK(m)) {
                        if constexpr (F16) store_split8_h2(g.C3, m, n, g.N, v, g.c_scale);
                        else if (big) store_split8<true>(g.C3, m, n, g.N, v);
                        else store_split8<false>(g.C3, m, n, g.N, v);
                    }
                }
            }
    }
    if constexpr (EPI == S3_EPI_RES_IMG) {
        // the two lane halves of a row hold its columns 4 hi + .. of every 8: one exchange, then lane hi = 0 owns the wave's 64 columns
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float tot = ssq[i] + __shfl_xor(ssq[i], 32, 64);
            const int64_t m = mwave0 + 32 * i + l31;
            if (hi == 0 && m < g.M) g.ss_out[m * (g.N >> 6) + (nbase >> 6)] = tot;
        }
    }
}

template <int EPI, bool F16>
__device__ __forceinline__ void s3_epilogue_img(const S3Args& g, f32x16 (&acc)[4][2], int64_t mwave0, int nbase, int lane) {
    bool big = false;
    if constexpr (!F16) {       // fp16 planes: out-of-range turns into inf / NaN by itself
        // |raw sum| bounds what the epilogue can produce only loosely (bias, q scale), so the test is generous: anything past 2^100
        const int l31 = lane & 31;
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) a = fmaxf(a, fabsf(acc[i][j][r]));     // NaN operands are ignored: NaN stays NaN either way
            if (mwave0 + 32 * i + l31 < g.M) amax = fmaxf(amax, a);                 // rows past M hold whatever was in memory
        }
        float bmax = 0.f;
#pragma unroll
        for (int e = 0; e < 64; e += 4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + nbase + e);
            bmax = fmaxf(fmaxf(bmax, fmaxf(fabsf(b4[0]), fabsf(b4[1]))), fmaxf(fabsf(b4[2]), fabsf(b4[3])));
        }
        big = __any(!(amax < 1.2676506e30f) || !(bmax < 1.2676506e30f) || !(fabsf(g.qscale) < 1.0e6f));
        // the residual stream is unbounded and is read only inside the epilogue: its image always takes the checked split
        if constexpr (EPI == S3_EPI_RES_IMG) big = true;
    }
    s3_epilogue_img_t<EPI, F16>(g, acc, mwave0, nbase, lane, big);
}

// EPI_RES_NORM (row-owner blocks: 8 waves side by side over the 512 columns of 128 rows).  Pass 1: v = acc + bias + residual is stored
// as fp32 and kept in the accumulator registers; the rows' sums of squares go wave -> LDS -> every wave (fixed order: lane halves,
// then waves 0..7).  Pass 2: scale * v / (sqrt(ss) / sqrt(d) + eps), the expression rmsnorm_split3_kernel evaluates, split and stored
// as the image.  `red` = 8 x 128 floats overlaying the stages.
template <bool F16>
__device__ __forceinline__ void rown_pass1(const S3Args& g, f32x16 (&acc)[4][2], int64_t mblock0, int nbase, int lane, float (&ssq)[4]) {
    const int l31 = lane & 31, hi = lane >> 5;
    float bv[2][2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const float* bp = g.bias + nbase + 32 * j + 8 * (2 * cg + hi);
            *reinterpret_cast<f32x4*>(bv[j][cg]) = *reinterpret_cast<const f32x4*>(bp);
            *reinterpret_cast<f32x4*>(bv[j][cg] + 4) = *reinterpret_cast<const f32x4*>(bp + 4);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = mblock0 + 32 * i + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * cg + e]), __float_as_uint(acc[i][j][8 * cg + 4 + e]),
                                                                     false, false);
                    v[e] = __uint_as_float(sw[0]);
                    v[4 + e] = __uint_as_float(sw[1]);
                }
                const int n = nbase + 32 * j + 8 * (2 * cg + hi);
                if (m < g.M) {
                    const float* rp = g.R + m * g.N + n;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((F16 ? v[e] * g.ab_inv : v[e]) + bv[j][cg][e]) + (e < 4 ? r0[e] : r1[e - 4]);
                    float* cp = g.C + m * g.N + n;
                    *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    ssq[i] += ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[i][j][8 * cg + e] = v[e];       // this lane's 8 consecutive columns of row m
            }
    }
}

template <bool F16>
__device__ __forceinline__ void rown_pass2(const S3Args& g, f32x16 (&acc)[4][2], int64_t mblock0, int nbase, int lane, const float (&den)[4]) {
    const int l31 = lane & 31, hi = lane >> 5;
    float gv[2][2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const float* gp = g.gamma + nbase + 32 * j + 8 * (2 * cg + hi);
            *reinterpret_cast<f32x4*>(gv[j][cg]) = *reinterpret_cast<const f32x4*>(gp);
            *reinterpret_cast<f32x4*>(gv[j][cg] + 4) = *reinterpret_cast<const f32x4*>(gp + 4);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = mblock0 + 32 * i + l31;
        if (m < g.M) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gv[j][cg][e] * acc[i][j][8 * cg + e] / den[i];
                    const int n = nbase + 32 * j + 8 * (2 * cg + hi);
                    if constexpr (F16) store_split8_h2(g.C3, m, n, g.N, v, g.c_scale);
                    else store_split8<true>(g.C3, m, n, g.N, v);
                }
        }
    }
}

// rows' sums of squares: lane halves, then the eight 64-column parts of the row in part order (red = 8 x 128 floats)
__device__ __forceinline__ void rown_publish(const float (&ssq)[4], int part, int lane, float* red) {
    const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float tot = ssq[i] + __shfl_xor(ssq[i], 32, 64);
        if (hi == 0) red[part * 128 + 32 * i + l31] = tot;
    }
}
__device__ __forceinline__ void rown_den(const S3Args& g, int lane, const float* red, float (&den)[4]) {
    const int l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float ss = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) ss += red[w8 * 128 + 32 * i + l31];
        den[i] = sqrtf(ss) / g.ss_sqrt_d + g.ss_eps;
    }
}

template <bool F16>
__device__ __forceinline__ void s3_epilogue_rownorm(const S3Args& g, f32x16 (&acc)[4][2], int64_t mblock0, int wave, int lane, float* red) {
    float ssq[4] = {0.f, 0.f, 0.f, 0.f}, den[4];
    rown_pass1<F16>(g, acc, mblock0, wave * 64, lane, ssq);
    __syncthreads();          // every wave is past its last fragment read: the stages may be overwritten
    rown_publish(ssq, wave, lane, red);
    __syncthreads();
    rown_den(g, lane, red, den);
    rown_pass2<F16>(g, acc, mblock0, wave * 64, lane, den);
}

// wait until at most `tiles` x STEP of this wave's DMA pieces are still in flight (tiles is wave-uniform; capped at MAXN, < 0 = 0)
template <int MAXN, int STEP>
__device__ __forceinline__ void wait_vm_tiles(int tiles) {
    if constexpr (MAXN <= 0) { wait_vm<0>(); }
    else {
        if (tiles >= MAXN) wait_vm<MAXN * STEP>();
        else wait_vm_tiles<MAXN - 1, STEP>(tiles);
    }
}

// Tile configurations (wave tile 128 x 64 = 4 x 2 accumulators in both):
//   WAVES = 8: 256 x 256 block, one block per CU (out_proj / fc2 + residual when the grid covers most of the CUs);
//   WAVES = 4: 256 x 128 block, TWO blocks per CU, so one block's epilogue (the image epilogues move 1.5x the bytes of an fp32 one and
//              carry the GELU) runs beside the other's main loop.
// LDS stage = the planes a mode moves, compact: [region (128 rows)][plane][128 rows][32 B]; region stride = planes x 4 KiB; a
// ring of NST stages.
//   ROWN (8 waves, f16x2): 128 x 512 block — the block owns whole rows of an N = 512 output (EPI_RES_NORM); waves 1 x 8.
template <int TERMS, int WAVES, bool ROWN = false>
struct S3Cfg {
    static constexpr int BM = ROWN ? 128 : 256, BN = ROWN ? 512 : WAVES == 8 ? 256 : 128;
    static constexpr int NPL = s3_planes(TERMS);              // planes moved and read
    static constexpr int REGIONS = (BM + BN) / 128;           // 128-row regions per stage: A0 A1 W0 (W1)
    static constexpr int RCH = NPL * S3_PLANE;                // one region
    static constexpr int STAGE = REGIONS * RCH;               // 8 waves: 48 / 32 / 16 KiB; 4 waves: 36 / 24 / 12 KiB
    static constexpr int PPR = 4 * NPL;                       // one-KiB pieces per region
    static constexpr int PPW = REGIONS * PPR / WAVES;         // DMA pieces per wave per stage: 2 NPL (8 waves), 3 NPL (4 waves)
    static constexpr int NST = ROWN ? 3 : WAVES == 8 ? (NPL == 3 ? 3 : NPL == 2 ? 4 : 6) : (NPL == 3 ? 2 : NPL == 2 ? 3 : 4);
    static constexpr int SLABS = WAVES * 64 * 68 * 4;         // epilogue slabs, overlay the stages
    static constexpr int LDS = NST * STAGE > SLABS ? NST * STAGE : SLABS;
    static_assert(REGIONS * PPR % WAVES == 0, "pieces divide evenly over the waves");
};

// Product terms kept per k (A plane, B plane), in the order they are issued — small first, and chosen so that every operand plane
// but two is dead before the step ends and can be re-read IN PLACE for the next K tile:
//   6 (bf16x3, default): (m,m) (m,h) (l,h) (h,h) (h,m) (h,l) — everything down to 2^-16 relative, the error of an fp32 FMA chain;
//   9 (strict): all nine, nothing dropped;   1: (h,h) only — plain bf16 operands, reduced precision (BASELINE config C2);
//   3 (f16x2 images, planes h, l of 11 significant bits each): (h,l) (l,h) (h,h); (l,l) <= 2^-22 relative is dropped.
//
// Main loop (software pipeline, one barrier per 16-k step).  A wave holds the fragments of tile kt in registers; during step kt it
//   * issues the MFMAs of tile kt FIRST — right behind the barrier the matrix pipe has work from every wave,
//   * reads the fragments of tile kt+1 from its LDS stage between the MFMA groups, each plane into the registers of the plane that
//     has just died (the two planes that live to the end of a step — A plane h and one B plane — alternate between two register
//     sets, the loop is unrolled twice, nothing is copied),
//   * issues the global -> LDS DMA of tile kt+NST, one 1-KiB piece behind every four MFMAs (an LDS-DMA costs its wave 60-180 issue
//     cycles: in a block at the top of the step, as rounds 1-2 had it, the nine pieces of the 4-wave kernel kept the wave off the
//     matrix pipe for ~800 of a step's ~2,800 cycles; round-2 stamps), into the stage tile kt has just vacated.
// The ring's stages hold tiles kt+1 .. kt+NST.  Barrier invariants, top of step kt: every wave's own pieces of tile kt+1 have
// landed (counted vmcnt: the (NST-2) x PPW pieces of tiles kt+2.. may stay in flight) and its reads of tile kt are complete
// (lgkmcnt(0)); behind the barrier all of tile kt+1 is readable and tile kt's stage may be overwritten.
template <int EPI, int TERMS, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void gemm_bf16x3_kernel(S3Args g) {
    constexpr bool ROWN = EPI == S3_EPI_RES_NORM;
    using Cf = S3Cfg<TERMS, WAVES, ROWN>;
    constexpr int BM = Cf::BM, BN = Cf::BN, WM = 128, WN = 64, TM = 4, TN = 2;
    constexpr int RCH = Cf::RCH, STAGE = Cf::STAGE, PPW = Cf::PPW, NST = Cf::NST;
    constexpr int RA = BM / 128;                    // 128-row regions of A in a stage (the W regions follow)
    constexpr bool F16 = TERMS == 3;
    static_assert(!ROWN || WAVES == 8, "row-owner blocks: 8 waves");
    // image epilogues take the accumulator tiles transposed (s3_epilogue_img): W fragments as the MFMA's A operand
    constexpr bool TR = EPI == S3_EPI_GELU_SPLIT || EPI == S3_EPI_SPLIT || EPI == S3_EPI_QKV3 || EPI == S3_EPI_RES_IMG || ROWN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

    [[maybe_unused]] const unsigned long long t_entry = S3_T(), rt_entry = S3_RT();
    int bm, bn;
    s3_block_of(g, (int)((g.M + BM - 1) / BM), bm, bn);
    if ((int64_t)bm * BM >= g.M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keep what derives from it scalar
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int WAVES_N = BN / WN;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    if constexpr (WAVES == 4) {
        if (g.stagger > 0) {
            // Two blocks share a CU.  Started together they stay in phase: both in the K loop (each at half the matrix pipe's rate),
            // then both in the epilogue (the pipe idle).  The block whose waves sit in the odd slots of their SIMDs starts late, so
            // one block's epilogue runs beside the other's loop.
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            if ((hwid & 1u) && (int)blockIdx.x < g.first_gen)      // first generation only (2 slots per CU); later blocks inherit the phase
                for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(16);
        }
    }
    const int nk = g.K >> 4;
    const int nrtA = (int)((g.M + 127) >> 7);
    // split-K launches (gemm_bf16x3_splitk): blockIdx.y = K slice z of gridDim.y; g.K is ONE slice's length, the images hold all of
    // them (a row group's k-groups are nk * gridDim.y apart), slice z writes its partial sums to C + z * M * N
    const int nz = gridDim.y, kz = blockIdx.y;
    if (nz > 1) g.C += (int64_t)kz * g.M * g.N;

    // fragment addresses inside a stage: the wave's A rows all sit in region wm, its W rows in one W region, and rows 32 apart
    // are 1 KiB apart with the same chunk swizzle — one lane offset per operand, everything else is an immediate
    // (fragment i of plane p at a_base + i * 1024 + p * S3_PLANE)
    const int swz = (hi ^ ((l31 >> 4) & 1)) << 4;
    const int a_base = wm * RCH + l31 * 32 + swz;
    const int b_base = (RA + (wn * WN) / 128) * RCH + ((wn * WN) & 127) * 32 + l31 * 32 + swz;

    // DMA: the stage image is [A row-tile 0 | A row-tile 1 | W row-tile 0 (| W row-tile 1)], each a 12 KiB chunk in global memory
    // of which the first NPL planes (4 KiB each) are moved in 1-KiB pieces.  4 waves: every wave moves a quarter of each of the
    // three regions; 8 waves: every wave moves half of one region.  A piece's source is a wave-uniform pointer (scalar registers:
    // region base + K-tile + piece) plus the lane's 16 bytes.
    // Row-owner blocks (five regions: A0 W0 W1 W2 W3): piece P = PPW wave + i of the stage's pieces, region P / pieces-per-region.
    constexpr int NRW = ROWN ? PPW : WAVES == 4 ? 3 : 1;    // regions (row-owner: single pieces) a wave moves pieces of
    constexpr int PRW = PPW / NRW;                          // pieces per such region
    const unsigned char* rbase[NRW];
    int rdst[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) {
        const int region = ROWN ? (wave * PPW + r) / Cf::PPR : WAVES == 4 ? r : wave >> 1;
        const int within = (ROWN ? (wave * PPW + r) % Cf::PPR : WAVES == 4 ? wave * PRW : (wave & 1) * PRW) * 1024;
        rdst[r] = region * RCH + within;
        if (region < RA) {
            int rt = bm * RA + region;
            rt = rt < nrtA ? rt : nrtA - 1;
            rbase[r] = g.A + ((int64_t)rt * nz + kz) * nk * S3_CHUNK + within;
        } else {
            rbase[r] = g.W + ((int64_t)(bn * (BN / 128) + region - RA) * nz + kz) * nk * S3_CHUNK + within;
        }
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_piece = [&](int i, int kt, int buf) {
        const int r = i / PRW, k = (i % PRW) * 1024;
        __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(rbase[r] + ((int64_t)kt * S3_CHUNK + k) + lane16),
                                         AVD_LDS_PTR(smem3 + (buf * STAGE + rdst[r] + k)), 16, 0, 0);
    };
    auto issue_tile = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_piece(i, kt, buf);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragments: planes that die before the end of a step are re-read in place (e_*); the two that live to the end alternate
    // between the sets L0 / L1
    struct Late { bf16x8 a[TM]; bf16x8 b[TN]; };
    Late L0, L1;
    [[maybe_unused]] bf16x8 ea1[TM], ea2[TM], eb0[TN], eb1[TN];      // A planes 1, 2; B planes 0, 1 (use depends on TERMS)
    auto lda = [&](bf16x8 (&dst)[TM], const unsigned char* st, int p) {
#ifdef AVD_LAB_NOLDS       // diagnostic build: fragments stay whatever the registers hold (results are wrong by design)
        if (p >= 0) { asm volatile("" : "+v"(dst[0])); return; }
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(st + a_base + i * 1024 + S3_PLANE * p);
    };
    auto ldb = [&](bf16x8 (&dst)[TN], const unsigned char* st, int p) {
#ifdef AVD_LAB_NOLDS
        if (p >= 0) { asm volatile("" : "+v"(dst[0])); return; }
#endif
#pragma unroll
        for (int j = 0; j < TN; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(st + b_base + j * 1024 + S3_PLANE * p);
    };
    // MFMAs of rows [i0, i1) of one product term
    auto mm = [&](const bf16x8 (&A_)[TM], const bf16x8 (&B_)[TN], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = TR ? mma16<F16>(B_[j], A_[i], acc[i][j]) : mma16<F16>(A_[i], B_[j], acc[i][j]);
    };

    // prologue: the whole ring is put in flight, tile 0 is awaited and its fragments read
    const int npre = nk < NST ? nk : NST;
#pragma unroll
    for (int t = 0; t < NST; ++t)
        if (t < nk) issue_tile(t, t);
    wait_vm_tiles<NST - 1, PPW>(npre - 1);
    asm volatile("s_barrier" ::: "memory");
    // which B plane lives to the end of a step: l (plane 2) for 6 / 9 terms, h (plane 0) for 3 / 1
    constexpr int LATE_B = (TERMS == 6 || TERMS == 9) ? 2 : 0;
    lda(L0.a, smem3, 0);
    ldb(L0.b, smem3, LATE_B);
    if constexpr (TERMS == 6 || TERMS == 9) {
        lda(ea1, smem3, 1);
        lda(ea2, smem3, 2);
        ldb(eb0, smem3, 0);
        ldb(eb1, smem3, 1);
    } else if constexpr (TERMS == 3) {
        lda(ea1, smem3, 1);
        ldb(eb1, smem3, 1);
    }

    int st_cur = 0, st_nx = 1 % NST;      // stage of tile kt (refilled with tile kt+NST), stage of tile kt+1
#define S3_SB() __builtin_amdgcn_sched_barrier(0)
    // DMA pieces per slot (a slot = behind four MFMAs): one, except where the ring is two stages deep (the tile issued in step kt is
    // awaited at the top of step kt+1: front-load it) or a step has fewer slots than pieces
    constexpr int NSLOT = 2 * TERMS;
    constexpr int PER = NST == 2 ? 2 : (PPW + NSLOT - 1) / NSLOT;
    auto step = [&](auto main_tag, int kt, Late& cur, Late& nxt) {
        constexpr bool MAIN = decltype(main_tag)::value;
        if constexpr (MAIN) wait_vm<(NST - 2) * PPW>();
        else wait_vm_tiles<NST - 2, PPW>((kt + NST - 1 < nk - 1 ? kt + NST - 1 : nk - 1) - (kt + 1));
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned char* nx = smem3 + st_nx * STAGE;
        const int dbuf = st_cur;
        st_cur = st_nx;
        st_nx = st_nx + 1 == NST ? 0 : st_nx + 1;
        if constexpr (!MAIN) {
            if (kt + NST < nk) issue_tile(kt + NST, dbuf);      // the odd step between the unrolled main loop and the tail
        }
        // slot s: DMA pieces [s PER, (s+1) PER) of tile kt+NST
        auto slot = [&](int s) {
#ifdef AVD_LAB_NODMA       // diagnostic build: no global traffic inside the main loop (results are wrong by design)
            if constexpr (false) {
#else
            if constexpr (MAIN) {
#endif
#pragma unroll
                for (int q = 0; q < PER; ++q)
                    if (s * PER + q < PPW) { S3_SB(); issue_piece(s * PER + q, kt + NST, dbuf); S3_SB(); }
            }
        };
        S3_SB();
        if constexpr (TERMS == 1) {
            mm(cur.a, cur.b, 0, 2); slot(0); lda(nxt.a, nx, 0); S3_SB();
            mm(cur.a, cur.b, 2, 4); slot(1); ldb(nxt.b, nx, 0); S3_SB();
        } else if constexpr (TERMS == 3) {                 // planes: 0 = h, 1 = l
            mm(cur.a, eb1, 0, 2); slot(0); S3_SB();
            mm(cur.a, eb1, 2, 4); slot(1); ldb(eb1, nx, 1); lda(nxt.a, nx, 0); S3_SB();      // (h,l)  -> B l dead
            mm(ea1, cur.b, 0, 2); slot(2); S3_SB();
            mm(ea1, cur.b, 2, 4); slot(3); lda(ea1, nx, 1); ldb(nxt.b, nx, 0); S3_SB();      // (l,h)  -> A l dead
            mm(cur.a, cur.b, 0, 2); slot(4); S3_SB();
            mm(cur.a, cur.b, 2, 4); slot(5); S3_SB();                                         // (h,h)
        } else {                                            // planes: 0 = h, 1 = m, 2 = l
            int s = 0;
            mm(ea1, eb1, 0, 2); slot(s++); S3_SB();
            mm(ea1, eb1, 2, 4); slot(s++); lda(nxt.a, nx, 0); S3_SB();                       // (m,m)
            if constexpr (TERMS == 9) {
                mm(ea1, cur.b, 0, 2); slot(s++); S3_SB();
                mm(ea1, cur.b, 2, 4); slot(s++); S3_SB();                                     // (m,l)
            }
            mm(ea1, eb0, 0, 2); slot(s++); S3_SB();
            mm(ea1, eb0, 2, 4); slot(s++); lda(ea1, nx, 1); S3_SB();                         // (m,h)  -> A m dead
            if constexpr (TERMS == 9) {
                mm(ea2, cur.b, 0, 2); slot(s++); S3_SB();
                mm(ea2, cur.b, 2, 4); slot(s++); S3_SB();                                     // (l,l)
                mm(ea2, eb1, 0, 2); slot(s++); S3_SB();
                mm(ea2, eb1, 2, 4); slot(s++); S3_SB();                                       // (l,m)
            }
            mm(ea2, eb0, 0, 2); slot(s++); S3_SB();
            mm(ea2, eb0, 2, 4); slot(s++); lda(ea2, nx, 2); S3_SB();                         // (l,h)  -> A l dead
            mm(cur.a, eb0, 0, 2); slot(s++); S3_SB();
            mm(cur.a, eb0, 2, 4); slot(s++); ldb(eb0, nx, 0); ldb(nxt.b, nx, 2); S3_SB();    // (h,h)  -> B h dead
            mm(cur.a, eb1, 0, 2); slot(s++); S3_SB();
            mm(cur.a, eb1, 2, 4); slot(s++); ldb(eb1, nx, 1); S3_SB();                       // (h,m)  -> B m dead
            mm(cur.a, cur.b, 0, 2); slot(s++); S3_SB();
            mm(cur.a, cur.b, 2, 4); slot(s++); S3_SB();                                       // (h,l)
        }
    };
    static_assert(PER * NSLOT >= PPW, "every DMA piece has a slot");
    using MainT = std::integral_constant<bool, true>;
    using TailT = std::integral_constant<bool, false>;
    [[maybe_unused]] const unsigned long long t_loop = S3_T();
    {
        // main: constant waits, interleaved DMA, an even number of steps; tail: the last NST (+1) steps, counted waits
        const int n_main = nk > NST ? (nk - NST) & ~1 : 0;
        int kt = 0;
        for (; kt < n_main; kt += 2) {
            step(MainT{}, kt, L0, L1);
            step(MainT{}, kt + 1, L1, L0);
        }
        for (; kt + 1 < nk; kt += 2) {
            step(TailT{}, kt, L0, L1);
            step(TailT{}, kt + 1, L1, L0);
        }
        if (kt < nk) step(TailT{}, kt, L0, L1);
    }
#undef S3_SB
    [[maybe_unused]] const unsigned long long t_end = S3_T();
    if constexpr (ROWN) {
        s3_epilogue_rownorm<F16>(g, acc, (int64_t)bm * BM, wave, lane, reinterpret_cast<float*>(smem3));
    } else if constexpr (TR) {
        s3_epilogue_img<EPI, F16>(g, acc, (int64_t)bm * BM + wm * WM, bn * BN + wn * WN, lane);      // registers only: no barrier, no LDS
    } else {
        __syncthreads();      // the slabs overlay the stages: every wave is past its last fragment read, no DMA is in flight
        constexpr int CLD = WN + 4;
        float* slab = reinterpret_cast<float*>(smem3) + wave * 64 * CLD;
        s3_epilogue<EPI, F16>(g, acc, slab, (int64_t)bm * BM + wm * WM, bn * BN + wn * WN, lane);
    }
#ifdef AVD_S3_STAMPS
    const unsigned long long t_issued = S3_T();      // every store issued, none awaited
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S3_DBG(4, t_issued);
    S3_DBG(0, t_entry); S3_DBG(1, t_loop); S3_DBG(2, t_end); S3_DBG(3, S3_T());
    S3_DBG(8, (unsigned long long)nk); S3_DBG(9, S3_RT()); S3_DBG(10, rt_entry);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3 on v_mfma_f32_16x16x32_bf16, TWO product terms per MFMA (round 3)
// ---------------------------------------------------------------------------------------------------------
// Why: under the chip's power management the 16x16x32 shape sustains more than the 32x32x16 shape at equal cycles per FLOP — measured
// on this pool (tools/micro/mfma_shapes.hip, random operands, two waves per SIMD): 2,07-2,14 against 1,79-1,86 PFLOP/s from registers,
// 1,90 against 1,73 with every fragment re-read from LDS (MI355X_MICROARCH.md "DVFS give-back" item 7 reports the same 1.12-1.15x).
// How, without touching the 16-k tiling of the images: the MFMA's K = 32 is fed with TWO planes of the same 16 k — lanes with k-group
// kq = lane >> 4 in {0,1} carry plane X of k 8 kq .. +7, lanes with kq in {2,3} plane Y of k 8 (kq-2) .. +7 — so one MFMA sums two of
// the six terms:   [h|l] x [l|h] = hl + lh,   [h|m] x [m|h] = hm + mh,   [h|m] x [h|m] = hh + mm      (small pairs first).
// A 16-k step of the 128 x 64 wave tile is 3 groups of 32 MFMAs (16 cycles each: the same 1,536 matrix-pipe cycles as 48 of the
// 32x32x16) on 8 x 4 accumulator tiles of 16 x 16 (128 registers, as before), fed by 28 ds_read_b128 (18 before); the fragment of
// row tile i is one read per (pair type): lane (l15, kq) reads row 16 i + l15, k-half kq & 1 of plane (kq < 2 ? X : Y).  The image
// swizzle (half ^ ((row >> 4) & 1)) keeps that read — and the 32-row read of the other modes — bank-conflict free.
// Ring: the stage of tile kt stays resident during step kt (fragments are read group by group: all of them do not fit beside the
// accumulators) and is refilled with tile kt + NST at the top of step kt + 1; DMA pieces are spread behind the MFMAs as above.
// Accumulator tile (i, j), register r:  plain: row 16 i + 4 kq + r, column 16 j + l15;  transposed (image epilogues, W as the first
// operand): row (of the output) 16 i + l15, columns 16 j + 4 kq + r — one v_permlane16_swap per register pair of two row tiles then
// leaves every lane with 8 consecutive columns of one row: lane (l15, kq) holds row 16 (i + (kq & 1)) + l15, columns 16 j + 8 (kq >> 1) ...


// fp32 epilogues (bias / bias + residual) from plain accumulator tiles, through the per-wave LDS slab in two 64-row passes
template <int EPI>
__device__ __forceinline__ void s3_epilogue16(const S3Args& g, f32x4t (&acc)[8][4], float* slab, int64_t mwave0, int nbase, int lane) {
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int CLD = 64 + 4;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int il = 0; il < 4; ++il)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(il * 16 + 4 * kq + r) * CLD + 16 * j + l15] = acc[4 * ps + il][j][r];
        const int64_t m0 = mwave0 + ps * 64;
        // 16 lanes per row (4 columns each), 4 rows per wave instruction
        const int cr = lane >> 4, cc = (lane & 15) * 4;
        const int n = nbase + cc;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
        float* cptr = g.C + (m0 + cr) * g.N + n;
        const float* rptr = EPI == S3_EPI_RES ? g.R + (m0 + cr) * g.N + n : nullptr;
#pragma unroll
        for (int c0 = 0; c0 < 16; c0 += 8) {
            f32x4 rv[8];
            if constexpr (EPI == S3_EPI_RES) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    rv[u] = m0 + cr + (c0 + u) * 4 < g.M ? *reinterpret_cast<const f32x4*>(rptr + (int64_t)(c0 + u) * 4 * g.N) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int it = c0 + u;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab + (cr + it * 4) * CLD + cc);
                v += bv;
                if constexpr (EPI == S3_EPI_RES) v += rv[u];
                if (m0 + cr + it * 4 < g.M) *reinterpret_cast<f32x4*>(cptr + (int64_t)it * 4 * g.N) = v;
            }
        }
    }
}

// RT (blocks with an image epilogue only): row tiles per wave = 32 RT rows per block.  8 waves: 8 = 256-row blocks; 7 = 224-row blocks,
// chosen by the host when it puts every block into ONE generation on the device's CUs (C3: 26,944 rows x 512 columns = 212 blocks of
// 256 x 256 on 256 CUs, 83 % of the chip for the whole launch; 242 blocks of 224 x 256 use 95 % of it and each is 7/8 of the work).
// 4 waves (round 4): 2 .. 7 (2 .. 4 for the residual launches only) — mid-size and small batches, where 256-row blocks leave CUs idle or give a few CUs one block more than the rest
// (the shipped 128 x 128 geometry at batch 32, 8,512 rows: fc2 is 136 blocks of 256 x 128 on 256 CUs, 216 blocks of 160 x 128 put
// the launch on 216 CUs with 5/8 of the work each; the host picks RT per launch, launch_s3t).
// NSTK (4 waves): LDS stages of the ring; 0 = the configuration's two (two blocks per CU, each hiding the other's DMA round trip).
// 4 = 144 KiB, ONE block per CU, for launches whose blocks fit the CUs once: a lone block on a two-stage ring waits out the whole DMA
// round trip of tile kt + 1 every step (issued during step kt, needed at its end); with four stages three tiles are in flight.
template <int EPI, int WAVES, int RT = 8, int NSTK = 0>
__global__ __launch_bounds__(WAVES * 64, 2) void gemm_bf16x3_m16_kernel(S3Args g) {
    using Cf = S3Cfg<6, WAVES>;
    constexpr bool GEN4 = WAVES == 4 && RT < 8;      // 4 waves, block rows start at any multiple of 32: A staged as the 8-wave blocks stage it
    constexpr bool GENA = WAVES == 8 || GEN4;
    constexpr int BM = GENA ? 32 * RT : Cf::BM, BN = Cf::BN, WM = 16 * RT, WN = 64;
    constexpr int RCH = Cf::RCH, STAGE = Cf::STAGE, NST = NSTK ? NSTK : Cf::NST;
    static_assert(NSTK == 0 || (WAVES == 4 && NSTK * STAGE <= 160 * 1024), "deeper ring: the 4-wave blocks");
    // DMA pieces per wave and stage.  GEN4: only the 3 RT live A pieces and the 12 of W are moved, dealt out evenly.  The last wave may be
    // short of pieces: in a two-stage ring every wait is vmcnt(0) and it simply issues fewer; deeper rings count, so there its surplus
    // slots repeat the last piece (same bytes to the same place)
    constexpr int NPIECE = 3 * RT + 12;
    constexpr int PPW = GEN4 ? (NPIECE + 3) / 4 : Cf::PPW;
    constexpr bool TR = EPI == S3_EPI_GELU_SPLIT || EPI == S3_EPI_SPLIT || EPI == S3_EPI_QKV3 || EPI == S3_EPI_RES_IMG || EPI == S3_EPI_BIAS_REG;
    static_assert(RT == 8 || (RT == 7 && WAVES == 8 && TR) || (WAVES == 4 && RT >= 2 && RT <= 7 && TR), "short blocks: register image epilogue");
    // LDS stage.  4 waves, RT = 8: [region A0 A1 W0][plane][128 rows][32 B] as in the 32x32 kernel.  8 waves: [A | W][plane][256 rows][32 B]
    // — plane-major over the block's 256 staged rows, so a wave's row tiles are 512 bytes apart wherever its first row falls.
    // 4 waves, RT < 8: A as the 8 waves have it (24 KiB), W as one region (12 KiB): the same 36 KiB
    constexpr int PSA = GENA ? 256 * 32 : S3_PLANE;                   // plane strides of the A and W parts of the stage
    constexpr int PSB = WAVES == 8 ? 256 * 32 : S3_PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

    [[maybe_unused]] const unsigned long long t_entry = S3_T(), rt_entry = S3_RT();
    int bm, bn;
    s3_block_of(g, (int)((g.M + BM - 1) / BM), bm, bn);
    if ((int64_t)bm * BM >= g.M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    constexpr int WAVES_N = BN / WN;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    if constexpr (WAVES == 4) {
        if (g.stagger > 0) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            if ((hwid & 1u) && (int)blockIdx.x < g.first_gen)
                for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(16);
        }
    }
    const int nk = g.K >> 4;
    const int nrtA = (int)((g.M + 127) >> 7);
#ifdef AVD_LAB_ALIAS       // diagnostic build (tools/micro/s3_phase.py): bit 0 — every block stages row block 0 of A, bit 1 — column block 0 of W:
    // the same instruction stream and L2 -> LDS bytes with (almost) every piece an L2 hit; results are wrong by design
    const int bm_s = (AVD_LAB_ALIAS & 1) ? 0 : bm, bn_s = (AVD_LAB_ALIAS & 2) ? 0 : bn;
#else
    const int bm_s = bm, bn_s = bn;
#endif

    // fragment addressing (see the header of this section): per-lane base for even row tiles, odd tiles flip the k-half slot;
    // the plane a lane reads depends on its k-group pair (kq >> 1) and on the fragment's pair type
    // Only three lane-dependent address registers stay live: the lane's byte offset inside an even / an odd 16-row tile (odd tiles
    // flip the k-half slot) and one plane stride; everything else is wave-uniform (scalar) or an immediate, and an address is one
    // add of (lane base, plane part, uniform part) right before its read (hoisted per-type bases spilled into the loop).
    const int hiq = kq >> 1;
    const int base_e = l15 * 32 + ((kq & 1) << 4), base_o = l15 * 32 + (((kq & 1) ^ 1) << 4);
    // the wave's first row tile is odd in the block when RT is odd and wm = 1: its even / odd bases trade places
    const bool flip_a = ((wm * RT) & 1) != 0;
    const int abase_e = flip_a ? base_o : base_e, abase_o = flip_a ? base_e : base_o;
    const int p1a = hiq * PSA, p1b = hiq * PSB;                      // [h|m]: p1;  [h|l]: 2 p1;  [m|h]: PS - p1;  [l|h]: 2 PS - 2 p1
    const int a_uni = GENA ? wm * (RT * 512) : wm * RCH;
    const int b_uni = WAVES == 8 ? 3 * PSA + wn * WN * 32 : (2 + (wn * WN) / 128) * RCH + ((wn * WN) & 127) * 32;      // (2 RCH == 3 PSA for GEN4)
    enum { T_HM = 0, T_HL = 1, T_MH = 2, T_LH = 3 };
    auto plane_of_a = [&](int type) { return type == T_HM ? p1a : type == T_HL ? 2 * p1a : type == T_MH ? PSA - p1a : 2 * PSA - 2 * p1a; };
    auto plane_of_b = [&](int type) { return type == T_HM ? p1b : type == T_HL ? 2 * p1b : type == T_MH ? PSB - p1b : 2 * PSB - 2 * p1b; };
    auto lda = [&](bf16x8 (&dst)[8], const unsigned char* st, int type, int i0 = 0, int i1 = 8) {
#ifdef AVD_LAB_NOLDS       // diagnostic build: fragments stay whatever the registers hold (results are wrong by design)
        if (type >= 0) { asm volatile("" : "+v"(dst[0])); return; }
#endif
        const int pl = plane_of_a(type);
#pragma unroll
        for (int i = i0; i < i1; ++i)
            if (i < RT) dst[i] = *reinterpret_cast<const bf16x8*>(st + a_uni + ((i & 1) ? abase_o : abase_e) + pl + i * 512);
    };
    auto ldb = [&](bf16x8 (&dst)[4], const unsigned char* st, int type) {
#ifdef AVD_LAB_NOLDS
        if (type >= 0) { asm volatile("" : "+v"(dst[0])); return; }
#endif
        const int pl = plane_of_b(type);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(st + b_uni + ((j & 1) ? base_o : base_e) + pl + j * 512);
    };

    // DMA pieces (1 KiB = one plane of 32 rows of one 16-k group), all wave-uniform.  4 waves: PRW consecutive pieces of each of the
    // three regions.  8 waves: piece P = 6 wave + i of 48 — P < 24: plane P / 8, rows 32 (P % 8) .. +31 of the block's A rows, which
    // start at ANY multiple of 32 (the image keeps 128-row groups: group and 32-row quarter are taken from the global row; rows past
    // the last group re-read it — their results are never stored); P >= 24: the same over the block's 256 W rows.
    // GEN4: live piece n = PPW wave + i of NPIECE — n < 3 RT: plane n / RT, 32-row quarter n % RT of the block's A rows; then the 12 of W.
    constexpr int NRW = WAVES == 4 && !GEN4 ? 3 : PPW;
    constexpr int PRW = PPW / NRW;
    const unsigned char* rbase[NRW];
    int rdst[NRW];
#pragma unroll
    for (int r = 0; r < NRW; ++r) {
        if constexpr (GEN4) {
            int n = wave * PPW + r;
            n = n < NPIECE ? n : NPIECE - 1;               // (a surplus slot: never issued, or a repeat of the last piece — issue_piece)
            const bool isw = n >= 3 * RT;
            const int pq = isw ? n - 3 * RT : n, pl = isw ? pq >> 2 : pq / RT, q = isw ? pq & 3 : pq % RT;
            rdst[r] = isw ? 3 * PSA + pl * S3_PLANE + q * 1024 : pl * PSA + q * 1024;
            if (!isw) {
                const int64_t grow = (int64_t)bm_s * BM + 32 * q;
                int64_t grp = grow >> 7;
                grp = grp < nrtA ? grp : nrtA - 1;
                rbase[r] = g.A + grp * nk * S3_CHUNK + pl * S3_PLANE + (int)((grow >> 5) & 3) * 1024;
            } else {
                rbase[r] = g.W + (int64_t)bn_s * nk * S3_CHUNK + pl * S3_PLANE + q * 1024;
            }
        } else if constexpr (WAVES == 4) {
            const int within = wave * PRW * 1024;
            rdst[r] = r * RCH + within;
            if (r < 2) {
                int rt = bm_s * 2 + r;
                rt = rt < nrtA ? rt : nrtA - 1;
                rbase[r] = g.A + (int64_t)rt * nk * S3_CHUNK + within;
            } else {
                rbase[r] = g.W + (int64_t)bn_s * nk * S3_CHUNK + within;
            }
        } else {
            const int P = wave * PPW + r, isw = P >= 24, pq = isw ? P - 24 : P, pl = pq >> 3, q = pq & 7;
            rdst[r] = (isw ? 3 * PSA : 0) + pl * PSA + q * 1024;
            if (!isw) {
                const int64_t grow = (int64_t)bm_s * BM + 32 * q;
                int64_t grp = grow >> 7;
                grp = grp < nrtA ? grp : nrtA - 1;
                rbase[r] = g.A + grp * nk * S3_CHUNK + pl * S3_PLANE + (int)((grow >> 5) & 3) * 1024;
            } else {
                rbase[r] = g.W + (int64_t)(bn_s * 2 + (q >> 2)) * nk * S3_CHUNK + pl * S3_PLANE + (q & 3) * 1024;
            }
        }
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_piece = [&](int i, int kt, int buf) {
        if constexpr (GEN4 && (NPIECE & 3) != 0 && NST == 2) {
            if (wave * PPW + i >= NPIECE) return;           // wave-uniform
        }
        const int r = i / PRW, k = (i % PRW) * 1024;
        __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(rbase[r] + ((int64_t)kt * S3_CHUNK + k) + lane16),
                                         AVD_LDS_PTR(smem3 + (buf * STAGE + rdst[r] + k)), 16, 0, 0);
    };

    f32x4t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4t{0.f, 0.f, 0.f, 0.f};
    auto mm = [&](const bf16x8 (&A_)[8], const bf16x8 (&B_)[4], int i0, int i1) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
            if (i < RT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = TR ? mma16x16(B_[j], A_[i], acc[i][j]) : mma16x16(A_[i], B_[j], acc[i][j]);
            }
    };

    // prologue: tiles 0 .. NST-2 in flight; step kt issues tile kt + NST - 1 into the stage tile kt - 1 vacated
#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) issue_piece(i, t, t);
        }
    int st_cur = 0, st_fill = NST - 1;
#define S3_SB() __builtin_amdgcn_sched_barrier(0)
    constexpr int NSLOT = 12;                                   // a DMA slot behind every 8 MFMAs
    constexpr int PER = NST == 2 ? 2 : (PPW + NSLOT - 1) / NSLOT;
    static_assert(PER * NSLOT >= PPW, "every DMA piece has a slot");
    auto step = [&](auto main_tag, int kt) {
        constexpr bool MAIN = decltype(main_tag)::value;
        // tile kt has landed (the NST - 2 tiles issued after it may stay in flight); every wave is past its reads of tile kt - 1
        if constexpr (MAIN) wait_vm<(NST - 2) * PPW>();
        else wait_vm_tiles<NST - 2, PPW>((kt + NST - 2 < nk - 1 ? kt + NST - 2 : nk - 1) - kt);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned char* st = smem3 + st_cur * STAGE;
        const int dbuf = st_fill;
        st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
        st_fill = st_fill + 1 == NST ? 0 : st_fill + 1;
        auto slot = [&](int s) {
#ifdef AVD_LAB_NODMA
            if constexpr (false) {
#else
            if constexpr (MAIN) {
#endif
#pragma unroll
                for (int q = 0; q < PER; ++q)
                    if (s * PER + q < PPW) { S3_SB(); issue_piece(s * PER + q, kt + NST - 1, dbuf); S3_SB(); }
            }
        };
        bf16x8 ahl[8], bx[4], by[4];
        S3_SB();
        lda(ahl, st, T_HL);
        ldb(bx, st, T_LH);                       // [l|h]
        slot(0);                                  // the first DMA pieces are issued while those reads are in flight
#ifndef AVD_LAB_HALFREAD
        bf16x8 ahm[8];
        lda(ahm, st, T_HM);
#else       // diagnostic build (round 5, tools/rounds/r5_j.sh): correct results, measured neutral — DESIGN.md 4.4
        // [h|m] shares its lower 32 lanes (the h plane) with [h|l]: once a row-tile pair's hl + lh MFMAs are through, only the upper 32
        // lanes are read again, from the m plane, into the same registers — 12 instead of 16 fragment reads' worth of LDS bytes per step,
        // and 32 fragment registers fewer.  The masked reads are asm (exec is narrowed and restored inside one statement, so the block is
        // not split and the sched_barrier slots hold); the compiler does not count them, so the waits before hm + mh are written out:
        // what may still be in flight is the later pairs' reads and the four [h|m] W reads behind them.
        bf16x8 (&ahm)[8] = ahl;
        const unsigned am_e = (unsigned)(uintptr_t)AVD_LDS_PTR(st + a_uni + abase_e + PSA), am_o = (unsigned)(uintptr_t)AVD_LDS_PTR(st + a_uni + abase_o + PSA);
        auto upm = [&](auto i0_tag) {
            constexpr int I0 = decltype(i0_tag)::value;
            if constexpr (I0 + 1 < RT) {
                asm volatile("s_mov_b64 exec, %[m]\n\tds_read_b128 %[d0], %[a0] offset:%[o0]\n\tds_read_b128 %[d1], %[a1] offset:%[o1]\n\ts_mov_b64 exec, -1"
                             : [d0] "+v"(ahl[I0]), [d1] "+v"(ahl[I0 + 1])
                             : [a0] "v"(am_e), [a1] "v"(am_o), [o0] "n"(I0 * 512), [o1] "n"((I0 + 1) * 512), [m] "s"(0xFFFFFFFF00000000ull) : "memory");
            } else if constexpr (I0 < RT) {
                asm volatile("s_mov_b64 exec, %[m]\n\tds_read_b128 %[d0], %[a0] offset:%[o0]\n\ts_mov_b64 exec, -1"
                             : [d0] "+v"(ahl[I0]) : [a0] "v"(am_e), [o0] "n"(I0 * 512), [m] "s"(0xFFFFFFFF00000000ull) : "memory");
            }
        };
        auto upm_wait = [&](auto i0_tag) {          // row tiles I0, I0 + 1 carry [h|m]
            constexpr int I0 = decltype(i0_tag)::value;
            constexpr int later = (RT < 8 ? RT : 8) - (I0 + 2);
            // (the fragments are operands: the MFMAs that read them cannot be scheduled above the wait)
            if constexpr (I0 + 1 < RT) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ahl[I0]), "+v"(ahl[I0 + 1]) : "n"((later > 0 ? later : 0) + 4) : "memory");
            else if constexpr (I0 < RT) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ahl[I0]) : "n"((later > 0 ? later : 0) + 4) : "memory");
        };
        using I0T = std::integral_constant<int, 0>; using I2T = std::integral_constant<int, 2>;
        using I4T = std::integral_constant<int, 4>; using I6T = std::integral_constant<int, 6>;
#endif
        ldb(by, st, T_MH);                       // [m|h]
        S3_SB();
#ifndef AVD_LAB_HALFREAD
        mm(ahl, bx, 0, 2); slot(1); S3_SB();     // hl + lh
        mm(ahl, bx, 2, 4); slot(2); S3_SB();
        mm(ahl, bx, 4, 6); slot(3); S3_SB();
        mm(ahl, bx, 6, 8); slot(4); ldb(bx, st, T_HM); S3_SB();       // [h|m] into the dead [l|h] registers
#else
        mm(ahl, bx, 0, 2); slot(1); upm(I0T{}); S3_SB();     // hl + lh
        mm(ahl, bx, 2, 4); slot(2); upm(I2T{}); S3_SB();
        mm(ahl, bx, 4, 6); slot(3); upm(I4T{}); S3_SB();
        mm(ahl, bx, 6, 8); slot(4); upm(I6T{}); ldb(bx, st, T_HM); S3_SB();       // [h|m] into the dead [l|h] registers
        upm_wait(I0T{}); mm(ahm, by, 0, 2); slot(5); S3_SB();     // hm + mh
        upm_wait(I2T{}); mm(ahm, by, 2, 4); slot(6); S3_SB();
        upm_wait(I4T{}); mm(ahm, by, 4, 6); slot(7); S3_SB();
        upm_wait(I6T{}); mm(ahm, by, 6, 8); slot(8); S3_SB();
#endif
#ifndef AVD_LAB_HALFREAD
        mm(ahm, by, 0, 2); slot(5); S3_SB();     // hm + mh
        mm(ahm, by, 2, 4); slot(6); S3_SB();
        mm(ahm, by, 4, 6); slot(7); S3_SB();
        mm(ahm, by, 6, 8); slot(8); S3_SB();
#endif
        mm(ahm, bx, 0, 2); slot(9); S3_SB();     // hh + mm
        mm(ahm, bx, 2, 4); slot(10); S3_SB();
        mm(ahm, bx, 4, 6); slot(11); S3_SB();
        mm(ahm, bx, 6, 8); S3_SB();
    };
    using MainT = std::integral_constant<bool, true>;
    using TailT = std::integral_constant<bool, false>;
    [[maybe_unused]] const unsigned long long t_loop = S3_T();
    {
        // (A variant that read the first group's fragments a step ahead and front-loaded the DMA was built for the 8-wave blocks: it
        // needs ~30 more live registers, spilled lane addresses into the loop — every reload waits on the DMA queue — and measured
        // 212 us against this loop's 195 us on the fc2 shape.  DESIGN.md, negative results.)
        const int n_main = nk - NST + 1 > 0 ? nk - NST + 1 : 0;       // steps that still have a tile to issue
        int kt = 0;
        for (; kt < n_main; ++kt) step(MainT{}, kt);
        for (; kt < nk; ++kt) step(TailT{}, kt);
    }
#undef S3_SB
    [[maybe_unused]] const unsigned long long t_end = S3_T();
    // the epilogue's lane-dependent addresses are derived from this copy: opaque to the compiler, so none of them is hoisted above the
    // K loop (where they cost registers the loop needs; hoisted, they spilled into it)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    if constexpr (TR) {
        s3_epilogue_img16<EPI, RT>(g, acc, (int64_t)bm * BM + wm * WM, bn * BN + wn * WN, lane_e);
    } else {
        __syncthreads();
        constexpr int CLD = WN + 4;
        float* slab = reinterpret_cast<float*>(smem3) + wave * 64 * CLD;
        s3_epilogue16<EPI>(g, acc, slab, (int64_t)bm * BM + wm * WM, bn * BN + wn * WN, lane_e);
    }
#ifdef AVD_S3_STAMPS
    const unsigned long long t_issued = S3_T();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S3_DBG(4, t_issued);
    S3_DBG(0, t_entry); S3_DBG(1, t_loop); S3_DBG(2, t_end); S3_DBG(3, S3_T());
    S3_DBG(8, (unsigned long long)nk); S3_DBG(9, S3_RT()); S3_DBG(10, rt_entry);
    {   // which XCD and CU ran the block, and which tile it was (tools/micro/s3_phase.py replays the stage traffic through an L2 model)
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        S3_DBG(11, (unsigned long long)(xcc & 15u) | ((unsigned long long)hwid << 8));
        S3_DBG(12, (unsigned long long)(unsigned)bm | ((unsigned long long)(unsigned)bn << 32));
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// bf16x3, 16x16x32 MFMA, FOUR waves with a 128 x 128 wave tile (one wave per SIMD, accumulators in the AGPR half of its 512 registers)
// ---------------------------------------------------------------------------------------------------------
// Why: diagnostic builds of the 8-wave kernel above (tools/micro/s3_stamps.py --variant, profiles/r03_s3_stamps.txt) put a price on its
// fragment reads — with them removed fc2 runs 200 us instead of 267 (15 % fewer cycles per step AND 2.39 instead of 2.11 GHz: LDS
// traffic is power) — and 28 ds_read_b128 per 96 MFMAs is the minimum for a 128 x 64 wave tile.  A 128 x 128 wave tile needs 40 reads
// per 192 MFMAs (-29 % LDS bytes per FLOP); its 8 x 8 accumulator tiles are 256 registers, so a SIMD holds ONE wave and nothing hides
// that wave's waits but its own schedule: every fragment a step starts with is read during the step before (the 256 VGPRs beside
// the accumulators hold two A sets and two W sets: [h|l] of tile kt+1 replaces [h|l] of tile kt as soon as the first MFMA group is
// through, the W sets trade roles from step to step), the DMA of tile kt+2 is spread behind the MFMAs of step kt and awaited at the top
// of step kt+1.  Same 256 x 256 (224 x 256) blocks, stage layout and DMA pieces as the 8-wave kernel; image epilogues only.
template <int EPI, int RT>
__global__ __launch_bounds__(256, 1) void gemm_bf16x3_w128_kernel(S3Args g) {
    constexpr int BM = 32 * RT, BN = 256, WM = 16 * RT, PS = 256 * 32, STAGE = 6 * PS, NST = 3, PPW = 12;
    static_assert(EPI == S3_EPI_GELU_SPLIT || EPI == S3_EPI_SPLIT || EPI == S3_EPI_QKV3 || EPI == S3_EPI_RES_IMG, "register image epilogues only");
    static_assert(RT == 8 || RT == 7 || RT == 6, "row tiles per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

    [[maybe_unused]] const unsigned long long t_entry = S3_T(), rt_entry = S3_RT();
    int bm, bn;
    s3_block_of(g, (int)((g.M + BM - 1) / BM), bm, bn);
    if ((int64_t)bm * BM >= g.M) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int nk = g.K >> 4;
    const int nrtA = (int)((g.M + 127) >> 7);

    const int hiq = kq >> 1;
    const int base_e = l15 * 32 + ((kq & 1) << 4), base_o = l15 * 32 + (((kq & 1) ^ 1) << 4);
    const bool flip_a = ((wm * RT) & 1) != 0;
    const int abase_e = flip_a ? base_o : base_e, abase_o = flip_a ? base_e : base_o;
    const int p1 = hiq * PS;
    const int a_uni = wm * (RT * 512), b_uni = 3 * PS + wn * 128 * 32;
    enum { T_HM = 0, T_HL = 1, T_MH = 2, T_LH = 3 };
    auto plane_of = [&](int type) { return type == T_HM ? p1 : type == T_HL ? 2 * p1 : type == T_MH ? PS - p1 : 2 * PS - 2 * p1; };
    auto lda = [&](bf16x8 (&dst)[8], const unsigned char* st, int type, int i0, int i1) {
        const int pl = plane_of(type);
#pragma unroll
        for (int i = i0; i < i1; ++i)
            if (i < RT) dst[i] = *reinterpret_cast<const bf16x8*>(st + a_uni + ((i & 1) ? abase_o : abase_e) + pl + i * 512);
    };
    auto ldb = [&](bf16x8 (&dst)[8], const unsigned char* st, int type, int j0, int j1) {
        const int pl = plane_of(type);
#pragma unroll
        for (int j = j0; j < j1; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(st + b_uni + ((j & 1) ? base_o : base_e) + pl + j * 512);
    };

    // DMA pieces as in the 8-wave kernel: piece P = 12 wave + r of 48; P < 24: A plane P / 8, rows 32 (P % 8) ..; else the same over W
    const unsigned char* rbase[PPW];
    int rdst[PPW];
#pragma unroll
    for (int r = 0; r < PPW; ++r) {
        const int P = wave * PPW + r, isw = P >= 24, pq = isw ? P - 24 : P, pl = pq >> 3, q = pq & 7;
        rdst[r] = (isw ? 3 * PS : 0) + pl * PS + q * 1024;
        if (!isw) {
            const int64_t grow = (int64_t)bm * BM + 32 * q;
            int64_t grp = grow >> 7;
            grp = grp < nrtA ? grp : nrtA - 1;
            rbase[r] = g.A + grp * nk * S3_CHUNK + pl * S3_PLANE + (int)((grow >> 5) & 3) * 1024;
        } else {
            rbase[r] = g.W + (int64_t)(bn * 2 + (q >> 2)) * nk * S3_CHUNK + pl * S3_PLANE + (q & 3) * 1024;
        }
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    // (A pieces of the 32-row groups past the block's 32 RT rows are never read: not moved)
    auto issue_piece = [&](int i, int kt, int buf) {
        const int P = wave * PPW + i;       // wave-uniform
        if (RT < 8 && P < 24 && (P & 7) >= RT) return;
        __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(rbase[i] + (int64_t)kt * S3_CHUNK + lane16), AVD_LDS_PTR(smem3 + (buf * STAGE + rdst[i])), 16, 0, 0);
    };

    // The MFMAs of this kernel are asm statements whose accumulator operand is tied to an AGPR ("+a"): left to itself the compiler keeps
    // the 256 accumulator registers in VGPRs and uses the AGPR half of the wave's 512 registers as spill space (two v_accvgpr moves per
    // accumulator register per MFMA).  A tile is touched again 63 MFMAs later, fragments are written by ds_read only (the compiler's
    // s_waitcnt covers asm operands), and the epilogue's first accumulator read sits behind explicit s_nops.
    f32x4t acc[2][8][4];          // [64-column half][row tile][column tile]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4t{0.f, 0.f, 0.f, 0.f};
    // The compiler does not see an MFMA in the asm statements below, so its hazard recogniser pads neither the zero-init
    // (v_accvgpr_write) -> first MFMA (accumulator read) pair nor the last MFMA -> epilogue (v_accvgpr_read) pair.  Both are made
    // explicit: every tile is named as an operand of an empty statement (its init is ordered in front of it), then one s_nop
    // statement — volatile asm statements keep their order, so the wait states sit between all the inits and the first MFMA.
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc[h][i][j]));
    asm volatile("s_nop 4");
    // transposed accumulators (W fragment as the MFMA's first operand): the image epilogues want one output row per lane
    // FENCE (the last MFMA group of a TAIL step): the wait states that let the MFMAs' AGPR writes land ride in the SAME asm
    // statement as the group's last MFMA — whatever the register allocator puts at the loop's exit (it moves accumulator tiles between
    // registers there when it likes, and pads nothing: the MFMAs are opaque to its hazard recogniser) comes behind them
    // (a plain bool, folded after inlining: asm operands inside a GENERIC lambda do not capture — clang)
    auto mm = [&](const bf16x8 (&A_)[8], const bf16x8 (&B_)[8], int i0, int i1, const bool FENCE) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
            if (i < RT) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (FENCE && i == RT - 1 && j == 7)
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+a"(acc[j >> 2][i][j & 3]) : "v"(B_[j]), "v"(A_[i]));
                    else
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j >> 2][i][j & 3]) : "v"(B_[j]), "v"(A_[i]));
                }
            }
    };

#pragma unroll
    for (int t = 0; t < NST - 1; ++t)
        if (t < nk) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) issue_piece(i, t, t);
        }
    int st_cur = 0, st_fill = NST - 1;
#define S3_SB() __builtin_amdgcn_sched_barrier(0)
    bf16x8 ahl[8], ahm[8], bset0[8], bset1[8];
    // step kt: bp = [l|h] of W, read a step ago; bq is free.  Groups of 8 MFMAs (one row tile x 8 column tiles); behind each group a
    // DMA piece and / or two or three fragment reads
    auto step = [&](auto main_tag, int kt, bf16x8 (&bp)[8], bf16x8 (&bq)[8]) {
        constexpr bool MAIN = decltype(main_tag)::value;      // main loop: tile kt + 2 exists, its DMA is unconditional
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // tile kt + 1 has landed, tile kt - 1 is no longer read
        const unsigned char* st = smem3 + st_cur * STAGE;
        const int dbuf = st_fill;
        st_cur = st_cur + 1 == NST ? 0 : st_cur + 1;
        st_fill = st_fill + 1 == NST ? 0 : st_fill + 1;
        const unsigned char* stn = smem3 + st_cur * STAGE;           // stage of tile kt + 1
        const bool more = MAIN || kt + NST - 1 < nk;
        auto slot = [&](int s) {
#ifndef AVD_LAB_NODMA
            if (s < PPW) {
                if (MAIN || more) { S3_SB(); issue_piece(s, kt + NST - 1, dbuf); S3_SB(); }
            }
#endif
        };
        S3_SB();
        // hl + lh on fragments read during the previous step; meanwhile [m|h] of W and [h|m] of A arrive
        mm(ahl, bp, 0, 1, false); slot(0); ldb(bq, st, T_MH, 0, 4); lda(ahm, st, T_HM, 0, 1); S3_SB();
        mm(ahl, bp, 1, 2, false); slot(1); ldb(bq, st, T_MH, 4, 8); lda(ahm, st, T_HM, 1, 2); S3_SB();
        mm(ahl, bp, 2, 3, false); slot(2); lda(ahm, st, T_HM, 2, 4); S3_SB();
        mm(ahl, bp, 3, 4, false); slot(3); lda(ahm, st, T_HM, 4, 5); S3_SB();
        mm(ahl, bp, 4, 5, false); slot(4); lda(ahm, st, T_HM, 5, 6); S3_SB();
        mm(ahl, bp, 5, 6, false); slot(5); lda(ahm, st, T_HM, 6, 7); S3_SB();
        mm(ahl, bp, 6, 7, false); slot(6); lda(ahm, st, T_HM, 7, 8); S3_SB();
        mm(ahl, bp, 7, 8, false); slot(7); S3_SB();
        // hm + mh; [h|m] of W goes into the dead [l|h] registers, the next tile's [h|l] of A into the dead [h|l] registers
        mm(ahm, bq, 0, 1, false); slot(8); ldb(bp, st, T_HM, 0, 4); S3_SB();
        mm(ahm, bq, 1, 2, false); slot(9); ldb(bp, st, T_HM, 4, 8); S3_SB();
        mm(ahm, bq, 2, 3, false); slot(10); lda(ahl, stn, T_HL, 0, 2); S3_SB();
        mm(ahm, bq, 3, 4, false); slot(11); lda(ahl, stn, T_HL, 2, 4); S3_SB();
        mm(ahm, bq, 4, 5, false); lda(ahl, stn, T_HL, 4, 6); S3_SB();
        mm(ahm, bq, 5, 6, false); lda(ahl, stn, T_HL, 6, 8); S3_SB();
        mm(ahm, bq, 6, 7, false); S3_SB();
        mm(ahm, bq, 7, 8, false); S3_SB();
        // hh + mm; the next tile's [l|h] of W into the dead [m|h] registers
        mm(ahm, bp, 0, 1, false); ldb(bq, stn, T_LH, 0, 4); S3_SB();
        mm(ahm, bp, 1, 2, false); ldb(bq, stn, T_LH, 4, 8); S3_SB();
        mm(ahm, bp, 2, 3, !MAIN); S3_SB();
        mm(ahm, bp, 3, 4, !MAIN); S3_SB();
        mm(ahm, bp, 4, 5, !MAIN); S3_SB();
        mm(ahm, bp, 5, 6, !MAIN); S3_SB();
        mm(ahm, bp, 6, 7, !MAIN); S3_SB();
        mm(ahm, bp, 7, 8, !MAIN); S3_SB();
    };
    [[maybe_unused]] const unsigned long long t_loop = S3_T();
    {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // tiles 0 and 1 landed
        lda(ahl, smem3, T_HL, 0, 8);
        ldb(bset0, smem3, T_LH, 0, 8);
        using MainT = std::integral_constant<bool, true>;
        using TailT = std::integral_constant<bool, false>;
        int kt = 0;
        for (; kt + 3 < nk; kt += 2) {
            step(MainT{}, kt, bset0, bset1);
            step(MainT{}, kt + 1, bset1, bset0);
        }
        for (; kt + 1 < nk; kt += 2) {
            step(TailT{}, kt, bset0, bset1);
            step(TailT{}, kt + 1, bset1, bset0);
        }
        if (kt < nk) step(TailT{}, kt, bset0, bset1);
    }
#undef S3_SB
    [[maybe_unused]] const unsigned long long t_end = S3_T();
    int lane_e = lane;
    // the last MFMAs have written their AGPRs before the epilogue reads them: the wait states are one statement with the tiles of the
    // last two MFMA groups (row tiles RT - 2 and RT - 1) as read-write operands, so no accumulator read of the epilogue can be scheduled
    // in front of it (MFMAs retire in order: older tiles are complete once these are)
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(lane_e), "+a"(acc[0][RT - 2][0]), "+a"(acc[0][RT - 2][1]), "+a"(acc[0][RT - 2][2]), "+a"(acc[0][RT - 2][3]),
                   "+a"(acc[1][RT - 2][0]), "+a"(acc[1][RT - 2][1]), "+a"(acc[1][RT - 2][2]), "+a"(acc[1][RT - 2][3]), "+a"(acc[0][RT - 1][0]),
                   "+a"(acc[0][RT - 1][1]), "+a"(acc[0][RT - 1][2]), "+a"(acc[0][RT - 1][3]), "+a"(acc[1][RT - 1][0]), "+a"(acc[1][RT - 1][1]),
                   "+a"(acc[1][RT - 1][2]), "+a"(acc[1][RT - 1][3]));
    // (the other row tiles were last written >= 16 MFMAs = 256 issue cycles before the loop's final sched_barrier, above which nothing of the
    // epilogue can be scheduled: they need no wait states.  Naming all 64 tiles here as well made the register allocator spill.)
    s3_epilogue_img16<EPI, RT>(g, acc[0], (int64_t)bm * BM + wm * WM, bn * BN + wn * 128, lane_e);
    s3_epilogue_img16<EPI, RT>(g, acc[1], (int64_t)bm * BM + wm * WM, bn * BN + wn * 128 + 64, lane_e);
#ifdef AVD_S3_STAMPS
    const unsigned long long t_issued = S3_T();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S3_DBG(4, t_issued);
    S3_DBG(0, t_entry); S3_DBG(1, t_loop); S3_DBG(2, t_end); S3_DBG(3, S3_T());
    S3_DBG(8, (unsigned long long)nk); S3_DBG(9, S3_RT()); S3_DBG(10, rt_entry);
#endif
}

// (The f16x2 row-owner GEMM — EPI_RES_NORM — was built in the same shape too: four waves side by side over the 128 x 512 block, 4 x 4
// tiles of v_mfma_f32_32x32x16_f16 in AGPRs, 16 fragment reads per 48 MFMAs instead of 12 per 24.  Bit-identical and slower: 155 us
// against 136 us per launch, 164.9 against 176.3 steps/s.  An f16x2 step is 1,536 matrix-pipe cycles per wave, half of bf16x3's, so the
// per-step barrier and the epilogue's memory latency — which a second wave on the SIMD hides — weigh twice as much.  Removed.)

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
int64_t split3_bytes(int64_t rows, int K) { return ((rows + 255) / 256 * 256) * (int64_t)K * 6; }
// algorithmic bytes per element of a producer that reads fp32 and writes an operand image: 4 + three bf16 planes, or 4 + the two
// fp16 planes of an f16x2 image (the third plane is never written)
static double image_rw_bytes(float h2_scale) { return h2_scale > 0.f ? 8.0 : 10.0; }

int split3_f32(const float* x, int64_t ld, void* out, int64_t rows, int K, hipStream_t st, float h2_scale) {
    return split3_rows_f32(x, RowMap{ld, 0, 0}, out, rows, K, st, h2_scale);
}

int split3_rows_f32(const float* x, RowMap xm, void* out, int64_t rows, int K, hipStream_t st, float h2_scale, float* ss) {
    AVD_REQUIRE(x && out, AVD_EINVAL, "split3: null pointer");
    AVD_REQUIRE(!ss || K % 64 == 0, AVD_EUNSUPPORTED, "split3: sums of squares need K %% 64 == 0 (K=%d)", K);
    const int64_t ld = xm.ld;
    AVD_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ld >= K && ld % 4 == 0 && xm.stride % 4 == 0, AVD_EUNSUPPORTED,
                "split3: need rows > 0, K %% 16 == 0, ld %% 4 == 0 (rows=%lld K=%d ld=%lld)", (long long)rows, K, (long long)ld);
    AVD_REQUIRE(aligned16(x) && aligned16(out), AVD_EUNSUPPORTED, "split3: pointers must be 16-byte aligned");
    const int64_t rows_pad = (rows + 255) / 256 * 256;
    const int64_t n = rows_pad * (K / 8);
    AVD_REQUIRE((n + 255) / 256 < (1ll << 31), AVD_EUNSUPPORTED, "split3: grid too large");
    static const int tag = prof_tag_id("split3_kernel");
    AVD_REQUIRE(h2_scale >= 0.f && h2_scale < __builtin_inff(), AVD_EINVAL, "split3: f16x2 image scale must be positive and finite");
    ProfScope prof(tag, (double)rows * K * image_rw_bytes(h2_scale), st);
    if (h2_scale > 0.f)
        hipLaunchKernelGGL(split3_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xm,
                           static_cast<unsigned char*>(out), rows, rows_pad, K, h2_scale, ss);
    else
        hipLaunchKernelGGL(split3_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xm,
                           static_cast<unsigned char*>(out), rows, rows_pad, K, 0.f, ss);
    AVD_CHECK_LAUNCH("split3");
    return AVD_OK;
}

int rmsnorm_split3_f32(const float* x, const float* scale, void* out, int64_t rows, int d, float eps, hipStream_t st, float h2_scale) {
    AVD_REQUIRE(x && scale && out, AVD_EINVAL, "rmsnorm_split3: null pointer");
    AVD_REQUIRE(rows > 0 && d > 0 && d % 16 == 0 && d <= 2048, AVD_EUNSUPPORTED, "rmsnorm_split3: d=%d must be a multiple of 16, <= 2048", d);
    AVD_REQUIRE(aligned16(x) && aligned16(out), AVD_EUNSUPPORTED, "rmsnorm_split3: pointers must be 16-byte aligned");
    static const int tag = prof_tag_id("rmsnorm_split3_kernel");
    ProfScope prof(tag, image_rw_bytes(h2_scale) * (double)rows * d, st);
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const float isd = (float)sqrt((double)d);
    unsigned char* o = static_cast<unsigned char*>(out);
    AVD_REQUIRE(h2_scale >= 0.f && h2_scale < __builtin_inff(), AVD_EINVAL, "rmsnorm_split3: f16x2 image scale must be positive and finite");
#define AVD_RMS3(NC)                                                                                                                  \
    if (h2_scale > 0.f) hipLaunchKernelGGL((rmsnorm_split3_kernel<NC, true>), dim3(grid), dim3(256), 0, st, x, scale, o, rows, d, eps, isd, h2_scale); \
    else hipLaunchKernelGGL((rmsnorm_split3_kernel<NC, false>), dim3(grid), dim3(256), 0, st, x, scale, o, rows, d, eps, isd, 0.f)
    switch ((d + 511) / 512) {
        case 1: AVD_RMS3(1); break;
        case 2: AVD_RMS3(2); break;
        case 3: AVD_RMS3(3); break;
        default: AVD_RMS3(4); break;
    }
#undef AVD_RMS3
    AVD_CHECK_LAUNCH("rmsnorm_split3");
    return AVD_OK;
}

int layernorm_act_split3_f32(const float* x, const float* gamma, const float* beta, void* out, int64_t rows, int d, float eps, int act,
                             hipStream_t st, float h2_scale) {
    AVD_REQUIRE(x && gamma && beta && out, AVD_EINVAL, "layernorm_split3: null pointer");
    AVD_REQUIRE(rows > 0 && d > 0 && d % 16 == 0 && d <= 2048, AVD_EUNSUPPORTED, "layernorm_split3: d=%d must be a multiple of 16, <= 2048", d);
    AVD_REQUIRE(aligned16(x) && aligned16(out) && aligned16(gamma) && aligned16(beta), AVD_EUNSUPPORTED, "layernorm_split3: pointers must be 16-byte aligned");
    AVD_REQUIRE(h2_scale >= 0.f && h2_scale < __builtin_inff(), AVD_EINVAL, "layernorm_split3: f16x2 image scale must be positive and finite");
    static const int tag = prof_tag_id("layernorm_act_split3_kernel");
    ProfScope prof(tag, image_rw_bytes(h2_scale) * (double)rows * d, st);
    const unsigned grid = (unsigned)((rows + 3) / 4);
    unsigned char* o = static_cast<unsigned char*>(out);
#define AVD_LN3K(NC, F16, G) hipLaunchKernelGGL((layernorm_act_split3_kernel<NC, F16, G>), dim3(grid), dim3(256), 0, st, x, gamma, beta, o, rows, d, eps, act, h2_scale)
#define AVD_LN3(NC)                                                                            \
    if (h2_scale > 0.f) { if (act == AVD_ACT_GELU) AVD_LN3K(NC, true, true); else AVD_LN3K(NC, true, false); } \
    else { if (act == AVD_ACT_GELU) AVD_LN3K(NC, false, true); else AVD_LN3K(NC, false, false); }
    switch ((d + 511) / 512) {
        case 1: AVD_LN3(1); break;
        case 2: AVD_LN3(2); break;
        case 3: AVD_LN3(3); break;
        default: AVD_LN3(4); break;
    }
#undef AVD_LN3
#undef AVD_LN3K
    AVD_CHECK_LAUNCH("layernorm_split3");
    return AVD_OK;
}

bool gemm_bf16x3_supported(int64_t M, int N, int K) { return M > 0 && N > 0 && N % 256 == 0 && K > 0 && K % 16 == 0; }
// residual + RMSNorm epilogue (a block owns whole rows): 512 columns, f16x2 images
bool gemm_bf16x3_rownorm_supported(int N, int terms) { return N == 512 && terms == 3; }

// tile configuration: 0 = 256x256, 8 waves, one block per CU; 1 = 256x128, 4 waves, two blocks per CU.
// AVD_S3_TILE=0|1 forces one (measurement aid, also avd_tune_set "s3_tile"); default: per epilogue, what measured faster in the C3 pipeline.
int g_s3_tile = getenv("AVD_S3_TILE") ? atoi(getenv("AVD_S3_TILE")) : -1;
static int s3_tile_for(int epi, int64_t M, int N) {
    if (g_s3_tile == 0 || g_s3_tile == 1) return g_s3_tile;
    if (epi == S3_EPI_GELU_SPLIT || epi == S3_EPI_QKV3 || epi == S3_EPI_SPLIT) return 1;
    // 256x256 tiles only when they occupy most of the 256 CUs (C3: 106 x 2 = 212 blocks); at the 128x128 geometry
    // (8,512 rows) the 256x128 tiles run the whole step in 3.75 ms against 4.78
    return (M + 255) / 256 * (N / 256) >= 192 ? 0 : 1;
}

// Stagger of the two co-resident blocks of the 4-wave kernel, x 1024 cycles; -1 = automatic: about a third of a tile's life when
// the launch has at least two generations of blocks and the caller is not already running two kernel chains on two streams (those
// drift apart by themselves; there the start-up delay only costs).  avd_tune_set "s3_stagger".
int g_s3_stagger = getenv("AVD_S3_STAGGER") ? atoi(getenv("AVD_S3_STAGGER")) : -1;
thread_local bool t_s3_two_streams = false;      // set by avd_denoise_step_f32 around its two-stream section
#ifdef AVD_S3_STAMPS
unsigned long long* g_s3_dbg = nullptr;
extern "C" void lab_set_dbg(unsigned long long* p) { g_s3_dbg = p; }
#endif

static int s3_cu_count() {                 // CUs of the current device (looked up once per device)
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int n = dev < 64 ? cache[dev].load(std::memory_order_acquire) : 0;
    if (!n) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n = prop.multiProcessorCount;
        if (dev < 64) cache[dev].store(n, std::memory_order_release);
    }
    return n;
}

// Super-tiles: the blocks an XCD runs together (consecutive ids after the XCD remap) form sm x sn blocks that share sm A panels and sn W
// panels.  256x128 tiles (tile = true, two blocks per CU): rounds 2-4 ran whole block rows per super-tile (the 12-16 column blocks that
// share an A panel together, 2 rows x all columns); 16 blocks per super-tile for the one-block-per-CU tiles.
// AVD_S3_SN / AVD_S3_SUPER4 / AVD_S3_SUPER8 (avd_tune_set "s3_sn" / "s3_super4" / "s3_super8"): measurement aids — super-tile width in blocks /
// blocks per super-tile of the two-per-CU and one-per-CU kernels; profiles/r05_fetch_ab.txt holds the FETCH_SIZE of fc1 / in_proj across them.
int g_s3_sn = getenv("AVD_S3_SN") ? atoi(getenv("AVD_S3_SN")) : 0;                  // avd_tune_set "s3_sn": 0 = the rule below
int g_s3_super4 = getenv("AVD_S3_SUPER4") ? atoi(getenv("AVD_S3_SUPER4")) : 0;      // avd_tune_set "s3_super4" / "s3_super8": 0 = 32 / 16 blocks
int g_s3_super8 = getenv("AVD_S3_SUPER8") ? atoi(getenv("AVD_S3_SUPER8")) : 0;
static void s3_supertile(bool tile, int nbn, int& sn_out, int& sm_out) {
    int sn = 8;
    while (nbn % sn) sn >>= 1;
    if (tile && nbn <= 16) sn = nbn;
    int total = tile ? 32 : 16;
    // round 5 (profiles/r05_supertile_ab.txt): two-per-CU kernels with 8 or more column blocks run 16 x 4 super-tiles — the XCD keeps only FOUR
    // W panels live (1.5 MB of its 4 MB L2 at K = 512; they are re-read by every row block and now stay resident) and streams each A panel
    // past them once per column group: fc1 294 -> 278 us, in_proj 226 -> 208 us per launch at C3, nothing at 8,512 rows
    if (tile && nbn >= 8 && nbn % 4 == 0) { sn = 4; total = 64; }
    if (g_s3_sn > 0) { sn = g_s3_sn < nbn ? g_s3_sn : nbn; while (nbn % sn) --sn; }
    if ((tile ? g_s3_super4 : g_s3_super8) > 0) total = tile ? g_s3_super4 : g_s3_super8;
    sn_out = sn;
    sm_out = total / sn > 0 ? total / sn : 1;
}

template <int EPI, int TERMS, int WAVES>
static int launch_s3w(S3Args g, hipStream_t st, int nz = 1) {
    using Cf = S3Cfg<TERMS, WAVES, EPI == S3_EPI_RES_NORM>;
    constexpr bool tile = WAVES == 4;
    static LdsAttr attr;
    auto kern = gemm_bf16x3_kernel<EPI, TERMS, WAVES>;
    if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), Cf::LDS, "gemm_bf16x3")) return rc;
#ifdef AVD_S3_STAMPS
    g.dbg = g_s3_dbg;
#endif
    g.nbn = g.N / Cf::BN;
    s3_supertile(tile, g.nbn, g.sn, g.sm);
    const int64_t nbm = (g.M + Cf::BM - 1) / Cf::BM;
    const int64_t nwg = nbm * g.nbn;
    AVD_REQUIRE(nwg < (1ll << 31), AVD_EUNSUPPORTED, "gemm_bf16x3 grid too large");
    g.stagger = 0;
    g.first_gen = 2 * s3_cu_count();
    if (tile && g.first_gen > 0 && nbm * g.nbn >= 2 * g.first_gen)
        g.stagger = g_s3_stagger >= 0 ? g_s3_stagger : t_s3_two_streams ? 0 : (TERMS == 3 ? 24 : TERMS == 1 ? 12 : 48) * (g.K >= 1024 ? 2 : 1);
    // tag = the kernel name as rocprofv3 prints its template arguments (EPI, TERMS, WAVES)
    static const int tag = prof_tag_id("gemm_bf16x3_kernel<%d, %d, %d>", EPI, TERMS, WAVES);
    ProfScope prof(tag, 2.0 * (double)g.M * g.N * g.K * nz, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)nz), dim3(WAVES * 64), Cf::LDS, st, g);
    AVD_CHECK_LAUNCH("gemm_bf16x3");
    return AVD_OK;
}

// bf16x3 (six terms) on the 16x16x32 MFMA with two terms per instruction; avd_tune_set "s3_m16" 0 takes the 32x32x16 kernel instead
int g_s3_m16 = getenv("AVD_S3_M16") ? atoi(getenv("AVD_S3_M16")) : 1;
// rows per 8-wave block of the residual + image epilogue: 0 = automatic (224 when that saves a generation of blocks), 7 / 8 forced
// (avd_tune_set "s3_rt", AVD_S3_RT)
int g_s3_rt = getenv("AVD_S3_RT") ? atoi(getenv("AVD_S3_RT")) : 0;
template <int EPI, int WAVES, int RT = 8, int NSTK = 0>
static int launch_s3w16(S3Args g, hipStream_t st) {
    using Cf = S3Cfg<6, WAVES>;
    constexpr bool tile = WAVES == 4;
    constexpr int BM = WAVES == 8 || RT < 8 ? 32 * RT : Cf::BM;
    constexpr int LDS = NSTK ? NSTK * Cf::STAGE : Cf::LDS;
    static LdsAttr attr;
    auto kern = gemm_bf16x3_m16_kernel<EPI, WAVES, RT, NSTK>;
    if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), LDS, "gemm_bf16x3 (16x16x32)")) return rc;
#ifdef AVD_S3_STAMPS
    g.dbg = g_s3_dbg;
#endif
    g.nbn = g.N / Cf::BN;
    s3_supertile(tile, g.nbn, g.sn, g.sm);
    const int64_t nbm = (g.M + BM - 1) / BM;
    const int64_t nwg = nbm * g.nbn;
    AVD_REQUIRE(nwg < (1ll << 31), AVD_EUNSUPPORTED, "gemm_bf16x3 grid too large");
    g.stagger = 0;
    g.first_gen = 2 * s3_cu_count();
    if (tile && g.first_gen > 0 && nbm * g.nbn >= 2 * g.first_gen)
        g.stagger = g_s3_stagger >= 0 ? g_s3_stagger : t_s3_two_streams ? 0 : 48 * (g.K >= 1024 ? 2 : 1);
    // tag = the kernel name as rocprofv3 prints it (a defaulted NSTK = 0 is printed too)
    static const int tag = prof_tag_id("gemm_bf16x3_m16_kernel<%d, %d, %d, %d>", EPI, WAVES, RT, NSTK);
    ProfScope prof(tag, 2.0 * (double)g.M * g.N * g.K, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WAVES * 64), LDS, st, g);
    AVD_CHECK_LAUNCH("gemm_bf16x3 (16x16x32)");
    return AVD_OK;
}

// 4 waves with a 128 x 128 wave tile (gemm_bf16x3_w128_kernel) in place of the 8-wave blocks of the image epilogues (by default that is the
// residual + image epilogue of out_proj / fc2: 190 -> 180 us per launch at C3; fc1 / in_proj keep two 4-wave blocks per CU — with one
// wave per SIMD nothing runs beside their GELU / split epilogue: 389 against 292 us).  avd_tune_set "s3_w128" 0 takes the 8-wave kernel.
int g_s3_w128 = getenv("AVD_S3_W128") ? atoi(getenv("AVD_S3_W128")) : 1;
template <int EPI, int RT>
static int launch_s3w128(S3Args g, hipStream_t st) {
    constexpr int BM = 32 * RT, LDS = 3 * 6 * 256 * 32;
    static LdsAttr attr;
    auto kern = gemm_bf16x3_w128_kernel<EPI, RT>;
    if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), LDS, "gemm_bf16x3 (128x128 wave tile)")) return rc;
#ifdef AVD_S3_STAMPS
    g.dbg = g_s3_dbg;
#endif
    g.nbn = g.N / 256;
    s3_supertile(false, g.nbn, g.sn, g.sm);
    const int64_t nbm = (g.M + BM - 1) / BM;
    const int64_t nwg = nbm * g.nbn;
    AVD_REQUIRE(nwg < (1ll << 31), AVD_EUNSUPPORTED, "gemm_bf16x3 grid too large");
    g.stagger = 0;
    g.first_gen = 0;
    static const int tag = prof_tag_id("gemm_bf16x3_w128_kernel<%d, %d>", EPI, RT);
    ProfScope prof(tag, 2.0 * (double)g.M * g.N * g.K, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), LDS, st, g);
    AVD_CHECK_LAUNCH("gemm_bf16x3 (128x128 wave tile)");
    return AVD_OK;
}

// Rows per 4-wave block (32 RT) of the image epilogues: 0 = automatic, 2 .. 8 forced (avd_tune_set "s3_rt4", AVD_S3_RT4; in_proj / fc1
// have no blocks shorter than 5 row tiles and take 5 for less).
// Automatic: the CU with the most blocks sets the launch's time, ceil(blocks / CUs) blocks of RT row tiles each, and a block carries
// about three row tiles' worth of work that does not shrink with it (W tile DMA, barriers, the W fragment reads; fitted to the
// residual launches at 3,904 rows: 71.4 / 60.8 / 56.7 / 45.1 us for RT = 5 / 4 / 3 / 2): RT minimises ceil(blocks / CUs) x (RT + 3),
// and 8 stays unless a shorter block models at least 6 % better — many generations of blocks
// (C3: 26,944 rows) keep 256-row blocks, whose W traffic and fragment reads per MFMA are the lowest.
int g_s3_rt4 = getenv("AVD_S3_RT4") ? atoi(getenv("AVD_S3_RT4")) : 0;
static int s3_rt4_for(int64_t M, int N, int rt_min) {
    if (g_s3_rt4 >= 2 && g_s3_rt4 <= 8) return g_s3_rt4 > rt_min ? g_s3_rt4 : rt_min;
    const int64_t cu = s3_cu_count() > 0 ? s3_cu_count() : 256, nbn = N / 128;
    int best = 8;
    int64_t cost8 = 0, cbest = 0;
    for (int rt = 8; rt >= rt_min; --rt) {
        const int64_t blocks = (M + 32 * rt - 1) / (32 * rt) * nbn, cost = (blocks + cu - 1) / cu * (rt + 3);
        if (rt == 8) cost8 = cbest = cost;
        else if (cost < cbest) { cbest = cost; best = rt; }
    }
    return cbest * 100 <= cost8 * 94 ? best : 8;
}
// The four-stage ring (one block per CU) of the residual + image launches: 1 (default) when the launch's blocks fit the CUs once,
// 0 never (avd_tune_set "s3_deep4", AVD_S3_DEEP4)
int g_s3_deep4 = getenv("AVD_S3_DEEP4") ? atoi(getenv("AVD_S3_DEEP4")) : 1;
static bool s3_deep4_for(int64_t M, int N, int rt) {
    const int64_t cu = s3_cu_count() > 0 ? s3_cu_count() : 256;
    return g_s3_deep4 != 0 && (M + 32 * rt - 1) / (32 * rt) * (N / 128) <= cu;
}

template <int EPI, int TERMS>
static int launch_s3t(const S3Args& a, hipStream_t st) {
    if constexpr (TERMS == 6 && EPI != S3_EPI_RES_NORM) {
        if (g_s3_m16) {
            if (s3_tile_for(EPI, a.M, a.N)) {
                // the noise head's launches (fp32 out with a bias: shared Linears, out_proj; image out: input_proj) when their blocks fit the
                // CUs once — short blocks, one per CU, on the four-stage ring; EPI_BIAS then runs from the registers (S3_EPI_BIAS_REG)
                if constexpr (EPI == S3_EPI_BIAS || EPI == S3_EPI_SPLIT) {
                    constexpr int E2 = EPI == S3_EPI_BIAS ? (int)S3_EPI_BIAS_REG : (int)S3_EPI_SPLIT;
                    const int rt = s3_rt4_for(a.M, a.N, 2);
                    if (a.bias && a.N % 128 == 0 && s3_deep4_for(a.M, a.N, rt)) {
                        switch (rt) {
                            case 2: return launch_s3w16<E2, 4, 2, 4>(a, st);
                            case 3: return launch_s3w16<E2, 4, 3, 4>(a, st);
                            case 4: return launch_s3w16<E2, 4, 4, 4>(a, st);
                            case 5: return launch_s3w16<E2, 4, 5, 4>(a, st);
                            case 6: return launch_s3w16<E2, 4, 6, 4>(a, st);
                            case 7: return launch_s3w16<E2, 4, 7, 4>(a, st);
                            default: return launch_s3w16<E2, 4, 8, 4>(a, st);
                        }
                    }
                }
                if constexpr (EPI == S3_EPI_RES_IMG) {
                    const int rt = s3_rt4_for(a.M, a.N, 2);
                    if (s3_deep4_for(a.M, a.N, rt)) {
                        switch (rt) {
                            case 2: return launch_s3w16<EPI, 4, 2, 4>(a, st);
                            case 3: return launch_s3w16<EPI, 4, 3, 4>(a, st);
                            case 4: return launch_s3w16<EPI, 4, 4, 4>(a, st);
                            case 5: return launch_s3w16<EPI, 4, 5, 4>(a, st);
                            case 6: return launch_s3w16<EPI, 4, 6, 4>(a, st);
                            case 7: return launch_s3w16<EPI, 4, 7, 4>(a, st);
                            default: return launch_s3w16<EPI, 4, 8, 4>(a, st);
                        }
                    }
                }
                // (in_proj / fc1 on the four-stage ring when their blocks fit the CUs once: measured, no gain — K = 512 is 32 steps, and at
                // 3,904 rows in_proj's 252 blocks run 62 us two to a CU and 79 us one to a CU; they keep the two-stage ring)
                if constexpr (EPI == S3_EPI_RES_IMG) {
                    switch (s3_rt4_for(a.M, a.N, 2)) {
                        case 2: return launch_s3w16<EPI, 4, 2>(a, st);
                        case 3: return launch_s3w16<EPI, 4, 3>(a, st);
                        case 4: return launch_s3w16<EPI, 4, 4>(a, st);
                        default: break;
                    }
                }
                if constexpr (EPI == S3_EPI_GELU_SPLIT || EPI == S3_EPI_QKV3 || EPI == S3_EPI_RES_IMG) {
                    switch (s3_rt4_for(a.M, a.N, 5)) {
                        case 5: return launch_s3w16<EPI, 4, 5>(a, st);
                        case 6: return launch_s3w16<EPI, 4, 6>(a, st);
                        case 7: return launch_s3w16<EPI, 4, 7>(a, st);
                        default: break;
                    }
                }
                return launch_s3w16<EPI, 4>(a, st);
            }
            constexpr bool TR = EPI == S3_EPI_GELU_SPLIT || EPI == S3_EPI_SPLIT || EPI == S3_EPI_QKV3 || EPI == S3_EPI_RES_IMG;
            bool rt7 = false, rt6 = false;
            if constexpr (EPI == S3_EPI_RES_IMG) {
                // 224-row (192-row) blocks when they need fewer generations of blocks x rows than 256-row blocks (one block per CU)
                const int64_t cu = s3_cu_count() > 0 ? s3_cu_count() : 256, nbn = a.N / 256;
                const int64_t g8 = ((a.M + 255) / 256 * nbn + cu - 1) / cu * 8, g7 = ((a.M + 223) / 224 * nbn + cu - 1) / cu * 7;
                const int64_t g6 = ((a.M + 191) / 192 * nbn + cu - 1) / cu * 6;
                rt6 = g_s3_w128 && (g_s3_rt == 6 || (g_s3_rt == 0 && g6 < g7 && g6 < g8));       // 192 rows: the four-wave kernel only
                rt7 = !rt6 && (g_s3_rt == 7 || (g_s3_rt != 8 && g7 < g8));
            }
            if constexpr (TR) {
                if (g_s3_w128) {
                    if constexpr (EPI == S3_EPI_RES_IMG) {
                        if (rt6) return launch_s3w128<EPI, 6>(a, st);
                    }
                    return rt7 ? launch_s3w128<EPI, 7>(a, st) : launch_s3w128<EPI, 8>(a, st);
                }
            }
            if constexpr (EPI == S3_EPI_RES_IMG) {
                if (rt7) return launch_s3w16<EPI, 8, 7>(a, st);
            }
            return launch_s3w16<EPI, 8>(a, st);
        }
    }
    if constexpr (EPI == S3_EPI_RES_NORM) {
        if constexpr (TERMS == 3) return launch_s3w<EPI, 3, 8>(a, st);
        else AVD_REQUIRE(false, AVD_EUNSUPPORTED, "gemm_bf16x3: the residual + RMSNorm epilogue exists for f16x2 images only");
    } else {
        return s3_tile_for(EPI, a.M, a.N) ? launch_s3w<EPI, TERMS, 4>(a, st) : launch_s3w<EPI, TERMS, 8>(a, st);
    }
}

template <int EPI>
static int launch_s3(const S3Args& a, hipStream_t st) {
    switch (a.terms) {
        case 0: case 6: return launch_s3t<EPI, 6>(a, st);
        case 9: return launch_s3t<EPI, 9>(a, st);
        case 1: return launch_s3t<EPI, 1>(a, st);
        case 3: return launch_s3t<EPI, 3>(a, st);
        default: return set_error(AVD_EINVAL, "gemm_bf16x3: terms must be 6 (default), 9 (strict), 1 (plain bf16) or 3 (f16x2), got %d", a.terms);
    }
}

// C = act(A W^T + bias) (+ residual).  C3 != null: the output is written as a split3 image (act must be GELU);
// otherwise fp32 row-major into C (act NONE; residual optional, may alias C).
// Folded RMSNorm (see S3Args): ss_in -> image outputs scale their rows by 1 / (||row of the un-normalised A|| / sqrt(K) + eps);
// C, C3, R and ss_out all given -> the new residual stream as fp32, as an image and as sums of squares in one epilogue.
bool gemm_bf16x3_resmap_supported(int terms) { return (terms == 0 || terms == 6) && g_s3_m16 != 0; }

int gemm_bf16x3(const void* A3, const void* W3, const float* bias, const float* R, float* C, void* C3, int64_t M, int N, int K,
                int act, int terms, hipStream_t st, float ab_scale, float c_scale, const float* ss_in, float eps, float* ss_out,
                const float* gamma, int r_seg, int r_stride) {
    AVD_REQUIRE(r_seg == 0 || (r_seg > 0 && r_stride >= r_seg && ss_out && gemm_bf16x3_resmap_supported(terms) && M < (1ll << 31)), AVD_EUNSUPPORTED,
                "gemm_bf16x3: a residual row map needs the fp32 + image residual epilogue of the six-term 16x16x32 kernels");
    AVD_REQUIRE(A3 && W3 && (C || C3), AVD_EINVAL, "gemm_bf16x3: null pointer");
    AVD_REQUIRE(!gamma || (gemm_bf16x3_rownorm_supported(N, terms) && C && C3 && R && bias && !ss_in && !ss_out && act == AVD_ACT_NONE && aligned16(gamma)),
                AVD_EUNSUPPORTED, "gemm_bf16x3: the residual + RMSNorm epilogue needs N == 512, f16x2 images, fp32 and image outputs, bias and residual");
    AVD_REQUIRE(!ss_in || (C3 && !C && K % 64 == 0), AVD_EUNSUPPORTED, "gemm_bf16x3: a folded norm needs an image output and K %% 64 == 0");
    AVD_REQUIRE(!ss_out || (C && C3 && R && bias && N % 64 == 0 && act == AVD_ACT_NONE && terms != 3), AVD_EUNSUPPORTED,
                "gemm_bf16x3: sums of squares are written by the fp32 + image residual epilogue only (bf16 planes)");
    AVD_REQUIRE(ab_scale > 0.f && ab_scale < __builtin_inff() && c_scale > 0.f && c_scale < __builtin_inff(), AVD_EINVAL,
                "gemm_bf16x3: image scales must be positive and finite");
    AVD_REQUIRE(gemm_bf16x3_supported(M, N, K), AVD_EUNSUPPORTED, "gemm_bf16x3: need N %% 256 == 0 and K %% 16 == 0 (M=%lld N=%d K=%d)",
                (long long)M, N, K);
    AVD_REQUIRE(aligned16(A3) && aligned16(W3) && aligned16(C) && aligned16(C3) && aligned16(bias) && aligned16(R), AVD_EUNSUPPORTED,
                "gemm_bf16x3: pointers must be 16-byte aligned");
    S3Args a{static_cast<const unsigned char*>(A3), static_cast<const unsigned char*>(W3), bias, R, C,
             static_cast<unsigned char*>(C3), M, N, K, 0, 0, 0, 0, 0, 0, 0.f, 0, 0, terms, 1.0f / ab_scale, c_scale,
             ss_in, ss_out, (float)sqrt((double)K), eps, gamma, r_seg, r_stride};
    if (gamma) {
        a.ss_sqrt_d = (float)sqrt((double)N);
        return launch_s3<S3_EPI_RES_NORM>(a, st);
    }
    if (C && C3) {
        AVD_REQUIRE(ss_out, AVD_EUNSUPPORTED, "gemm_bf16x3: fp32 and image output together imply the residual + sums-of-squares epilogue");
        return launch_s3<S3_EPI_RES_IMG>(a, st);
    }
    if (C3) {
        AVD_REQUIRE((act == AVD_ACT_GELU || act == AVD_ACT_NONE) && !R && bias, AVD_EUNSUPPORTED,
                    "gemm_bf16x3: image output implies bias, act NONE or GELU, no residual");
        return act == AVD_ACT_GELU ? launch_s3<S3_EPI_GELU_SPLIT>(a, st) : launch_s3<S3_EPI_SPLIT>(a, st);
    }
    AVD_REQUIRE(act == AVD_ACT_NONE, AVD_EUNSUPPORTED, "gemm_bf16x3: fp32 output supports act NONE only");
    if (R) {
        // six-term 16x16x32 kernels: the residual + image epilogue with no image and no sums of squares to write IS the fp32 residual
        // epilogue ((acc + bias) + R, straight from the accumulator registers) — and it has the short-block / deep-ring variants the
        // slab epilogue lacks (the trimmed last fc2 at the shipped 128 x 128 geometry: 6,144 rows, 96 blocks of 256 x 128)
        if ((terms == 0 || terms == 6) && g_s3_m16 && bias) return launch_s3<S3_EPI_RES_IMG>(a, st);
        return launch_s3<S3_EPI_RES>(a, st);
    }
    return launch_s3<S3_EPI_BIAS>(a, st);
}

// ---- split-K for launches that cannot fill the chip (fc2 of the small configurations: 3,904 rows x 512 columns are 64 blocks) ----
// Slice z of NS multiplies k in [z K / NS, (z + 1) K / NS) and stores its fp32 partial sums; this kernel adds them in slice order, then
// bias and residual as the residual epilogues do, and writes what they write: the fp32 stream and (IMG) its operand image and the
// rows' sums of squares per 64-column chunk.  Deterministic: fixed order, no atomics.  One thread per 8 columns.
template <bool IMG>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ns, const float* __restrict__ bias,
                                                            const float* R, float* C, unsigned char* __restrict__ C3,
                                                            float* __restrict__ ss, int64_t M, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = N >> 3;
    const int64_t m = i / per_row;
    const int n = (int)(i % per_row) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (m < M) {
        const int64_t o = m * N + n;
        *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(part + o);
        *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(part + o + 4);
        for (int z = 1; z < ns; ++z) {
            const float* pz = part + (int64_t)z * M * N + o;
            const f32x4 a = *reinterpret_cast<const f32x4*>(pz), b = *reinterpret_cast<const f32x4*>(pz + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += a[e]; v[4 + e] += b[e]; }
        }
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + n), b1 = *reinterpret_cast<const f32x4*>(bias + n + 4);
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(R + o), r1 = *reinterpret_cast<const f32x4*>(R + o + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = (v[e] + b0[e]) + r0[e]; v[4 + e] = (v[4 + e] + b1[e]) + r1[e]; }
        *reinterpret_cast<f32x4*>(C + o) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(C + o + 4) = f32x4{v[4], v[5], v[6], v[7]};
        if constexpr (IMG) store_split8<true>(C3, m, n, N, v);
    }
    if constexpr (IMG) {      // 8 consecutive threads hold one 64-column chunk of one row (N % 64 == 0)
        float q = ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
        q += __shfl_xor(q, 1, 64);
        q += __shfl_xor(q, 2, 64);
        q += __shfl_xor(q, 4, 64);
        if ((threadIdx.x & 7) == 0 && m < M) ss[m * (N >> 6) + (n >> 6)] = q;
    }
}

// K slices for a residual GEMM of [M][N] over K, 0 = do not split.  A launch whose 256 x 128 blocks cover at most half of the CUs is cut
// into the largest power-of-two number of K slices (<= "s3_splitk", default 4) that still leaves at most ONE block per CU and a slice of
// >= 32 k-steps: C2 (3,904 rows) fc2 64 blocks -> 4 slices; 6,736 rows 108 blocks -> 2.  (Round 4 tried filling the two slots per CU as
// well — 8,512 rows, 136 blocks x 2 slices = 272 blocks: the 16 CUs that get two blocks set the launch time, 130 + 21 us against 150 us
// unsplit, profiles/r04_bench_128_splitk.txt — so a CU never gets more than one.)  The slices run on the 32x32x16 kernel (one-term and
// nine-term modes, and the six-term mode when "s3_m16" is 0).  avd_tune_set "s3_splitk" (AVD_S3_SPLITK): 0 off, else the largest slice count tried.
int g_s3_splitk = [] { const int v = getenv("AVD_S3_SPLITK") ? atoi(getenv("AVD_S3_SPLITK")) : 4; return v < 0 ? 0 : v > kS3SplitKMax ? kS3SplitKMax : v; }();
static bool splitk_shape_ok(int64_t M, int N, int K) {
    if (N % 128 || K % 32 || K / 2 < 512) return false;
    return (M + 255) / 256 * (N / 128) * 2 <= (int64_t)s3_cu_count();
}
int gemm_bf16x3_splitk_slices(int64_t M, int N, int K, int terms) {
    // (six-term mode on the 16x16x32 kernels: no slices — at 6,736 rows two slices did not pay, profiles/r04_bench_128_splitk.txt, and a
    // plan that changes with the row count would break the bit-identity of the one- and two-stream CFG layouts at the bench's size)
    if (g_s3_splitk < 2 || terms == 3 || ((terms == 0 || terms == 6) && g_s3_m16) || !splitk_shape_ok(M, N, K)) return 0;
    const int64_t cu = s3_cu_count(), blocks = (M + 255) / 256 * (N / 128);
    int ns = 1;
    while (ns * 2 <= g_s3_splitk && blocks * ns * 2 <= cu && K % (32 * ns) == 0 && K / (ns * 2) >= 512) ns *= 2;
    return ns >= 2 ? ns : 0;
}
int64_t gemm_bf16x3_splitk_ws_floats(int64_t M, int N, int ns) { return (int64_t)ns * M * N; }
// the most a split-K launch of this shape can ever ask for, whatever the tunables say (s3_splitk <= kS3SplitKMax): sizing only
int64_t gemm_bf16x3_splitk_ws_max_floats(int64_t M, int N, int K) {
    return splitk_shape_ok(M, N, K) ? gemm_bf16x3_splitk_ws_floats(M, N, kS3SplitKMax) : 0;
}

// C = A W^T + bias + R (fp32), optionally C3 = operand image of C and ss = its rows' sums of squares; bf16-plane images
int gemm_bf16x3_splitk(const void* A3, const void* W3, const float* bias, const float* R, float* C, void* C3, float* ss, int64_t M, int N,
                       int K, int terms, int ns, float* part, hipStream_t st) {
    AVD_REQUIRE(A3 && W3 && bias && R && C && part && ns >= 2, AVD_EINVAL, "gemm_bf16x3_splitk: null pointer");
    AVD_REQUIRE((C3 != nullptr) == (ss != nullptr), AVD_EINVAL, "gemm_bf16x3_splitk: the image and the sums of squares come together");
    AVD_REQUIRE(K % (16 * ns) == 0 && N % 128 == 0 && N % 64 == 0 && gemm_bf16x3_supported(M, N, K / ns), AVD_EUNSUPPORTED,
                "gemm_bf16x3_splitk: shape (M=%lld N=%d K=%d slices=%d)", (long long)M, N, K, ns);
    AVD_REQUIRE(aligned16(A3) && aligned16(W3) && aligned16(C) && aligned16(C3) && aligned16(bias) && aligned16(R) && aligned16(part),
                AVD_EUNSUPPORTED, "gemm_bf16x3_splitk: pointers must be 16-byte aligned");
    S3Args a{static_cast<const unsigned char*>(A3), static_cast<const unsigned char*>(W3), nullptr, nullptr, part, nullptr, M, N, K / ns,
             0, 0, 0, 0, 0, 0, 0.f, 0, 0, terms, 1.0f, 1.0f, nullptr, nullptr, 1.0f, 0.f, nullptr, 0, 0};
    int rc;
    switch (terms) {
        case 1: rc = launch_s3w<S3_EPI_BIAS, 1, 4>(a, st, ns); break;
        case 9: rc = launch_s3w<S3_EPI_BIAS, 9, 4>(a, st, ns); break;
        case 0: case 6: rc = launch_s3w<S3_EPI_BIAS, 6, 4>(a, st, ns); break;
        default: AVD_REQUIRE(false, AVD_EUNSUPPORTED, "gemm_bf16x3_splitk: bf16-plane modes only (terms %d)", terms);
    }
    if (rc) return rc;
    const int64_t threads = M * (N >> 3);
    static const int tag = prof_tag_id("splitk_reduce_kernel");
    ProfScope prof(tag, (double)M * N * 4.0 * (ns + 2) + (C3 ? (double)M * N * 6.0 : 0.0), st);
    if (C3)
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, part, ns, bias, R, C,
                           static_cast<unsigned char*>(C3), ss, M, N);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, part, ns, bias, R, C,
                           nullptr, nullptr, M, N);
    AVD_CHECK_LAUNCH("splitk_reduce");
    return AVD_OK;
}

// in_proj for the bf16x3 attention: qkv = A W^T + bias written as the qkv3 image (q pre-multiplied by qscale)
int gemm_bf16x3_qkv3(const void* A3, const void* W3, const float* bias, void* img, int64_t M, int tokens, int heads, int K, float qscale,
                     int terms, hipStream_t st, float ab_scale, float c_scale, const float* ss_in, float eps) {
    AVD_REQUIRE(!ss_in || K % 64 == 0, AVD_EUNSUPPORTED, "gemm_bf16x3_qkv3: a folded norm needs K %% 64 == 0");
    AVD_REQUIRE(A3 && W3 && bias && img, AVD_EINVAL, "gemm_bf16x3_qkv3: null pointer");
    AVD_REQUIRE(ab_scale > 0.f && ab_scale < __builtin_inff() && c_scale > 0.f && c_scale < __builtin_inff(), AVD_EINVAL,
                "gemm_bf16x3_qkv3: image scales must be positive and finite");
    const int N = 3 * heads * 64;
    AVD_REQUIRE(tokens > 0 && heads > 0 && M > 0 && M % tokens == 0, AVD_EINVAL, "gemm_bf16x3_qkv3: rows %lld not a multiple of tokens %d",
                (long long)M, tokens);
    AVD_REQUIRE(M < (1ll << 31) - 256, AVD_EUNSUPPORTED, "gemm_bf16x3_qkv3: more than 2^31 rows");
    AVD_REQUIRE(gemm_bf16x3_supported(M, N, K), AVD_EUNSUPPORTED, "gemm_bf16x3_qkv3: need 3*heads*64 %% 256 == 0 and K %% 16 == 0");
    AVD_REQUIRE(aligned16(A3) && aligned16(W3) && aligned16(bias) && aligned16(img), AVD_EUNSUPPORTED, "gemm_bf16x3_qkv3: alignment");
    S3Args a{static_cast<const unsigned char*>(A3), static_cast<const unsigned char*>(W3), bias, nullptr, nullptr,
             static_cast<unsigned char*>(img), M, N, K, 0, 0, 0, tokens, qkv3_npad(tokens), heads, qscale, 0, 0, terms, 1.0f / ab_scale, c_scale,
             ss_in, nullptr, (float)sqrt((double)K), eps, nullptr, 0, 0};
    return launch_s3<S3_EPI_QKV3>(a, st);
}

}  // namespace avd

using namespace avd;

extern "C" int64_t avd_split3_bytes(int64_t rows, int K) {
    if (rows <= 0 || K <= 0 || K % 16) return -1;
    return split3_bytes(rows, K);
}
extern "C" int avd_split3_f32(const float* x, void* out, int64_t rows, int K, avd_stream_t stream) {
    return split3_f32(x, K, out, rows, K, static_cast<hipStream_t>(stream));
}
extern "C" int avd_rmsnorm_split3_f32(const float* x, const float* scale, void* out, int64_t rows, int d, float eps,
                                      avd_stream_t stream) {
    return rmsnorm_split3_f32(x, scale, out, rows, d, eps, static_cast<hipStream_t>(stream));
}
extern "C" int avd_gemm_bf16x3_f32(const void* A3, const void* W3, const float* bias, const float* residual, float* C, void* C3,
                                   int64_t M, int N, int K, int act, int terms, avd_stream_t stream) {
    return gemm_bf16x3(A3, W3, bias, residual, C, C3, M, N, K, act, terms, static_cast<hipStream_t>(stream));
}
extern "C" int avd_gemm_bf16x3_qkv3_f32(const void* A3, const void* W3, const float* bias, void* qkv3, int64_t M, int tokens, int heads,
                                        int K, float qscale, int terms, avd_stream_t stream) {
    return gemm_bf16x3_qkv3(A3, W3, bias, qkv3, M, tokens, heads, K, qscale, terms, static_cast<hipStream_t>(stream));
}

// f16x2 mode (two fp16 planes, three product terms; avd_common.h): the same images with a caller-chosen power-of-two scale
extern "C" int avd_split_f16x2_f32(const float* x, void* out, int64_t rows, int K, float scale, avd_stream_t stream) {
    AVD_REQUIRE(scale > 0.f, AVD_EINVAL, "avd_split_f16x2_f32: scale must be positive");
    return split3_f32(x, K, out, rows, K, static_cast<hipStream_t>(stream), scale);
}
extern "C" int avd_rmsnorm_split_f16x2_f32(const float* x, const float* gamma, void* out, int64_t rows, int d, float eps, float scale,
                                           avd_stream_t stream) {
    AVD_REQUIRE(scale > 0.f, AVD_EINVAL, "avd_rmsnorm_split_f16x2_f32: scale must be positive");
    return rmsnorm_split3_f32(x, gamma, out, rows, d, eps, static_cast<hipStream_t>(stream), scale);
}
extern "C" int avd_gemm_f16x2_f32(const void* A2, const void* W2, const float* bias, const float* residual, float* C, void* C2, int64_t M,
                                  int N, int K, int act, float ab_scale, float c_scale, avd_stream_t stream) {
    return gemm_bf16x3(A2, W2, bias, residual, C, C2, M, N, K, act, 3, static_cast<hipStream_t>(stream), ab_scale, c_scale);
}
extern "C" int avd_gemm_f16x2_qkv_f32(const void* A2, const void* W2, const float* bias, void* qkv, int64_t M, int tokens, int heads, int K,
                                      float qscale, float ab_scale, float qkv_scale, avd_stream_t stream) {
    return gemm_bf16x3_qkv3(A2, W2, bias, qkv, M, tokens, heads, K, qscale, 3, static_cast<hipStream_t>(stream), ab_scale, qkv_scale);
}
