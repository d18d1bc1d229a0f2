// Pieces of the split-operand GEMMs shared by gemm_bf16x3.hip and mlp_bf16x3.hip: the launch arguments, the epilogue kinds, the
// counted-vmcnt wait and the register-only image epilogue of the 16x16x32 kernels (layouts and rationale: gemm_bf16x3.hip).
#pragma once
#include "avd_common.h"

namespace avd {

#define AVD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AVD_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 5: bias, image out, no activation.  6: bias + residual, fp32 out AND its operand image AND the rows' sums of squares (the
// producer side of a folded RMSNorm)
// 7: bias + residual, fp32 out AND the operand image of RMSNorm(out) — the block owns whole rows (128 x 512 block, N == 512), so the
// norm that follows the residual add (mmdt.py:97-98 -> 39-42) is finished inside the epilogue; f16x2 images (a normalised row has the
// bound its image scale needs, the un-normalised stream has none)
enum { S3_EPI_BIAS = 0, S3_EPI_RES = 2, S3_EPI_GELU_SPLIT = 3, S3_EPI_QKV3 = 4, S3_EPI_SPLIT = 5, S3_EPI_RES_IMG = 6, S3_EPI_RES_NORM = 7,
       S3_EPI_BIAS_REG = 8 };   // EPI_BIAS (fp32 out = acc + bias) run from the accumulator registers: the 16x16x32 kernels' short-block variants

struct S3Args {
    const unsigned char* A;   // split3 image of [M][K]
    const unsigned char* W;   // split3 image of [N][K]
    const float* bias;
    const float* R;           // residual [M][N] (may alias C)
    float* C;                 // [M][N] fp32 (EPI_BIAS, EPI_RES)
    unsigned char* C3;        // split3 image of [M][N] (EPI_GELU_SPLIT)
    int64_t M;
    int N, K, nbn, sm, sn;
    int tokN, tokNpad, heads;   // EPI_QKV3: tokens per sample, padded tokens per sample, heads (N == 3 * heads * 64)
    float qscale;               // EPI_QKV3: factor folded into q before it is split (softmax scale * log2 e)
    int stagger;                // 4-wave kernel: the block in the odd wave slots of its SIMDs starts stagger x 1024 cycles late
    int first_gen;              // ... if it belongs to the first generation of blocks (blockIdx < 2 x CUs of the device)
    int terms;                  // 6 (default), 9 (strict), 1 (plain bf16 operands) or 3 (f16x2 images)
    float ab_inv, c_scale;      // terms 3: 1 / (A image scale x W image scale) applied to the sums; scale of the image written
    // RMSNorm folding (mmdt.py:39-42 moved into its neighbours, as gemm_f32.hip does on the fp32 path).
    // Consumer (image epilogues): ss_in[row][K / 64] = sums of squares of the UN-normalised rows whose image is A; W carries the norm's
    // scale; every output row is multiplied by 1 / (sqrt(sum_c ss_in[row][c]) / ss_sqrt_d + ss_eps) before the bias.
    // Producer (EPI_RES_IMG): ss_out[row][N / 64] receives the sums of squares of the rows it writes.
    const float* ss_in;
    float* ss_out;
    float ss_sqrt_d, ss_eps;
    const float* gamma;         // EPI_RES_NORM: the norm's scale vector [N]; ss_sqrt_d = sqrt(N), ss_eps as above
    // EPI_RES_IMG on the 16x16x32 kernels: residual rows in groups — output row m adds R row (m / r_seg) * r_stride + m % r_seg
    // (r_seg == 0: R row m).  The last block of the core runs on its target rows only: outputs compact, residual stream not.
    int r_seg, r_stride;
#ifdef AVD_S3_STAMPS            // diagnostic build only (tools/micro/s3_stamps.py), never in the product library
    unsigned long long* dbg;
#endif
};

#ifdef AVD_S3_STAMPS
// every block stamps (core clock and 100 MHz real time) its entry, loop start, loop end and exit: four s_memtime per block,
// none inside the K loop (stamps inside the loop drain lgkmcnt and change what they measure)
#define S3_T() __builtin_amdgcn_s_memtime()
#define S3_RT() __builtin_amdgcn_s_memrealtime()
#define S3_DBG(i, v) do { if (threadIdx.x == 0) g.dbg[(size_t)blockIdx.x * 16 + (i)] = (v); } while (0)
#else
#define S3_T() 0ull
#define S3_RT() 0ull
#define S3_DBG(i, v) do { } while (0)
#endif

// Which tile a block computes.  (1) XCD-contiguous remap: the hardware deals consecutive block ids round-robin over the 8 XCDs; after the
// remap every XCD — its own L2 — works through one contiguous range of `wg`.  (2) `wg` walks super-tiles of sm x sn blocks (the blocks an
// XCD runs together share sm A panels and sn W panels), super-rows of sm block rows top to bottom; the LAST super-row holds the nbm % sm
// rows that are left, so the grid is exactly nbm x nbn blocks (round 5: a grid padded to whole super-rows put all its empty blocks on
// the last XCD — 96 of its 224 slots at C3 with 16-row super-tiles — and the other seven XCDs carried 6 % more work each).
__device__ __forceinline__ void s3_block_of(const S3Args& g, int nbm, int& bm, int& bn) {
    const int b = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
    const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    const int per_row = g.sm * g.nbn;
    const int srow = wg / per_row, rem = wg - srow * per_row;
    const int left = nbm - srow * g.sm, h = left < g.sm ? left : g.sm;      // block rows of this super-row
    const int per_st = h * g.sn;
    const int sc = rem / per_st, rem2 = rem - sc * per_st;
    bm = srow * g.sm + rem2 / g.sn;
    bn = sc * g.sn + rem2 % g.sn;
}

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

#ifdef AVD_LAB_NOSTORE     // diagnostic build: the epilogue's arithmetic runs, (almost) nothing is stored
#define S3_ROW_OK(m) ((m) < g.M && __float_as_uint(v[0]) == 0x7fc12345u)
#else
#define S3_ROW_OK(m) ((m) < g.M)
#endif

typedef float f32x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4t mma16x16(bf16x8 a, bf16x8 b, f32x4t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// image epilogues from TRANSPOSED accumulator tiles (see s3_epilogue_img_t for the 32 x 32 version and for `big`)
// RT = row tiles of the wave that are live (8, or 7 in the 224-row blocks: tile 7 is all zeros and its rows belong to the next wave)
template <int EPI, int RT = 8>
__device__ __forceinline__ void s3_epilogue_img16(const S3Args& g, f32x4t (&acc)[8][4], int64_t mwave0, int nbase, int lane) {
    const int l15 = lane & 15, kq = lane >> 4;
    // range test over the valid rows (lane (l15, kq) of row tile i holds output row 16 i + l15 before the exchange)
    bool big = EPI == S3_EPI_RES_IMG;
    if constexpr (EPI != S3_EPI_RES_IMG && EPI != S3_EPI_BIAS_REG) {
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) a = fmaxf(a, fabsf(acc[i][j][r]));
            if (mwave0 + 16 * i + l15 < g.M) amax = fmaxf(amax, a);
        }
        float bmax = 0.f;
#pragma unroll
        for (int e = 0; e < 64; e += 4) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + nbase + e);
            bmax = fmaxf(fmaxf(bmax, fmaxf(fabsf(b4[0]), fabsf(b4[1]))), fmaxf(fabsf(b4[2]), fabsf(b4[3])));
        }
        big = __any(!(amax < 1.2676506e30f) || !(bmax < 1.2676506e30f) || !(fabsf(g.qscale) < 1.0e6f));
    }
    // bias of this lane's chunk of every column tile: columns nbase + 16 j + 8 (kq >> 1) + (0..7)
    float bv[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* bp = g.bias + nbase + 16 * j + 8 * (kq >> 1);
        *reinterpret_cast<f32x4*>(bv[j]) = *reinterpret_cast<const f32x4*>(bp);
        *reinterpret_cast<f32x4*>(bv[j] + 4) = *reinterpret_cast<const f32x4*>(bp + 4);
    }
    [[maybe_unused]] float mul = 1.0f;
    [[maybe_unused]] int64_t qbase = 0;
    if constexpr (EPI == S3_EPI_QKV3) {
        const int dmodel = g.heads * 64;
        const int part = nbase / dmodel, head = (nbase % dmodel) >> 6;
        mul = part == 0 ? g.qscale : 1.0f;
        qbase = (((int64_t)part * (g.M / g.tokN)) * g.heads + head) * (int64_t)g.tokNpad * QKV3_ROWB;
    }
#pragma unroll
    for (int ip = 0; ip < 4; ++ip) {         // pairs of row tiles (2 ip, 2 ip + 1): this lane ends up with a row of tile 2 ip + (kq & 1)
        const int tile = 2 * ip + (kq & 1);
        const int64_t m = (RT == 8 || tile < RT) ? mwave0 + 16 * tile + l15 : g.M;      // a dead tile's rows fail every m < M test below
        float rinv = 1.0f;
        if (EPI != S3_EPI_RES_IMG && EPI != S3_EPI_BIAS_REG && g.ss_in != nullptr && m < g.M) {
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
            const unsigned b = (unsigned)m / (unsigned)g.tokN, tok = (unsigned)m - b * (unsigned)g.tokN;
            qrow = g.C3 + qbase + ((int64_t)b * g.heads * g.tokNpad + tok) * QKV3_ROWB;
            qsw = qkv3_swizzle(nbase / (g.heads * 64), (int)tok);
        }
        [[maybe_unused]] float ssq = 0.f;
        [[maybe_unused]] int64_t rrow = m;        // EPI_RES_IMG: row of the residual operand
        if constexpr (EPI == S3_EPI_RES_IMG) {
            if (g.r_seg > 0 && m < g.M) {
                const unsigned sgi = (unsigned)m / (unsigned)g.r_seg;
                rrow = (int64_t)sgi * g.r_stride + ((unsigned)m - sgi * (unsigned)g.r_seg);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * ip][j][r]), __float_as_uint(acc[2 * ip + 1][j][r]), false, false);
                v[r] = __uint_as_float(sw[0]);
                v[4 + r] = __uint_as_float(sw[1]);
            }
            const int n = nbase + 16 * j + 8 * (kq >> 1);
            if constexpr (EPI == S3_EPI_QKV3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], rinv, bv[j][e]) * mul;
                if (S3_ROW_OK(m)) {
                    const int c = 2 * j + (kq >> 1);
                    unsigned char* dst = qrow + ((c ^ qsw) << 4);
                    u32x4 Hh, Mi, Lo;
                    if (big) split8<true>(v, Hh, Mi, Lo);
                    else split8<false>(v, Hh, Mi, Lo);
                    *reinterpret_cast<u32x4*>(dst) = Hh;
                    *reinterpret_cast<u32x4*>(dst + 128) = Mi;
                    *reinterpret_cast<u32x4*>(dst + 256) = Lo;
                }
            } else if constexpr (EPI == S3_EPI_BIAS_REG) {
                if (S3_ROW_OK(m)) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += bv[j][e];
                    float* cp = g.C + m * g.N + n;
                    *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                }
            } else if constexpr (EPI == S3_EPI_RES_IMG) {
                if (S3_ROW_OK(m)) {
                    const float* rp = g.R + rrow * g.N + n;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (v[e] + bv[j][e]) + (e < 4 ? r0[e] : r1[e - 4]);
                    float* cp = g.C + m * g.N + n;
                    *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<f32x4*>(cp + 4) = f32x4{v[4], v[5], v[6], v[7]};
                    ssq += ((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])) + ((v[4] * v[4] + v[5] * v[5]) + (v[6] * v[6] + v[7] * v[7]));
                    if (g.C3) store_split8<true>(g.C3, m, n, g.N, v);      // (the fused MLP's last layer writes the fp32 stream only)
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = fmaf(v[e], rinv, bv[j][e]);      // explicit: every kernel that finishes a folded norm rounds alike
                    v[e] = EPI == S3_EPI_GELU_SPLIT ? gelu_erf(t) : t;
                }
                if (S3_ROW_OK(m)) {
                    if (big) store_split8<true>(g.C3, m, n, g.N, v);
                    else store_split8<false>(g.C3, m, n, g.N, v);
                }
            }
        }
        if constexpr (EPI == S3_EPI_RES_IMG) {
            // lanes kq and kq ^ 2 hold the two 8-column halves of the same row's 16-column groups: one exchange, lanes kq < 2 write
            const float tot = ssq + __shfl_xor(ssq, 32, 64);
            if (kq < 2 && m < g.M && g.ss_out) g.ss_out[m * (g.N >> 6) + (nbase >> 6)] = tot;
        }
    }
}

}  // namespace avd
