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
// the 16 bytes of (plane p, row r, half = (k%16)/8) sit at p*4096 + r*32 + (half ^ ((r>>3)&1))*16.  One block's K-tile of
// an operand is then contiguous, already bank-swizzled pieces — the global->LDS DMA copies whole 1 KiB pieces (only the planes the
// mode uses), the ds_read_b128 of fragment rows is conflict free — and a producer's store of one plane for consecutive rows is
// contiguous too.  Producers write the image directly (rmsnorm_split3_kernel and layernorm_act_split3_kernel here, the attention
// epilogue in attn_bf16x3.hip, the bias / bias+GELU epilogues below), so no fp32 copy of those activations exists in these modes;
// the in_proj epilogue writes the "qkv3" image the attention kernel reads (layout in avd_common.h).
//
// Kernels (wave tile 128x64 = 4x2 accumulators, K-tile 16, XCD-contiguous super-tiles, LDS stage = the planes moved, compact):
//   gemm_bf16x3_kernel    256x256, 8 waves, one block per CU, 3 / 4 / 6 stages (3 / 2 / 1 planes), counted vmcnt;
//   gemm_bf16x3_b_kernel  256x128, 4 waves, two blocks per CU, 2 / 3 / 4 stages, fragments one K-tile ahead in registers —
//                         for the heavy epilogues (image outputs) and for batches that do not fill 256-row tiles.
// What caps the rate is power: bf16x3 holds 2.07 GHz at 1.33 kW, f16x2 1.91 GHz at the 1.4 kW cap (DESIGN.md 4.5 / 4.6;
// tools/micro/s3_stamps.py shows the K loops at 79-91 % of the matrix pipe's issue rate at that clock).
#include "avd_common.h"

#include <stdlib.h>

namespace avd {

#define AVD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AVD_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---------------------------------------------------------------------------------------------------------
// producers
// ---------------------------------------------------------------------------------------------------------
// x [rows][K] (row stride ld) -> split3 image; rows in [rows, rows_pad) are written as zeros
template <bool F16>   // F16: f16x2 image with scale s (avd_common.h)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, RowMap xm, unsigned char* __restrict__ out,
                                                     int64_t rows, int64_t rows_pad, int K, float s) {
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
enum { S3_EPI_BIAS = 0, S3_EPI_RES = 2, S3_EPI_GELU_SPLIT = 3, S3_EPI_QKV3 = 4, S3_EPI_SPLIT = 5 };   // 5: bias, image out, no activation

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
    int stagger;                // 256x128 kernel: the block in the odd wave slots of its SIMDs starts stagger x 1024 cycles late
    int first_gen;              // ... if it belongs to the first generation of blocks (blockIdx < 2 x CUs of the device)
    int terms;                  // 6 (default), 9 (strict), 1 (plain bf16 operands) or 3 (f16x2 images)
    float ab_inv, c_scale;      // terms 3: 1 / (A image scale x W image scale) applied to the sums; scale of the image written
    // stream-K (sk_partial != null): the grid is one resident block per slot; XCD x owns a contiguous range of tiles and its
    // blocks split that range's (tile, K-step) sequence evenly.  A block whose range ends inside a tile parks its accumulators
    // in sk_partial[blockIdx] and raises sk_flags[blockIdx]; the block whose range reaches the tile's end adds the parked
    // parts in ascending block order (fixed, so results are deterministic), lowers the flags and runs the epilogue.
    float* sk_partial;
    unsigned int* sk_flags;
    int ntiles;
    int64_t sk_floats;          // host only: floats available at sk_partial
#ifdef AVD_S3_STAMPS            // diagnostic build only (tools/micro/s3_stamps.py), never in the product library
    unsigned long long* dbg;
#endif
};

#ifdef AVD_S3_STAMPS
// wave 0 of every block sums, over its K loop, the core-clock cycles between the loop top, the DMA wait, the barrier, the DMA
// issue and the end of the step; plus stamps at block entry, loop start, loop end, block end
#define S3_T() __builtin_amdgcn_s_memtime()
#define S3_RT() __builtin_amdgcn_s_memrealtime()
#define S3_DBG(i, v) do { if (threadIdx.x == 0) g.dbg[(size_t)blockIdx.x * 16 + (i)] = (v); } while (0)
#else
#define S3_T() 0ull
#define S3_RT() 0ull
#define S3_DBG(i, v) do { } while (0)
#endif

// flags of the stream-K launches: zero at module load, every launch leaves its slot zeroed (each flag has one consumer)
constexpr int S3_SK_SLOTS = 32, S3_SK_FLAGS = 512;
__device__ unsigned int g_s3_sk_flags[S3_SK_SLOTS * S3_SK_FLAGS];

// segment of work of a stream-K block: tile (bm, bn), K-steps [k0, k1)
struct SkRange {
    int steps_lo, steps_hi, tile_base, J, j, S;
    __device__ __forceinline__ int hi_of(int jj) const { return (int)(((int64_t)(jj + 1) * S) / J); }
};
__device__ __forceinline__ SkRange sk_range(int ntiles, int nk) {
    SkRange r;
    const int x = blockIdx.x & 7;
    r.j = blockIdx.x >> 3;
    r.J = gridDim.x >> 3;
    const int q = ntiles >> 3, rem = ntiles & 7;
    r.tile_base = x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q;
    r.S = (q + (x < rem ? 1 : 0)) * nk;
    r.steps_lo = (int)(((int64_t)r.j * r.S) / r.J);
    r.steps_hi = r.hi_of(r.j);
    return r;
}

// Segments of a block's step range [lo, hi), in the order they are PROCESSED: first the part that starts a tile it cannot
// finish (parked and published at once, so the block that finishes that tile never waits long), then the whole tiles, last
// the part that finishes a tile an earlier block started (by then that block's flag has long been raised).  Processing order
// does not touch the summation order of any tile.  phase: 0 head, 1 whole tiles, 2 tail, 3 done; s = first step of the segment.
__device__ __forceinline__ bool sk_next(const SkRange& r, int nk, int& phase, int& s, int& k0, int& k1) {
    const int lo = r.steps_lo, hi = r.steps_hi;
    if (lo >= hi) return false;
    const int beg_full = (lo + nk - 1) / nk * nk;       // end of the partial tail (== lo when lo starts a tile)
    const int end_full = hi / nk * nk;                  // start of the partial head (== hi when hi ends a tile)
    if (beg_full > hi) {                                 // the whole range lies inside one tile and reaches neither end
        if (phase != 0) return false;
        phase = 3; s = lo; k0 = lo % nk; k1 = k0 + (hi - lo);
        return true;
    }
    if (phase == 0) {
        phase = 1;
        s = beg_full - nk;                               // whole tiles start at beg_full (s is advanced before use)
        if (hi > end_full) { s = end_full; k0 = 0; k1 = hi - end_full; phase = -1; return true; }
    }
    if (phase == -1) { phase = 1; s = beg_full - nk; }
    if (phase == 1) {
        s += nk;
        if (s + nk <= end_full && s >= beg_full) { k0 = 0; k1 = nk; return true; }
        phase = 2;
    }
    if (phase == 2) {
        phase = 3;
        if (beg_full > lo) { s = lo; k0 = lo % nk; k1 = nk; return true; }
    }
    return false;
}

// contributor side: all stores of the block retired, then one agent-scope release and the flag (cdna guide, Guideline 16)
__device__ __forceinline__ void sk_publish(unsigned int* flag, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// owner side: one lane polls (bounded: a block only ever waits on LOWER block ids of its own XCD, which are dispatched first),
// one agent-scope acquire, barrier, then the whole block may read the parked data with plain loads
__device__ __forceinline__ void sk_await(unsigned int* flag, int tid) {
    if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && spins < (1 << 24)) {
            __builtin_amdgcn_s_sleep(8);
            ++spins;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

// epilogue shared by both tile configurations: the wave's 128 x 64 accumulator tile goes through a private LDS slab in two
// 64-row passes and is streamed out as whole 16-byte segments (mwave0 = first output row of the wave, nbase = first column)
// Stream-K: up to three parked partial tiles (fp32 [BM][bnt] row-major, see s3_park) of the blocks that started this tile are
// added while the tile streams out — in the fixed order p0, p1, p2 after the finishing block's own sums — so the gather costs
// no registers beyond one float4.  tile_m0 / tile_n0: first row / column of the block tile; bnt: its width.
struct SkParts {
    const float* p0;
    const float* p1;
    const float* p2;
    int64_t tile_m0;
    int tile_n0, bnt;
};
__device__ __forceinline__ void sk_add4(const SkParts& sp, int64_t m, int n, f32x4& v) {
    const int64_t off = (m - sp.tile_m0) * sp.bnt + (n - sp.tile_n0);
    if (sp.p0) v += *reinterpret_cast<const f32x4*>(sp.p0 + off);
    if (sp.p1) v += *reinterpret_cast<const f32x4*>(sp.p1 + off);
    if (sp.p2) v += *reinterpret_cast<const f32x4*>(sp.p2 + off);
}

// park the wave's 128 x 64 accumulator tile (raw sums, no bias) in the block's partial buffer through the same LDS slab
template <int BNT>
__device__ __forceinline__ void s3_park(f32x16 (&acc)[4][2], float* slab, float* part, int row0, int col0, int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    constexpr int CLD = 64 + 4;
    const int cr = lane >> 4, cc = (lane & 15) * 4;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[ps * 2 + i][j][r];
        float* dst = part + (int64_t)(row0 + ps * 64 + cr) * BNT + col0 + cc;
#pragma unroll
        for (int it = 0; it < 16; ++it)
            *reinterpret_cast<f32x4*>(dst + (int64_t)it * 4 * BNT) = *reinterpret_cast<const f32x4*>(slab + (cr + it * 4) * CLD + cc);
    }
}

template <int EPI, bool F16>
__device__ __forceinline__ void s3_epilogue(const S3Args& g, f32x16 (&acc)[4][2], float* slab, int64_t mwave0, int nbase, int lane,
                                            const SkParts& sp) {
    const int l31 = lane & 31, hi = lane >> 5;
    // epilogue: two 64-row passes per wave through a private LDS slab, streamed out as whole 16-byte segments
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
        if constexpr (EPI == S3_EPI_QKV3) {
            // packed in_proj output -> the qkv3 image attn_bf16x3.hip reads: a wave's 64 columns are one (part, head)
            const int cr = lane >> 3, c = lane & 7;
            const int n = nbase + c * 8;
            const int dmodel = g.heads * 64;
            const int part = nbase / dmodel, head = (nbase % dmodel) >> 6;
            const int Bt = (int)(g.M / g.tokN);
            float bv[8];
            *reinterpret_cast<f32x4*>(bv) = *reinterpret_cast<const f32x4*>(g.bias + n);
            *reinterpret_cast<f32x4*>(bv + 4) = *reinterpret_cast<const f32x4*>(g.bias + n + 4);
            const float mul = part == 0 ? g.qscale : 1.0f;
            unsigned char* pbase = g.C3 + (((int64_t)part * Bt) * g.heads + head) * (int64_t)g.tokNpad * QKV3_ROWB;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int64_t m = m0 + cr + it * 8;
                float v[8];
                *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(slab + (cr + it * 8) * CLD + c * 8);
                *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(slab + (cr + it * 8) * CLD + c * 8 + 4);
                if (sp.p0) {
                    sk_add4(sp, m, n, *reinterpret_cast<f32x4*>(v));
                    sk_add4(sp, m, n + 4, *reinterpret_cast<f32x4*>(v + 4));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = ((F16 ? v[e] * g.ab_inv : v[e]) + bv[e]) * mul;
                if (m < g.M) {
                    const int b = (int)(m / g.tokN), tok = (int)(m - (int64_t)b * g.tokN);
                    unsigned char* dst = pbase + ((int64_t)b * g.heads * g.tokNpad + tok) * QKV3_ROWB + ((c ^ qkv3_swizzle(part, tok)) << 4);
                    if constexpr (F16) {
                        u32x4 Hh, Lo;
                        split8_h2(v, g.c_scale, Hh, Lo);
                        *reinterpret_cast<u32x4*>(dst) = Hh;
                        *reinterpret_cast<u32x4*>(dst + 128) = Lo;
                    } else {
                        u32x4 Hh, Mi, Lo;
                        split8(v, Hh, Mi, Lo);
                        *reinterpret_cast<u32x4*>(dst) = Hh;
                        *reinterpret_cast<u32x4*>(dst + 128) = Mi;
                        *reinterpret_cast<u32x4*>(dst + 256) = Lo;
                    }
                }
            }
        } else if constexpr (EPI == S3_EPI_GELU_SPLIT || EPI == S3_EPI_SPLIT) {
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
                if (sp.p0) {
                    sk_add4(sp, m, n, *reinterpret_cast<f32x4*>(v));
                    sk_add4(sp, m, n + 4, *reinterpret_cast<f32x4*>(v + 4));
                }
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
                    if (sp.p0) sk_add4(sp, m0 + cr + it * 4, n, v);
                    if constexpr (F16) v *= g.ab_inv;
                    v += bv;
                    if constexpr (EPI == S3_EPI_RES) v += rv[u];
                    if (m0 + cr + it * 4 < g.M) *reinterpret_cast<f32x4*>(cptr + (int64_t)it * 4 * g.N) = v;
                }
            }
        }
    }
}

// Product terms kept per k (A plane, B plane), small first.  6: everything down to 2^-16 relative (hh, hm, mh, hl, lh, mm) — the
// default, error of an fp32 FMA chain.  9: all nine, nothing dropped ("strict": the only rounding left is the fp32 accumulate).
// 1: hh only — plain bf16 operands with fp32 accumulation, the reduced-precision variant BASELINE config C2 names (error is
// reported against the fp32 result, never a parity path); it also moves only the h plane from global memory.
template <int TERMS> struct S3Terms;
template <> struct S3Terms<6> { static constexpr int N = 6; static constexpr int PA[6] = {2, 0, 1, 1, 0, 0}; static constexpr int PB[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct S3Terms<9> { static constexpr int N = 9; static constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}; static constexpr int PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}; };
template <> struct S3Terms<1> { static constexpr int N = 1; static constexpr int PA[1] = {0}; static constexpr int PB[1] = {0}; };
// 3: f16x2 images (two fp16 planes h, l with 11 significant bits each): hl, lh, hh; ll (2^-22 relative) is dropped
template <> struct S3Terms<3> { static constexpr int N = 3; static constexpr int PA[3] = {0, 1, 0}; static constexpr int PB[3] = {1, 0, 0}; };

// LDS stage = the planes a mode moves, compact: [region (128 rows)][plane][128 rows][32 B]; region stride = planes x 4 KiB.
// The global -> LDS transport, not the matrix pipe, paces these loops (~25 GB/s per CU per tile in flight), so
// the modes that move fewer planes spend the LDS they save on a deeper ring: more tiles in flight per CU.
constexpr int S3_BM = 256, S3_BN = 256;
__host__ __device__ constexpr int s3_nst(int terms) { return terms == 3 ? 4 : terms == 1 ? 6 : 3; }
__host__ __device__ constexpr int s3_stage(int terms) { return (S3_BM + S3_BN) / 128 * s3_planes(terms) * S3_PLANE; }   // 48 / 32 / 16 KiB
constexpr int S3_SLABS = 8 * 64 * 68 * 4;                // epilogue slabs of the 8 waves, reuse the stage area
__host__ __device__ constexpr int s3_lds(int terms) {
    return s3_nst(terms) * s3_stage(terms) > S3_SLABS ? s3_nst(terms) * s3_stage(terms) : S3_SLABS;
}

// wait until at most n of this wave's DMA pieces are still in flight (n is wave-uniform, 0 <= n <= MAXN)
template <int MAXN, int STEP>
__device__ __forceinline__ void wait_vm_tiles(int tiles) {
    if constexpr (MAXN == 0) { wait_vm<0>(); }
    else {
        if (tiles >= MAXN) wait_vm<MAXN * STEP>();
        else wait_vm_tiles<MAXN - 1, STEP>(tiles);
    }
}

template <int EPI, int TERMS, bool SK>
__global__ __launch_bounds__(512, 1) void gemm_bf16x3_kernel(S3Args g) {
    constexpr int BM = S3_BM, BN = S3_BN, WM = 128, WN = 64;
    constexpr int TM = 4, TN = 2, NST = s3_nst(TERMS), STAGE = s3_stage(TERMS);
    constexpr int NPL = s3_planes(TERMS);                 // planes moved and read
    constexpr int RCH = NPL * S3_PLANE;                   // one 128-row region of a stage
    constexpr bool F16 = TERMS == 3;
    constexpr int PPR = 4 * NPL;                          // one-KiB pieces per 128-row region per stage
    constexpr int PPW = 4 * PPR / 8;                      // 4 regions per stage / 8 waves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

    constexpr bool sk = SK;
    [[maybe_unused]] const unsigned long long t_entry = S3_T(), rt_entry = S3_RT();
    [[maybe_unused]] unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_body = 0;
    int bm, bn;
    SkRange skr{};
    int sk_s = 0, sk_phase = 0;
    if (!sk) {
        int wg;
        {
            const int b = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
            wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        }
        // super-tiles of sm x sn blocks: the blocks an XCD runs together share sm A panels and sn W panels
        const int per_row = g.sm * g.nbn, per_st = g.sm * g.sn;
        const int srow = wg / per_row, rem = wg % per_row;
        const int sc = rem / per_st, rem2 = rem % per_st;
        bm = srow * g.sm + rem2 / g.sn;
        bn = sc * g.sn + rem2 % g.sn;
        if ((int64_t)bm * BM >= g.M) return;
    } else {
        skr = sk_range(g.ntiles, g.K >> 4);
        bm = bn = 0;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int ng = g.K >> 4;
    const int nrtA = (int)((g.M + 127) >> 7);

    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        a_off[i] = (r >> 7) * RCH + (r & 127) * 32 + ((hi ^ ((r >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = BM + wn * WN + j * 32 + l31;
        b_off[j] = (r >> 7) * RCH + (r & 127) * 32 + ((hi ^ ((r >> 3) & 1)) << 4);
    }
    using TT = S3Terms<TERMS>;
    constexpr int CLD = WN + 4;

    for (;;) {      // segments of a stream-K block; exactly one pass otherwise
        int k0 = 0, k1 = ng;
        if (sk) {
            if (!sk_next(skr, ng, sk_phase, sk_s, k0, k1)) break;
            const int t = skr.tile_base + sk_s / ng;
            bm = __builtin_amdgcn_readfirstlane(t / g.nbn);
            bn = __builtin_amdgcn_readfirstlane(t % g.nbn);
            k0 = __builtin_amdgcn_readfirstlane(k0);
            k1 = __builtin_amdgcn_readfirstlane(k1);
        }
        // DMA: the stage image is [A row-tile 0 | A row-tile 1 | W row-tile 0 | W row-tile 1], each a 12 KiB chunk of which the
        // first NPL planes (4 KiB each) are moved
        const unsigned char* src[PPW];
        int dst_off[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave * PPW + i;
            const int region = q / PPR, within = (q % PPR) * 1024;
            dst_off[i] = region * RCH + within;
            if (region < 2) {
                int rt = bm * 2 + region;
                rt = rt < nrtA ? rt : nrtA - 1;
                src[i] = g.A + ((int64_t)rt * ng + k0) * S3_CHUNK + within + lane * 16;
            } else {
                src[i] = g.W + ((int64_t)(bn * 2 + region - 2) * ng + k0) * S3_CHUNK + within + lane * 16;
            }
        }
        auto issue = [&](int kt, int buf) {
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(src[i] + (int64_t)kt * S3_CHUNK),
                                                 AVD_LDS_PTR(smem3 + buf * STAGE + dst_off[i]), 16, 0, 0);
        };

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        const int nk = k1 - k0;
#pragma unroll
        for (int t = 0; t < NST - 1; ++t)
            if (t < nk) issue(t, t);
        int cur = 0, nxt = NST - 1;
        [[maybe_unused]] const unsigned long long t_loop = S3_T();
        for (int kt = 0; kt < nk; ++kt) {
            [[maybe_unused]] const unsigned long long t0 = S3_T();
            // tile kt must have landed; the (up to NST - 2) tiles issued after it may stay in flight
            wait_vm_tiles<NST - 2, PPW>((kt + NST - 1 <= nk ? kt + NST - 1 : nk) - kt - 1);
            [[maybe_unused]] const unsigned long long t1 = S3_T();
            asm volatile("s_barrier" ::: "memory");   // no fence: a fence would drain vmcnt and with it the tiles in flight
            [[maybe_unused]] const unsigned long long t2 = S3_T();
            if (kt + NST - 1 < nk) issue(kt + NST - 1, nxt);
            [[maybe_unused]] const unsigned long long t3 = S3_T();
            const unsigned char* st = smem3 + cur * STAGE;
            cur = cur + 1 == NST ? 0 : cur + 1;
            nxt = nxt + 1 == NST ? 0 : nxt + 1;
            bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int p = 0; p < NPL; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + S3_PLANE * p);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int p = 0; p < NPL; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + S3_PLANE * p);
#pragma unroll
            for (int t = 0; t < TT::N; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = mma16<F16>(af[i][TT::PA[t]], bf[j][TT::PB[t]], acc[i][j]);
#ifdef AVD_S3_STAMPS
            asm volatile("s_nop 0" ::: "memory");
            c_wait += t1 - t0; c_bar += t2 - t1; c_issue += t3 - t2; c_body += S3_T() - t3;
#endif
        }
        [[maybe_unused]] const unsigned long long t_end = S3_T();
        __syncthreads();

        float* slab = reinterpret_cast<float*>(smem3) + wave * 64 * CLD;
        SkParts sp{nullptr, nullptr, nullptr, (int64_t)bm * BM, bn * BN, BN};
        if (sk && k1 < ng) {
            // the range ends inside this tile: park the partial sums for the block that finishes the tile
            s3_park<BN>(acc, slab, g.sk_partial + (int64_t)blockIdx.x * (BM * BN), wm * WM, wn * WN, lane);
            sk_publish(g.sk_flags + blockIdx.x, tid);
            continue;
        }
        if (sk && k0 > 0) {
            // this block finishes a tile other blocks of its XCD started: their parts are added in ascending block order
            const int T0 = (sk_s / ng) * ng;                     // first step of the tile in the XCD's step sequence
            int first = skr.j - 1;
            while (first > 0 && skr.hi_of(first - 1) > T0) --first;
            int np = 0;
            for (int jj = first; jj < skr.j; ++jj, ++np) {
                const int bid = jj * 8 + (blockIdx.x & 7);
                sk_await(g.sk_flags + bid, tid);
                if (tid == 0) __hip_atomic_store(g.sk_flags + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float* pp = g.sk_partial + (int64_t)bid * (BM * BN);
                if (np == 0) sp.p0 = pp; else if (np == 1) sp.p1 = pp; else sp.p2 = pp;
            }
        }
        s3_epilogue<EPI, F16>(g, acc, slab, (int64_t)bm * BM + wm * WM, bn * BN + wn * WN, lane, sp);
#ifdef AVD_S3_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        S3_DBG(0, t_entry); S3_DBG(1, t_loop); S3_DBG(2, t_end); S3_DBG(3, S3_T());
        S3_DBG(4, c_wait); S3_DBG(5, c_bar); S3_DBG(6, c_issue); S3_DBG(7, c_body); S3_DBG(8, (unsigned long long)nk);
        S3_DBG(9, S3_RT()); S3_DBG(10, rt_entry);
#endif
        if (!sk) break;
        __syncthreads();      // slabs drained before the next segment's DMA lands on them
    }
}

// Second tile configuration: 256 x 128 block, 4 waves (same 128 x 64 wave tile), TWO blocks per CU so that one block's
// epilogue (the split3 / qkv3 epilogues move 1.5x the bytes of an fp32 one) overlaps the other's main loop.  Two 36 KiB LDS
// stages; the fragments are kept one K-tile ahead in registers: while the 48 MFMAs of tile kt run, the 18 fragment reads of
// tile kt+1 are spread between them and the DMA of tile kt+2 is in flight.  The term order is chosen so that every operand
// plane is dead before its successor is read into the same registers.
constexpr int S3B_BM = 256, S3B_BN = 128;
// two blocks per CU: <= 80 KiB each.  Three planes: two 36 KiB stages; two planes (f16x2): three 24 KiB stages; one: four 12 KiB.
__host__ __device__ constexpr int s3b_nst(int terms) { return terms == 3 ? 3 : terms == 1 ? 4 : 2; }
__host__ __device__ constexpr int s3b_stage(int terms) { return (S3B_BM + S3B_BN) / 128 * s3_planes(terms) * S3_PLANE; }
constexpr int S3B_SLABS = 4 * 64 * 68 * 4;               // epilogue slabs of the 4 waves
__host__ __device__ constexpr int s3b_lds(int terms) {
    return s3b_nst(terms) * s3b_stage(terms) > S3B_SLABS ? s3b_nst(terms) * s3b_stage(terms) : S3B_SLABS;
}

template <int EPI, int TERMS, bool SK>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_b_kernel(S3Args g) {
    constexpr int BM = S3B_BM, BN = S3B_BN, WM = 128, WN = 64;
    constexpr int TM = 4, TN = 2, NST = s3b_nst(TERMS), STAGE = s3b_stage(TERMS);
    constexpr int NPL = s3_planes(TERMS);
    constexpr bool F16 = TERMS == 3;
    constexpr int PPR = 4 * NPL;                          // one-KiB pieces per 128-row region per stage
    constexpr int PPW = 3 * PPR / 4;
    constexpr int RCH = NPL * S3_PLANE;                   // one 128-row region of a stage                      // 3 regions per stage / 4 waves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];

    constexpr bool sk = SK;
    [[maybe_unused]] const unsigned long long t_entry = S3_T(), rt_entry = S3_RT();
    [[maybe_unused]] unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_body = 0;
    int bm, bn;
    SkRange skr{};
    int sk_s = 0, sk_phase = 0;
    if (!sk) {
        int wg;
        {
            const int b = blockIdx.x, nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
            wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
        }
        const int per_row = g.sm * g.nbn, per_st = g.sm * g.sn;
        const int srow = wg / per_row, rem = wg % per_row;
        const int sc = rem / per_st, rem2 = rem % per_st;
        bm = srow * g.sm + rem2 / g.sn;
        bn = sc * g.sn + rem2 % g.sn;
        if ((int64_t)bm * BM >= g.M) return;
    } else {
        skr = sk_range(g.ntiles, g.K >> 4);
        bm = bn = 0;
    }

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    if (g.stagger > 0) {
        // Two blocks share a CU.  Started together they stay in phase: both in the K loop (each at half the matrix pipe's rate),
        // then both in the epilogue (the pipe idle).  The block whose waves sit in the odd slots of their SIMDs starts late, so one
        // block's epilogue runs beside the other's loop.
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if ((hwid & 1u) && (int)blockIdx.x < g.first_gen)      // first generation only (2 slots per CU); later blocks inherit the phase
            for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const int ng = g.K >> 4;
    const int nrtA = (int)((g.M + 127) >> 7);

    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        a_off[i] = (r >> 7) * RCH + (r & 127) * 32 + ((hi ^ ((r >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = BM + wn * WN + j * 32 + l31;
        b_off[j] = (r >> 7) * RCH + (r & 127) * 32 + ((hi ^ ((r >> 3) & 1)) << 4);
    }
#define S3_LDA(dst, st, p) _Pragma("unroll") for (int i = 0; i < TM; ++i) dst[i] = *reinterpret_cast<const bf16x8*>((st) + a_off[i] + S3_PLANE * (p))
#define S3_LDB(dst, st, p) _Pragma("unroll") for (int j = 0; j < TN; ++j) dst[j] = *reinterpret_cast<const bf16x8*>((st) + b_off[j] + S3_PLANE * (p))
#define S3_MM(A_, B_) _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
        acc[i][j] = mma16<F16>(A_[i], B_[j], acc[i][j])

    for (;;) {      // segments of a stream-K block; exactly one pass otherwise
    int k0 = 0, k1 = ng;
    if (sk) {
        if (!sk_next(skr, ng, sk_phase, sk_s, k0, k1)) break;
        const int t = skr.tile_base + sk_s / ng;
        bm = __builtin_amdgcn_readfirstlane(t / g.nbn);
        bn = __builtin_amdgcn_readfirstlane(t % g.nbn);
        k0 = __builtin_amdgcn_readfirstlane(k0);
        k1 = __builtin_amdgcn_readfirstlane(k1);
    }
    // stage image [A row-tile 0 | A row-tile 1 | W row-tile], 12 KiB chunks of which the first NPL planes are moved
    const unsigned char* src[PPW];
    int dst_off[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int region = q / PPR, within = (q % PPR) * 1024;
        dst_off[i] = region * RCH + within;
        if (region < 2) {
            int rt = bm * 2 + region;
            rt = rt < nrtA ? rt : nrtA - 1;
            src[i] = g.A + ((int64_t)rt * ng + k0) * S3_CHUNK + within + lane * 16;
        } else {
            src[i] = g.W + ((int64_t)bn * ng + k0) * S3_CHUNK + within + lane * 16;
        }
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(src[i] + (int64_t)kt * S3_CHUNK),
                                             AVD_LDS_PTR(smem3 + buf * STAGE + dst_off[i]), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ring of NST stages: while the MFMAs of tile kt run on fragments held in registers, the fragments of tile kt+1 are read from
    // its stage and tiles kt+2 .. kt+NST-1 are in flight (the stage of tile kt is refilled with tile kt+NST at the top)
    const int nk = k1 - k0;
    issue(0, 0);
    wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
#pragma unroll
    for (int t = 1; t < NST; ++t)
        if (t < nk) issue(t, t);
    bf16x8 ah[TM], am[TM], al[TM], bh[TN], bmm[TN], bl[TN];
    S3_LDA(ah, smem3, 0);
    S3_LDB(bh, smem3, 0);
    if constexpr (TERMS != 1) {
        S3_LDA(am, smem3, 1);
        S3_LDB(bmm, smem3, 1);
    }
    if constexpr (TERMS == 6 || TERMS == 9) {
        S3_LDA(al, smem3, 2);
        S3_LDB(bl, smem3, 2);
    }

    int st_cur = 0, st_nx = 1 % NST;
    [[maybe_unused]] const unsigned long long t_loop = S3_T();
    for (int kt = 0; kt < nk; ++kt) {
        [[maybe_unused]] const unsigned long long t0 = S3_T();
        // tile kt+1 must have landed (tiles kt+2 .. may stay in flight); after the barrier every wave holds tile kt's fragments
        // in registers, so its stage is free for tile kt+NST
        wait_vm_tiles<NST - 2, PPW>((kt + NST <= nk ? kt + NST : nk) - kt - 2);
        [[maybe_unused]] const unsigned long long t1 = S3_T();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        [[maybe_unused]] const unsigned long long t2 = S3_T();
        if (kt + NST < nk) issue(kt + NST, st_cur);
        [[maybe_unused]] const unsigned long long t3 = S3_T();
        const unsigned char* nx = smem3 + st_nx * STAGE;
        st_cur = st_nx;
        st_nx = st_nx + 1 == NST ? 0 : st_nx + 1;
        bf16x8 ah_n[TM], bl_n[TN];
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TERMS == 1) {
            bf16x8 bh_n[TN];
            S3_LDA(ah_n, nx, 0);
            S3_LDB(bh_n, nx, 0);
            __builtin_amdgcn_sched_barrier(0);
            S3_MM(ah, bh);                     // (h,h)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) bh[j] = bh_n[j];
        } else if constexpr (TERMS == 3) {     // f16x2: plane 1 (am / bmm) is l
            bf16x8 bh_n[TN];
            S3_MM(ah, bmm);                    // (h,l)  -> bmm dead
            S3_LDB(bmm, nx, 1);
            S3_LDA(ah_n, nx, 0);
            __builtin_amdgcn_sched_barrier(0);
            S3_MM(am, bh);                     // (l,h)  -> am dead
            S3_LDA(am, nx, 1);
            S3_LDB(bh_n, nx, 0);
            __builtin_amdgcn_sched_barrier(0);
            S3_MM(ah, bh);                     // (h,h)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) bh[j] = bh_n[j];
        } else {
            S3_MM(am, bmm);                    // (m,m)
            S3_LDA(ah_n, nx, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TERMS == 9) {
                S3_MM(am, bl);                 // (m,l)
                __builtin_amdgcn_sched_barrier(0);
            }
            S3_MM(am, bh);                     // (m,h)  -> am dead
            S3_LDA(am, nx, 1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (TERMS == 9) {
                S3_MM(al, bl);                 // (l,l)
                __builtin_amdgcn_sched_barrier(0);
                S3_MM(al, bmm);                // (l,m)
                __builtin_amdgcn_sched_barrier(0);
            }
            S3_MM(al, bh);                     // (l,h)  -> al dead
            S3_LDA(al, nx, 2);
            __builtin_amdgcn_sched_barrier(0);
            S3_MM(ah, bh);                     // (h,h)  -> bh dead
            S3_LDB(bh, nx, 0);
            S3_LDB(bl_n, nx, 2);
            __builtin_amdgcn_sched_barrier(0);
            S3_MM(ah, bmm);                    // (h,m)  -> bmm dead
            S3_LDB(bmm, nx, 1);
            __builtin_amdgcn_sched_barrier(0);
            S3_MM(ah, bl);                     // (h,l)  -> ah, bl dead
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) bl[j] = bl_n[j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[i] = ah_n[i];
#ifdef AVD_S3_STAMPS
        asm volatile("s_nop 0" ::: "memory");
        c_wait += t1 - t0; c_bar += t2 - t1; c_issue += t3 - t2; c_body += S3_T() - t3;
#endif
    }
    [[maybe_unused]] const unsigned long long t_end = S3_T();
    __syncthreads();
    constexpr int CLD = WN + 4;
    float* slab = reinterpret_cast<float*>(smem3) + wave * 64 * CLD;
    SkParts sp{nullptr, nullptr, nullptr, (int64_t)bm * BM, bn * BN, BN};
    if (sk && k1 < ng) {
        s3_park<BN>(acc, slab, g.sk_partial + (int64_t)blockIdx.x * (BM * BN), wm * WM, wn * WN, lane);
        sk_publish(g.sk_flags + blockIdx.x, tid);
        continue;
    }
    if (sk && k0 > 0) {
        const int T0 = (sk_s / ng) * ng;
        int first = skr.j - 1;
        while (first > 0 && skr.hi_of(first - 1) > T0) --first;
        int np = 0;
        for (int jj = first; jj < skr.j; ++jj, ++np) {
            const int bid = jj * 8 + (blockIdx.x & 7);
            sk_await(g.sk_flags + bid, tid);
            if (tid == 0) __hip_atomic_store(g.sk_flags + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float* pp = g.sk_partial + (int64_t)bid * (BM * BN);
            if (np == 0) sp.p0 = pp; else if (np == 1) sp.p1 = pp; else sp.p2 = pp;
        }
    }
    s3_epilogue<EPI, F16>(g, acc, slab, (int64_t)bm * BM + wm * WM, bn * BN + wn * WN, lane, sp);
#ifdef AVD_S3_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    S3_DBG(0, t_entry); S3_DBG(1, t_loop); S3_DBG(2, t_end); S3_DBG(3, S3_T());
    S3_DBG(4, c_wait); S3_DBG(5, c_bar); S3_DBG(6, c_issue); S3_DBG(7, c_body); S3_DBG(8, (unsigned long long)nk);
    S3_DBG(9, S3_RT()); S3_DBG(10, rt_entry);
#endif
    if (!sk) break;
    __syncthreads();      // slabs drained before the next segment's DMA lands on them
    }   // segments
#undef S3_LDA
#undef S3_LDB
#undef S3_MM
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
int64_t split3_bytes(int64_t rows, int K) { return ((rows + 255) / 256 * 256) * (int64_t)K * 6; }

int split3_f32(const float* x, int64_t ld, void* out, int64_t rows, int K, hipStream_t st, float h2_scale) {
    return split3_rows_f32(x, RowMap{ld, 0, 0}, out, rows, K, st, h2_scale);
}

int split3_rows_f32(const float* x, RowMap xm, void* out, int64_t rows, int K, hipStream_t st, float h2_scale) {
    AVD_REQUIRE(x && out, AVD_EINVAL, "split3: null pointer");
    const int64_t ld = xm.ld;
    AVD_REQUIRE(rows > 0 && K > 0 && K % 16 == 0 && ld >= K && ld % 4 == 0 && xm.stride % 4 == 0, AVD_EUNSUPPORTED,
                "split3: need rows > 0, K %% 16 == 0, ld %% 4 == 0 (rows=%lld K=%d ld=%lld)", (long long)rows, K, (long long)ld);
    AVD_REQUIRE(aligned16(x) && aligned16(out), AVD_EUNSUPPORTED, "split3: pointers must be 16-byte aligned");
    const int64_t rows_pad = (rows + 255) / 256 * 256;
    const int64_t n = rows_pad * (K / 8);
    AVD_REQUIRE((n + 255) / 256 < (1ll << 31), AVD_EUNSUPPORTED, "split3: grid too large");
    static const int tag = prof_tag_id("split3_kernel");
    ProfScope prof(tag, (double)rows * K * 10.0, st);
    AVD_REQUIRE(h2_scale >= 0.f && h2_scale < __builtin_inff(), AVD_EINVAL, "split3: f16x2 image scale must be positive and finite");
    if (h2_scale > 0.f)
        hipLaunchKernelGGL(split3_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xm,
                           static_cast<unsigned char*>(out), rows, rows_pad, K, h2_scale);
    else
        hipLaunchKernelGGL(split3_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, xm,
                           static_cast<unsigned char*>(out), rows, rows_pad, K, 0.f);
    AVD_CHECK_LAUNCH("split3");
    return AVD_OK;
}

int rmsnorm_split3_f32(const float* x, const float* scale, void* out, int64_t rows, int d, float eps, hipStream_t st, float h2_scale) {
    AVD_REQUIRE(x && scale && out, AVD_EINVAL, "rmsnorm_split3: null pointer");
    AVD_REQUIRE(rows > 0 && d > 0 && d % 16 == 0 && d <= 2048, AVD_EUNSUPPORTED, "rmsnorm_split3: d=%d must be a multiple of 16, <= 2048", d);
    AVD_REQUIRE(aligned16(x) && aligned16(out), AVD_EUNSUPPORTED, "rmsnorm_split3: pointers must be 16-byte aligned");
    static const int tag = prof_tag_id("rmsnorm_split3_kernel");
    ProfScope prof(tag, 10.0 * (double)rows * d, st);
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
    ProfScope prof(tag, 10.0 * (double)rows * d, st);
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

// tile configuration: 0 = 256x256, 8 waves, one block per CU; 1 = 256x128, 4 waves, two blocks per CU.
// AVD_S3_TILE=0|1 forces one (measurement aid); default: per epilogue, what measured faster in the C3 pipeline.
static int s3_tile_for(int epi, int64_t M, int N) {
    static const int forced = [] { const char* e = getenv("AVD_S3_TILE"); return e ? atoi(e) : -1; }();
    if (forced == 0 || forced == 1) return forced;
    // C3 pipeline, ms per step over 8 launches: fc1+GELU->split3 2.75 (one block/CU) vs 2.48 (two); in_proj->qkv3 1.98 vs 1.94;
    // out_proj/fc2 + residual (16 launches) 3.07 vs 3.27
    if (epi == S3_EPI_GELU_SPLIT || epi == S3_EPI_QKV3 || epi == S3_EPI_SPLIT) return 1;
    // 256x256 tiles only when they occupy most of the 256 CUs (C3: 106 x 2 = 212 blocks); at the 128x128 geometry
    // (8,512 rows) the 256x128 tiles run the whole step in 3.75 ms against 4.78
    return (M + 255) / 256 * (N / 256) >= 192 ? 0 : 1;
}

int g_s3_streamk = getenv("AVD_S3_STREAMK") ? atoi(getenv("AVD_S3_STREAMK")) : 0;   // measured slower than plain tiling: off
// Stagger of the two co-resident blocks of the 256x128 kernel, x 1024 cycles; -1 = automatic: about a third of a tile's life when
// the launch has at least two generations of blocks and the caller is not already running two kernel chains on two streams (those
// drift apart by themselves; there the start-up delay only costs).  Measured at C3, single stream: f16x2 162.1 -> 166.3 steps/s,
// bf16x3 105.4 -> 107.7; with two streams 172.7 -> 167.2 (hence off).  avd_tune_set "s3_stagger".
int g_s3_stagger = getenv("AVD_S3_STAGGER") ? atoi(getenv("AVD_S3_STAGGER")) : -1;
thread_local bool t_s3_two_streams = false;      // set by avd_denoise_step_f32 around its two-stream section
#ifdef AVD_S3_STAMPS
unsigned long long* g_s3_dbg = nullptr;
extern "C" void lab_set_dbg(unsigned long long* p) { g_s3_dbg = p; }
#endif

static int sk_cu_count() {                 // CUs of the current device (looked up once per device)
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
static int sk_flags_base(unsigned int** out) {
    static std::atomic<unsigned int*> cache[64];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(AVD_ELAUNCH, "stream-K flags: hipGetDevice: %s", hipGetErrorString(e));
    unsigned int* p = dev < 64 ? cache[dev].load(std::memory_order_acquire) : nullptr;
    if (!p) {
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(g_s3_sk_flags));
        if (e != hipSuccess) return set_error(AVD_ELAUNCH, "stream-K flags: %s", hipGetErrorString(e));
        if (dev < 64) cache[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return AVD_OK;
}

template <int EPI, int TERMS>
static int launch_s3t(const S3Args& a, hipStream_t st) {
    const int tile = s3_tile_for(EPI, a.M, a.N);
    const int BMt = tile ? S3B_BM : S3_BM, BNt = tile ? S3B_BN : S3_BN, lds = tile ? s3b_lds(TERMS) : s3_lds(TERMS);
    static LdsAttr attr[2];
    const void* kern = tile ? reinterpret_cast<const void*>(gemm_bf16x3_b_kernel<EPI, TERMS, false>)
                            : reinterpret_cast<const void*>(gemm_bf16x3_kernel<EPI, TERMS, false>);
    if (int rc = attr[tile].ensure(kern, lds, "gemm_bf16x3")) return rc;
    S3Args g = a;
#ifdef AVD_S3_STAMPS
    g.dbg = g_s3_dbg;
#endif
    g.nbn = a.N / BNt;
    // stream-K when the caller lent scratch for the parked partial tiles and plain tiling would leave a ragged last round
    {
        const int64_t nbm_ = (a.M + BMt - 1) / BMt;
        const int64_t ntiles = nbm_ * g.nbn;
        const int G = (tile ? 2 : 1) * sk_cu_count();
        const bool ragged = ntiles % G != 0;
        if (g_s3_streamk && a.sk_partial && G > 0 && G % 8 == 0 && G <= S3_SK_FLAGS && ntiles >= G / 2 && ntiles < (1 << 20) && ragged &&
            a.sk_floats >= (int64_t)G * BMt * BNt) {
            static std::atomic<unsigned> seq{0};
            unsigned int* fbase = nullptr;
            if (int rc = sk_flags_base(&fbase)) return rc;
            g.sk_flags = fbase + (seq.fetch_add(1) % S3_SK_SLOTS) * S3_SK_FLAGS;
            g.ntiles = (int)ntiles;
            g.sm = g.sn = 1;
            static const int tagk0 = prof_tag_id("gemm_bf16x3_kernel<%d, %d, true>", EPI, TERMS), tagk1 = prof_tag_id("gemm_bf16x3_b_kernel<%d, %d, true>", EPI, TERMS);
            ProfScope prof(tile ? tagk1 : tagk0, 2.0 * (double)a.M * a.N * a.K, st);
            static LdsAttr attrk[2];
            const void* kk = tile ? reinterpret_cast<const void*>(gemm_bf16x3_b_kernel<EPI, TERMS, true>)
                                  : reinterpret_cast<const void*>(gemm_bf16x3_kernel<EPI, TERMS, true>);
            if (int rc = attrk[tile].ensure(kk, lds, "gemm_bf16x3 (stream-K)")) return rc;
            if (tile) hipLaunchKernelGGL((gemm_bf16x3_b_kernel<EPI, TERMS, true>), dim3((unsigned)G), dim3(256), lds, st, g);
            else hipLaunchKernelGGL((gemm_bf16x3_kernel<EPI, TERMS, true>), dim3((unsigned)G), dim3(512), lds, st, g);
            AVD_CHECK_LAUNCH("gemm_bf16x3 (stream-K)");
            return AVD_OK;
        }
        g.sk_partial = nullptr;
        g.sk_flags = nullptr;
    }
    int sn = 8;
    while (g.nbn % sn) sn >>= 1;
    // 256x128 tiles: whole block rows per super-tile, so the 12-16 column blocks that share an A panel run together and the panel
    // is fetched into the XCD's L2 once (in_proj at C3: 2.02 -> 1.84 ms per step)
    if (tile && g.nbn <= 16) sn = g.nbn;
    int total = tile ? 32 : 16;
    {   // AVD_S3_SN / AVD_S3_SUPER: measurement aids (super-tile width in blocks / blocks per super-tile)
        static const int e_sn = getenv("AVD_S3_SN") ? atoi(getenv("AVD_S3_SN")) : 0;
        static const int e_tot = getenv("AVD_S3_SUPER") ? atoi(getenv("AVD_S3_SUPER")) : 0;
        if (e_sn > 0) { sn = e_sn < g.nbn ? e_sn : g.nbn; while (g.nbn % sn) --sn; }
        if (e_tot > 0) total = e_tot;
    }
    g.sn = sn;
    g.sm = total / sn > 0 ? total / sn : 1;
    const int64_t nbm = (a.M + BMt - 1) / BMt;
    const int64_t nwg = (nbm + g.sm - 1) / g.sm * g.sm * g.nbn;
    AVD_REQUIRE(nwg < (1ll << 31), AVD_EUNSUPPORTED, "gemm_bf16x3 grid too large");
    g.stagger = 0;
    g.first_gen = 2 * sk_cu_count();
    if (tile && g.first_gen > 0 && nbm * g.nbn >= 2 * g.first_gen)
        g.stagger = g_s3_stagger >= 0 ? g_s3_stagger : t_s3_two_streams ? 0 : (TERMS == 3 ? 24 : TERMS == 1 ? 12 : 48) * (a.K >= 1024 ? 2 : 1);
    // tags = the kernel names as rocprofv3 prints their template arguments (EPI, TERMS, stream-K)
    static const int tag0 = prof_tag_id("gemm_bf16x3_kernel<%d, %d, false>", EPI, TERMS), tag1 = prof_tag_id("gemm_bf16x3_b_kernel<%d, %d, false>", EPI, TERMS);
    ProfScope prof(tile ? tag1 : tag0, 2.0 * (double)a.M * a.N * a.K, st);
    if (tile) hipLaunchKernelGGL((gemm_bf16x3_b_kernel<EPI, TERMS, false>), dim3((unsigned)nwg), dim3(256), lds, st, g);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<EPI, TERMS, false>), dim3((unsigned)nwg), dim3(512), lds, st, g);
    AVD_CHECK_LAUNCH("gemm_bf16x3");
    return AVD_OK;
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
int64_t gemm_bf16x3_sk_floats() { return (int64_t)2 * (sk_cu_count() > 0 ? sk_cu_count() : 256) * S3B_BM * S3B_BN; }

int gemm_bf16x3(const void* A3, const void* W3, const float* bias, const float* R, float* C, void* C3, int64_t M, int N, int K,
                int act, int terms, hipStream_t st, float* sk_ws, int64_t sk_floats, float ab_scale, float c_scale) {
    AVD_REQUIRE(A3 && W3 && (C || C3), AVD_EINVAL, "gemm_bf16x3: null pointer");
    AVD_REQUIRE(ab_scale > 0.f && ab_scale < __builtin_inff() && c_scale > 0.f && c_scale < __builtin_inff(), AVD_EINVAL,
                "gemm_bf16x3: image scales must be positive and finite");
    AVD_REQUIRE(gemm_bf16x3_supported(M, N, K), AVD_EUNSUPPORTED, "gemm_bf16x3: need N %% 256 == 0 and K %% 16 == 0 (M=%lld N=%d K=%d)",
                (long long)M, N, K);
    AVD_REQUIRE(aligned16(A3) && aligned16(W3) && aligned16(C) && aligned16(C3) && aligned16(bias) && aligned16(R), AVD_EUNSUPPORTED,
                "gemm_bf16x3: pointers must be 16-byte aligned");
    S3Args a{static_cast<const unsigned char*>(A3), static_cast<const unsigned char*>(W3), bias, R, C,
             static_cast<unsigned char*>(C3), M, N, K, 0, 0, 0, 0, 0, 0, 0.f, 0, 0, terms, 1.0f / ab_scale, c_scale, sk_ws, nullptr, 0, sk_floats};
    if (C3) {
        AVD_REQUIRE((act == AVD_ACT_GELU || act == AVD_ACT_NONE) && !R && bias, AVD_EUNSUPPORTED,
                    "gemm_bf16x3: image output implies bias, act NONE or GELU, no residual");
        return act == AVD_ACT_GELU ? launch_s3<S3_EPI_GELU_SPLIT>(a, st) : launch_s3<S3_EPI_SPLIT>(a, st);
    }
    AVD_REQUIRE(act == AVD_ACT_NONE, AVD_EUNSUPPORTED, "gemm_bf16x3: fp32 output supports act NONE only");
    if (R) return launch_s3<S3_EPI_RES>(a, st);
    return launch_s3<S3_EPI_BIAS>(a, st);
}

// in_proj for the bf16x3 attention: qkv = A W^T + bias written as the qkv3 image (q pre-multiplied by qscale)
int gemm_bf16x3_qkv3(const void* A3, const void* W3, const float* bias, void* img, int64_t M, int tokens, int heads, int K, float qscale,
                     int terms, hipStream_t st, float* sk_ws, int64_t sk_floats, float ab_scale, float c_scale) {
    AVD_REQUIRE(A3 && W3 && bias && img, AVD_EINVAL, "gemm_bf16x3_qkv3: null pointer");
    AVD_REQUIRE(ab_scale > 0.f && ab_scale < __builtin_inff() && c_scale > 0.f && c_scale < __builtin_inff(), AVD_EINVAL,
                "gemm_bf16x3_qkv3: image scales must be positive and finite");
    const int N = 3 * heads * 64;
    AVD_REQUIRE(tokens > 0 && heads > 0 && M > 0 && M % tokens == 0, AVD_EINVAL, "gemm_bf16x3_qkv3: rows %lld not a multiple of tokens %d",
                (long long)M, tokens);
    AVD_REQUIRE(gemm_bf16x3_supported(M, N, K), AVD_EUNSUPPORTED, "gemm_bf16x3_qkv3: need 3*heads*64 %% 256 == 0 and K %% 16 == 0");
    AVD_REQUIRE(aligned16(A3) && aligned16(W3) && aligned16(bias) && aligned16(img), AVD_EUNSUPPORTED, "gemm_bf16x3_qkv3: alignment");
    S3Args a{static_cast<const unsigned char*>(A3), static_cast<const unsigned char*>(W3), bias, nullptr, nullptr,
             static_cast<unsigned char*>(img), M, N, K, 0, 0, 0, tokens, qkv3_npad(tokens), heads, qscale, 0, 0, terms, 1.0f / ab_scale, c_scale,
             sk_ws, nullptr, 0, sk_floats};
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
    return gemm_bf16x3(A3, W3, bias, residual, C, C3, M, N, K, act, terms, static_cast<hipStream_t>(stream), nullptr, 0);
}
extern "C" int avd_gemm_bf16x3_qkv3_f32(const void* A3, const void* W3, const float* bias, void* qkv3, int64_t M, int tokens, int heads,
                                        int K, float qscale, int terms, avd_stream_t stream) {
    return gemm_bf16x3_qkv3(A3, W3, bias, qkv3, M, tokens, heads, K, qscale, terms, static_cast<hipStream_t>(stream), nullptr, 0);
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
    return gemm_bf16x3(A2, W2, bias, residual, C, C2, M, N, K, act, 3, static_cast<hipStream_t>(stream), nullptr, 0, ab_scale, c_scale);
}
extern "C" int avd_gemm_f16x2_qkv_f32(const void* A2, const void* W2, const float* bias, void* qkv, int64_t M, int tokens, int heads, int K,
                                      float qscale, float ab_scale, float qkv_scale, avd_stream_t stream) {
    return gemm_bf16x3_qkv3(A2, W2, bias, qkv, M, tokens, heads, K, qscale, 3, static_cast<hipStream_t>(stream), nullptr, 0, ab_scale,
                            qkv_scale);
}
