// Lab v2: fp32-accurate GEMM on the bf16 matrix pipe (three-plane split, six product terms; see split_lab.hip) with a
// PRE-TILED operand format so that one block's K-tile of an operand is one contiguous, already swizzled 12 KiB chunk:
//   X[rows][K] fp32  ->  T[ceil(rows/128)][K/16][128 rows][96 B],  row r of a chunk at r*96, 16-byte slot
//   (2*plane + half) ^ ((r>>3)&1) inside it (plane 0/1/2 = h/m/l, half = (k%16)/8).
// The DMA into LDS is then a pure linear copy (1 KiB per wave instruction, whole cache lines), and ds_read_b128 of
// fragment rows is bank-conflict free.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int CHUNK = 128 * 96;   // bytes of one (row-tile, k-group) chunk

__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
__device__ __forceinline__ unsigned short bf16_rn(float x) {
    unsigned int u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

__device__ __forceinline__ void split8(const float* v, u32x4& H, u32x4& Mi, u32x4& Lo) {
    unsigned short h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        h[e] = bf16_rn(v[e]);
        const float r1 = v[e] - bf16_f(h[e]);
        m[e] = bf16_rn(r1);
        const float r2 = r1 - bf16_f(m[e]);
        l[e] = bf16_rn(r2);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        H[e] = (unsigned int)h[2 * e] | ((unsigned int)h[2 * e + 1] << 16);
        Mi[e] = (unsigned int)m[2 * e] | ((unsigned int)m[2 * e + 1] << 16);
        Lo[e] = (unsigned int)l[2 * e] | ((unsigned int)l[2 * e + 1] << 16);
    }
}

// one thread = 8 consecutive k of one row; rows >= `rows` (padding of the last row tile) are written as zeros
__global__ __launch_bounds__(256) void split_tiled_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int64_t rows,
                                                          int64_t rows_pad, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = K / 8;
    if (i >= rows_pad * per_row) return;
    const int64_t r = i / per_row;
    const int k = (int)(i % per_row) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
        *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(x + r * K + k);
        *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(x + r * K + k + 4);
    }
    u32x4 H, Mi, Lo;
    split8(v, H, Mi, Lo);
    const int rr = (int)(r & 127), f = (rr >> 3) & 1, half = (k >> 3) & 1;
    unsigned char* dst = out + ((r >> 7) * (K / 16) + (k >> 4)) * (int64_t)CHUNK + rr * 96;
    *reinterpret_cast<u32x4*>(dst + (((0 + half) ^ f) << 4)) = H;
    *reinterpret_cast<u32x4*>(dst + (((2 + half) ^ f) << 4)) = Mi;
    *reinterpret_cast<u32x4*>(dst + (((4 + half) ^ f) << 4)) = Lo;
}

struct SArgs {
    const unsigned char* A;   // tiled, ceil(M/128) row tiles
    const unsigned char* W;   // tiled, N/128 row tiles
    const float* bias;
    float* C;
    int64_t M;
    int N, K, nbn, sm, sn;
    unsigned long long* dbg;   // MODE 6: per (block, wave) phase cycle sums
};

template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

template <int BM, int BN, int WM, int WN, int NST, int TERMS, int MODE>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, 1) void sgemm2_kernel(SArgs g) {
    constexpr int ROWB = 96;
    constexpr int WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RA = BM / 128, RB = BN / 128;
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int PIECES = (RA + RB) * 12;
    constexpr int PPW = (PIECES + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int bm, bn;
    if (g.sn > 0) {
        const int per_row = g.sm * g.nbn, per_st = g.sm * g.sn;
        const int srow = wg / per_row, rem = wg % per_row;
        const int sc = rem / per_st, rem2 = rem % per_st;
        bm = srow * g.sm + rem2 / g.sn;
        bn = sc * g.sn + rem2 % g.sn;
        if ((int64_t)bm * BM >= g.M) return;
    } else {
        bm = wg / g.nbn;
        bn = wg % g.nbn;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int ng = g.K / 16;
    const int nrtA = (int)((g.M + 127) >> 7);

    const unsigned char* src[PPW];
    int dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int q = wave * PPW + i;
        q = q < PIECES ? q : PIECES - 1;
        const int region = q / 12, within = (q % 12) * 1024 + lane * 16;
        if (region < RA) {
            int rt = bm * RA + region;
            rt = rt < nrtA ? rt : nrtA - 1;
            src[i] = g.A + (int64_t)rt * ng * CHUNK + within;
        } else {
            const int ct = bn * RB + region - RA;
            src[i] = g.W + (int64_t)ct * ng * CHUNK + within;
        }
        dst[i] = q * 1024;
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + (int64_t)kt * CHUNK), LDS_PTR(smem + buf * STAGE + dst[i]), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        a_off[i] = r * ROWB + ((hi ^ ((r >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = BM + wn * WN + j * 32 + l31;
        b_off[j] = r * ROWB + ((hi ^ ((r >> 3) & 1)) << 4);
    }

    const int nk = ng;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s, s);

    // MODE 4 / 5: software L2 prefetch — every wave touches one dword per 128-byte line of the stage image D K-tiles ahead
    constexpr int PFD = MODE == 4 ? 4 : MODE == 5 ? 6 : 0;
    const unsigned char* pf_base = nullptr;
    if (PFD) {
        int line = wave * 64 + lane;
        line = line < PIECES * 8 ? line : PIECES * 8 - 1;
        const int region = line / 96, within = (line % 96) * 128;
        if (region < RA) {
            int rt = bm * RA + region;
            rt = rt < nrtA ? rt : nrtA - 1;
            pf_base = g.A + (int64_t)rt * ng * CHUNK + within;
        } else {
            pf_base = g.W + (int64_t)(bn * RB + region - RA) * ng * CHUNK + within;
        }
    }
    int cur = 0, nxt = NST - 1;
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    unsigned long long c0 = 0, r0 = 0;
    if (MODE == 6) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int kt = 0; kt < nk; ++kt) {
        unsigned long long tA = 0, tB = 0, tC = 0, tD = 0;
        if (MODE == 6) tA = __builtin_amdgcn_s_memtime();
        if (PFD) {
            if (kt == 0) wait_vm<(NST - 2) * PPW>();
            else if (kt + NST - 1 <= nk) wait_vm<(NST - 2) * (PPW + 1)>();
            else wait_vm<0>();
        } else {
            if (kt + NST - 1 <= nk) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>();
        }
        if (MODE == 6) { tB = __builtin_amdgcn_s_memtime(); }
        asm volatile("s_barrier" ::: "memory");
        if (MODE == 6) { tC = __builtin_amdgcn_s_memtime(); ph[0] += tB - tA; ph[1] += tC - tB; }
        if (PFD) {
            const int kp = kt + PFD < nk ? kt + PFD : nk - 1;
            // a 4-byte-per-lane LDS-DMA into a scratch area: touches the line without a destination VGPR (an asynchronous load into
            // a register the compiler considers free would clobber whatever it allocates there next)
            __builtin_amdgcn_global_load_lds(GLB_PTR(pf_base + (int64_t)kp * CHUNK), LDS_PTR(smem + NST * STAGE + wave * 256), 4, 0, 0);
        }
        if (MODE != 1 && MODE != 3 && kt + NST - 1 < nk) issue(kt + NST - 1, nxt);
        const unsigned char* st = smem + cur * STAGE;
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
        if (MODE == 2) continue;
        bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + 32 * p);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + 32 * p);
        if (MODE == 6) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tD = __builtin_amdgcn_s_memtime();
            ph[2] += tD - tC;
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 6 - TERMS; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (MODE == 3) {
                        // shape experiment: the same flops as two v_mfma_f32_16x16x32_bf16 on the same operand registers
                        f32x4 q0 = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                        f32x4 q1 = {acc[i][j][4 + 4 * (t & 1)], acc[i][j][5 + 4 * (t & 1)], acc[i][j][6 + 4 * (t & 1)], acc[i][j][7 + 4 * (t & 1)]};
                        q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][PA[t]], bf[j][PB[t]], q0, 0, 0, 0);
                        q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][PA[t]], bf[j][PB[t]], q1, 0, 0, 0);
                        acc[i][j][0] = q0[0]; acc[i][j][1] = q0[1]; acc[i][j][2] = q0[2]; acc[i][j][3] = q0[3];
                        acc[i][j][4 + 4 * (t & 1)] = q1[0]; acc[i][j][5 + 4 * (t & 1)] = q1[1];
                        acc[i][j][6 + 4 * (t & 1)] = q1[2]; acc[i][j][7 + 4 * (t & 1)] = q1[3];
                    } else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
        if (MODE == 6) { const unsigned long long tE = __builtin_amdgcn_s_memtime(); ph[3] += tE - tD; }
    }
    if (MODE == 6) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        __syncthreads();
        ph[4] = __builtin_amdgcn_s_memtime() - t0;
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && g.dbg) {
#pragma unroll
            for (int i = 0; i < 5; ++i) g.dbg[((int64_t)blockIdx.x * NW + wave) * 5 + i] = ph[i];
            if (wave == 0) {       // main-loop clock of this block: shader cycles per 100 MHz tick
                g.dbg[(int64_t)gridDim.x * NW * 5 + blockIdx.x * 2] = c1 - c0;
                g.dbg[(int64_t)gridDim.x * NW * 5 + blockIdx.x * 2 + 1] = r1 - r0;
            }
        }
    } else
    __syncthreads();

    // epilogue: per wave, 64-row passes through a private LDS slab, streamed out as 16-byte row segments (+bias)
    constexpr int SUBM = WM < 64 ? WM : 64;
    constexpr int CLD = WN + 4;
    float* slab = reinterpret_cast<float*>(smem) + wave * SUBM * CLD;
    constexpr int LPR = WN / 4, RPI = 64 / LPR, NIT = SUBM / RPI;
    const int cr = lane / LPR, cc = (lane % LPR) * 4;
    const int n = bn * BN + wn * WN + cc;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n < g.N) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
    for (int ps = 0; ps < WM / SUBM; ++ps) {
#pragma unroll
        for (int i = 0; i < SUBM / 32; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[ps * (SUBM / 32) + i][j][r];
        const int64_t mbase = (int64_t)bm * BM + wm * WM + ps * SUBM + cr;
        float* cptr = g.C + mbase * g.N + n;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 v = *reinterpret_cast<const f32x4*>(slab + (cr + it * RPI) * CLD + cc);
            v += bv;
            if (n < g.N && mbase + (int64_t)it * RPI < g.M) *reinterpret_cast<f32x4*>(cptr + (int64_t)it * RPI * g.N) = v;
        }
    }
}

template <int BM, int BN, int WM, int WN, int NST, int TERMS, int MODE = 0>
static float run_gemm(const SArgs& a0, int iters, int super = 0) {
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int SUBM = WM < 64 ? WM : 64;
    constexpr int stage_lds = NST * (BM + BN) * 96, epi_lds = NW * SUBM * (WN + 4) * 4;
    constexpr int lds = (stage_lds > epi_lds ? stage_lds : epi_lds) + ((MODE == 4 || MODE == 5) ? 2048 : 0);
    static_assert(lds <= 160 * 1024, "LDS");
    auto kern = sgemm2_kernel<BM, BN, WM, WN, NST, TERMS, MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SArgs a = a0;
    a.nbn = (a.N + BN - 1) / BN;
    const int nbm = (int)((a.M + BM - 1) / BM);
    unsigned nwg = (unsigned)(nbm * a.nbn);
    a.sm = a.sn = 0;
    if (super > 0) {
        int sn = 8;
        while (a.nbn % sn) sn >>= 1;
        a.sn = sn;
        a.sm = super / sn;
        nwg = (unsigned)(((nbm + a.sm - 1) / a.sm) * a.sm * a.nbn);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * NW), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * NW), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}


// ---------------------------------------------------------------------------------------------------------
// v3: 256x128 block, 4 waves (wave tile 128x64), two blocks per CU, two LDS stages + fragments kept one K-tile ahead in
// registers: while the 48 MFMAs of tile kt run, the 18 fragment reads of tile kt+1 are spread between them and the DMA
// of tile kt+2 is in flight.  Term order is chosen so that every operand plane is dead before its successor is read.
// ---------------------------------------------------------------------------------------------------------
template <int TERMS, int MODE>
__global__ __launch_bounds__(256, 2) void sgemm3_kernel(SArgs g) {
    constexpr int BM = 256, BN = 128, WM = 128, WN = 64, ROWB = 96;
    constexpr int TM = 4, TN = 2;
    constexpr int STAGE = (BM + BN) * ROWB;      // 36 KiB
    constexpr int PPW = 9;                        // 36 pieces / 4 waves
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int bm, bn;
    if (g.sn > 0) {
        const int per_row = g.sm * g.nbn, per_st = g.sm * g.sn;
        const int srow = wg / per_row, rem = wg % per_row;
        const int sc = rem / per_st, rem2 = rem % per_st;
        bm = srow * g.sm + rem2 / g.sn;
        bn = sc * g.sn + rem2 % g.sn;
        if ((int64_t)bm * BM >= g.M) return;
    } else {
        bm = wg / g.nbn;
        bn = wg % g.nbn;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ng = g.K / 16;
    const int nrtA = (int)((g.M + 127) >> 7);

    // DMA: wave w copies pieces 9w..9w+8 of the 36-piece stage image [A rt0 | A rt1 | W ct]
    const unsigned char* src[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int region = q / 12, within = (q % 12) * 1024 + lane * 16;
        if (region < 2) {
            int rt = bm * 2 + region;
            rt = rt < nrtA ? rt : nrtA - 1;
            src[i] = g.A + (int64_t)rt * ng * CHUNK + within;
        } else {
            src[i] = g.W + (int64_t)bn * ng * CHUNK + within;
        }
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + (int64_t)kt * CHUNK), LDS_PTR(smem + buf * STAGE + (wave * PPW + i) * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_off[TM], b_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        a_off[i] = r * ROWB + ((hi ^ ((r >> 3) & 1)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = BM + wn * WN + j * 32 + l31;
        b_off[j] = r * ROWB + ((hi ^ ((r >> 3) & 1)) << 4);
    }
#define LDA(dst, st, p) _Pragma("unroll") for (int i = 0; i < TM; ++i) dst[i] = *reinterpret_cast<const bf16x8*>((st) + a_off[i] + 32 * (p))
#define LDB(dst, st, p) _Pragma("unroll") for (int j = 0; j < TN; ++j) dst[j] = *reinterpret_cast<const bf16x8*>((st) + b_off[j] + 32 * (p))
#define MM(A_, B_) _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[i], B_[j], acc[i][j], 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

    const int nk = ng;
    issue(0, 0);
    wait_vm<0>();
    asm volatile("s_barrier" ::: "memory");
    if (nk > 1) issue(1, 1);
    bf16x8 ah[TM], am[TM], al[TM], bh[TN], bmm[TN], bl[TN];
    LDA(ah, smem, 0); LDA(am, smem, 1); LDA(al, smem, 2);
    LDB(bh, smem, 0); LDB(bmm, smem, 1); LDB(bl, smem, 2);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt+1 has landed in stage (kt+1)&1; stage kt&1 (read during the previous iteration) is free for tile kt+2
        wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE != 1 && kt + 2 < nk) issue(kt + 2, kt & 1);
        const unsigned char* nx = smem + ((kt + 1) & 1) * STAGE;
        bf16x8 ah_n[TM], bl_n[TN];
        if (MODE == 2) continue;
        FENCE();
        MM(am, bmm);                       // 1: (m,m)
        LDA(ah_n, nx, 0);
        FENCE();
        if (TERMS > 1) { MM(am, bh); }     // 2: (m,h)   -> am dead
        LDA(am, nx, 1);
        FENCE();
        if (TERMS > 1) { MM(al, bh); }     // 3: (l,h)   -> al dead
        LDA(al, nx, 2);
        FENCE();
        MM(ah, bh);                        // 4: (h,h)   -> bh dead
        LDB(bh, nx, 0);
        LDB(bl_n, nx, 2);
        FENCE();
        if (TERMS > 1) { MM(ah, bmm); }    // 5: (h,m)   -> bmm dead
        LDB(bmm, nx, 1);
        FENCE();
        if (TERMS > 1) { MM(ah, bl); }     // 6: (h,l)   -> ah, bl dead
        FENCE();
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[i] = ah_n[i];
#pragma unroll
        for (int j = 0; j < TN; ++j) bl[j] = bl_n[j];
    }
    __syncthreads();

    // epilogue: two 64-row passes per wave through a private LDS slab
    constexpr int CLD = WN + 4;
    float* slab = reinterpret_cast<float*>(smem) + wave * 64 * CLD;
    constexpr int LPR = WN / 4, RPI = 64 / LPR, NIT = 64 / RPI;
    const int cr = lane / LPR, cc = (lane % LPR) * 4;
    const int n = bn * BN + wn * WN + cc;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n < g.N) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[ps * 2 + i][j][r];
        const int64_t mbase = (int64_t)bm * BM + wm * WM + ps * 64 + cr;
        float* cptr = g.C + mbase * g.N + n;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            f32x4 v = *reinterpret_cast<const f32x4*>(slab + (cr + it * RPI) * CLD + cc);
            v += bv;
            if (n < g.N && mbase + (int64_t)it * RPI < g.M) *reinterpret_cast<f32x4*>(cptr + (int64_t)it * RPI * g.N) = v;
        }
    }
}

template <int TERMS, int MODE = 0>
static float run_gemm3(const SArgs& a0, int iters, int super = 0) {
    constexpr int BM = 256, BN = 128;
    constexpr int lds = 2 * (BM + BN) * 96;
    auto kern = sgemm3_kernel<TERMS, MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SArgs a = a0;
    a.nbn = (a.N + BN - 1) / BN;
    const int nbm = (int)((a.M + BM - 1) / BM);
    unsigned nwg = (unsigned)(nbm * a.nbn);
    a.sm = a.sn = 0;
    if (super > 0) {
        int sn = 8;
        while (a.nbn % sn) sn >>= 1;
        a.sn = sn;
        a.sm = super / sn;
        nwg = (unsigned)(((nbm + a.sm - 1) / a.sm) * a.sm * a.nbn);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

static void fill(std::vector<float>& v, unsigned seed, float scale) {
    uint64_t s = seed * 6364136223846793005ull + 1442695040888963407ull;
    for (auto& x : v) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double u = (double)((s >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53);
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const double w = (double)((s >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53);
        x = (float)(scale * sqrt(-2.0 * log(u + 1e-300)) * cos(6.283185307179586 * w));
    }
}

static void do_split(const float* x, unsigned char* out, int64_t rows, int K) {
    const int64_t rows_pad = (rows + 127) / 128 * 128;
    const int64_t n = rows_pad * (K / 8);
    hipLaunchKernelGGL(split_tiled_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, out, rows, rows_pad, K);
}

static int check(int M, int N, int K, float scaleA) {
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hC((size_t)M * N);
    fill(hA, 1, scaleA); fill(hW, 2, 1.f / sqrtf((float)K)); fill(hb, 3, 1.f);
    float *dA, *dW, *db, *dC; unsigned char *dAs, *dWs;
    const size_t Mp = (M + 127) / 128 * 128, Np = (N + 255) / 256 * 256;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&db, N * 4));
    CK(hipMalloc(&dC, hC.size() * 4)); CK(hipMalloc(&dAs, Mp * K * 6)); CK(hipMalloc(&dWs, Np * K * 6));
    CK(hipMemset(dWs, 0, Np * K * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
    do_split(dA, dAs, M, K); do_split(dW, dWs, N, K);
    SArgs a{dAs, dWs, db, dC, M, N, K, 0, 0, 0, nullptr};
    std::vector<double> ref((size_t)M * N);
    double max_ref = 0, max_f32 = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double r = hb[n];
            float f32 = 0.f;
            for (int k = 0; k < K; ++k) {
                r += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
                f32 = fmaf(hA[(size_t)m * K + k], hW[(size_t)n * K + k], f32);
            }
            f32 += hb[n];
            ref[(size_t)m * N + n] = r;
            max_ref = fmax(max_ref, fabs(r));
            max_f32 = fmax(max_f32, fabs((double)f32 - r));
        }
    int bad = 0;
    for (int cfg = 0; cfg < 6; ++cfg) {
        CK(hipMemset(dC, 0xff, hC.size() * 4));
        if (cfg == 4) run_gemm3<6>(a, 1);
        else if (cfg == 5) run_gemm3<6>(a, 1, 64);
        else if (cfg == 0) run_gemm<256, 256, 128, 64, 3, 6>(a, 1);
        else if (cfg == 1) run_gemm<256, 128, 64, 64, 3, 6>(a, 1, 32);
        else if (cfg == 2) run_gemm<128, 128, 64, 64, 3, 6>(a, 1, 64);
        else run_gemm<256, 256, 128, 64, 3, 6, 4>(a, 1, 16);
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double max_err = 0;
        for (size_t i = 0; i < hC.size(); ++i) {
            const double e = fabs((double)hC[i] - ref[i]);
            max_err = e > max_err || e != e ? (e != e ? 1e30 : e) : max_err;
        }
        printf("check cfg=%d M=%d N=%d K=%d scaleA=%g: max|err| %.3e (fp32 FMA chain %.3e)  max|ref| %.3e  rel %.3e\n", cfg, M, N, K,
               scaleA, max_err, max_f32, max_ref, max_err / max_ref);
        if (!(max_err / max_ref < 2e-6)) bad = 1;
    }
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(db)); CK(hipFree(dC)); CK(hipFree(dAs)); CK(hipFree(dWs));
    return bad;
}

int main(int argc, char** argv) {
    if (argc > 1) {
        int bad = 0;
        bad |= check(200, 256, 512, 1.f);
        bad |= check(333, 384, 2048, 100.f);
        bad |= check(700, 512, 256, 1.f);
        printf(bad ? "CHECK FAILED\n" : "CHECK OK\n");
        return bad;
    }
    const int64_t M = 64 * 421, Mp = (M + 127) / 128 * 128;
    const int KMAX = 2048, NMAX = 2048;
    float *dA, *dW, *db, *dC; unsigned char *dAs, *dWs;
    CK(hipMalloc(&dA, (size_t)M * KMAX * 4)); CK(hipMalloc(&dW, (size_t)NMAX * KMAX * 4)); CK(hipMalloc(&db, NMAX * 4));
    CK(hipMalloc(&dC, (size_t)M * NMAX * 4)); CK(hipMalloc(&dAs, (size_t)Mp * KMAX * 6)); CK(hipMalloc(&dWs, (size_t)NMAX * KMAX * 6));
    {
        std::vector<float> h((size_t)M * KMAX);
        fill(h, 5, 1.f);
        CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, h.data(), (size_t)NMAX * KMAX * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, h.data(), NMAX * 4, hipMemcpyHostToDevice));
    }
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 1536, 512}, {"out_proj", 512, 512}, {"fc1", 2048, 512}, {"fc2", 512, 2048}};
    {
        do_split(dA, dAs, M, 512); do_split(dW, dWs, 2048, 512);
        SArgs a{dAs, dWs, db, dC, M, 2048, 512, 0, 0, 0, nullptr};
        run_gemm<256, 256, 128, 64, 3, 6>(a, 300);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (auto& s : shapes) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            do_split(dA, dAs, M, s.K);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float sms; CK(hipEventElapsedTime(&sms, e0, e1));
            do_split(dW, dWs, s.N, s.K);
            SArgs a{dAs, dWs, db, dC, M, s.N, s.K, 0, 0, 0, nullptr};
            const double fl = 2.0 * (double)M * s.N * s.K;
            const int it = 100;
            float t;
#define RUN(label, expr) t = (expr); printf(" %s %6.1f us %6.1f TF |", label, t * 1e3, fl / t / 1e9)
            printf("%-9s N=%4d K=%4d  split(A) %.1f us\n    256x256:", s.name, s.N, s.K, sms * 1e3);
            RUN("s3", (run_gemm<256, 256, 128, 64, 3, 6>(a, it)));
            RUN("s3 super16", (run_gemm<256, 256, 128, 64, 3, 6>(a, it, 16)));
            RUN("noDMA", (run_gemm<256, 256, 128, 64, 3, 6, 1>(a, it, 16)));
            RUN("DMAonly", (run_gemm<256, 256, 128, 64, 3, 6, 2>(a, it, 16)));
            RUN("noDMA 16x16x32", (run_gemm<256, 256, 128, 64, 3, 6, 3>(a, it, 16)));
            RUN("PF4", (run_gemm<256, 256, 128, 64, 3, 6, 4>(a, it, 16)));
            RUN("PF6", (run_gemm<256, 256, 128, 64, 3, 6, 5>(a, it, 16)));
            {
                unsigned long long* dbg; const int nblk = 1024;
                CK(hipMalloc(&dbg, (size_t)nblk * 8 * 6 * 8)); CK(hipMemset(dbg, 0, (size_t)nblk * 8 * 6 * 8));
                SArgs ad = a; ad.dbg = dbg;
                run_gemm<256, 256, 128, 64, 3, 6, 6>(ad, 200, 16);     // ~50 ms of back-to-back launches: the clock has settled
                std::vector<unsigned long long> h((size_t)nblk * 8 * 6);
                CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
                double sum[5] = {0, 0, 0, 0, 0}; int cnt = 0;
                for (int b = 0; b < nblk; ++b) for (int w = 0; w < 8; ++w) {
                    const unsigned long long* p = &h[((size_t)b * 8 + w) * 5];
                    if (p[3] == 0) continue;
                    for (int i = 0; i < 5; ++i) sum[i] += (double)p[i];
                    ++cnt;
                }
                const double steps = s.K / 16.0;
                printf("\n    phases (memtime ticks per K-step, mean over %d waves): wait_vm %.0f | barrier %.0f | dma-issue+frag-reads %.0f | mfma %.0f | final-sync %.0f (per block)",
                       cnt, sum[0] / cnt / steps, sum[1] / cnt / steps, sum[2] / cnt / steps, sum[3] / cnt / steps, sum[4] / cnt);
                {
                    // grid size used by run_gemm with super = 16 (same formula)
                    const int nbn_ = s.N / 256, nbm_ = (int)((M + 255) / 256);
                    int sn_ = 8; while (nbn_ % sn_) sn_ >>= 1;
                    const int sm_ = 16 / sn_;
                    const int grid = (nbm_ + sm_ - 1) / sm_ * sm_ * nbn_;
                    double cs = 0, rs = 0; int nb = 0;
                    for (int b = 0; b < grid && b < nblk; ++b) {
                        const unsigned long long c = h[(size_t)grid * 8 * 5 + b * 2], r = h[(size_t)grid * 8 * 5 + b * 2 + 1];
                        if (r == 0) continue;
                        cs += (double)c; rs += (double)r; ++nb;
                    }
                    if (nb) printf("\n    in-kernel clock over the main loop (s_memtime / s_memrealtime x 100 MHz, %d blocks): %.2f GHz", nb, cs / rs * 0.1);
                }
                CK(hipFree(dbg));
            }
            printf("\n    256x128:");
            RUN("s3", (run_gemm<256, 128, 64, 64, 3, 6>(a, it)));
            RUN("s4 super32", (run_gemm<256, 128, 64, 64, 4, 6>(a, it, 32)));
            RUN("noDMA", (run_gemm<256, 128, 64, 64, 4, 6, 1>(a, it, 32)));
            RUN("DMAonly", (run_gemm<256, 128, 64, 64, 4, 6, 2>(a, it, 32)));
            printf("\n    v3 256x128 4w x2:");
            RUN("plain", (run_gemm3<6>(a, it)));
            RUN("super64", (run_gemm3<6>(a, it, 64)));
            RUN("super32", (run_gemm3<6>(a, it, 32)));
            RUN("noDMA", (run_gemm3<6, 1>(a, it, 64)));
            RUN("DMAonly", (run_gemm3<6, 2>(a, it, 64)));
            RUN("1-term", (run_gemm3<1>(a, it, 64)));
            printf("\n    128x128:");
            RUN("s3 super64", (run_gemm<128, 128, 64, 64, 3, 6>(a, it, 64)));
            RUN("8w(64x32) s3 super64", (run_gemm<128, 128, 64, 32, 3, 6>(a, it, 64)));
            RUN("DMAonly", (run_gemm<128, 128, 64, 64, 3, 6, 2>(a, it, 64)));
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
