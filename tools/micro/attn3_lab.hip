// Lab: fp32-accurate attention on the bf16 matrix pipe (QK^T and PV with exactly split operands, six terms each).
//   attn3_lab check   -> small shapes against an fp64 host reference
//   attn3_lab         -> timing at the C3 shape (B=64, N=421, H=8)
// qkv3 image: for part in {q,k,v}, sample b, head h: rows n in [0,Npad) of 384 B = [plane h|m|l][64 d bf16]; the 16-byte
// chunk c (8 d) of a row sits at slot c ^ sw_part(n): sw_q = 0, sw_k = (n>>1)&7 (conflict-free ds_read_b128 of 32 key
// rows), sw_v = ((n>>1)&1)<<2 (conflict-free ds_read_b64_tr_b16 of 4-key x 16-d blocks).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int DH = 64, KT = 64, ROWB = 384;
constexpr float NEG = -1.0e30f, LOG2E = 1.4426950408889634f;

__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ unsigned int pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }
__device__ __forceinline__ void split8(const float* v, u32x4& H, u32x4& Mi, u32x4& Lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = v[2 * e], b = v[2 * e + 1];
        const unsigned int h = pk_bf16(a, b);
        const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
        const unsigned int m = pk_bf16(ra, rb);
        const float sa = ra - bf16_lo(m), sb = rb - bf16_hi(m);
        H[e] = h; Mi[e] = m; Lo[e] = pk_bf16(sa, sb);
    }
}

__device__ __forceinline__ int sw_part(int part, int n) { return part == 0 ? 0 : part == 1 ? (n >> 1) & 7 : ((n >> 1) & 1) << 2; }

// packed qkv fp32 [Bt, N, 3*H*64] -> qkv3 image; q is pre-scaled by qscale (= softmax scale * log2 e)
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ qkv, unsigned char* __restrict__ img, int Bt, int N, int Npad,
                                                   int H, float qscale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread = 8 d of one (b, n, part, h)
    const int64_t total = (int64_t)Bt * N * 3 * H * 8;
    if (i >= total) return;
    const int c = (int)(i & 7);
    const int h = (int)((i >> 3) % H);
    const int part = (int)((i / (8 * H)) % 3);
    const int64_t bn = i / (24 * H);
    const int n = (int)(bn % N), b = (int)(bn / N);
    const float* src = qkv + bn * (3 * H * DH) + part * H * DH + h * DH + c * 8;
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + 4);
    if (part == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= qscale;
    }
    u32x4 Hh, Mi, Lo;
    split8(v, Hh, Mi, Lo);
    unsigned char* dst = img + ((((int64_t)part * Bt + b) * H + h) * Npad + n) * ROWB + ((c ^ sw_part(part, n)) << 4);
    *reinterpret_cast<u32x4*>(dst) = Hh;
    *reinterpret_cast<u32x4*>(dst + 128) = Mi;
    *reinterpret_cast<u32x4*>(dst + 256) = Lo;
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn3_kernel(const unsigned char* __restrict__ img, float* __restrict__ out, int Bt, int N,
                                                           int Npad, int H, int n_query, int nqb) {
    constexpr int PPW = 24 / NW;                       // 1-KiB DMA pieces per wave per operand tile (24 KiB)
    __shared__ __attribute__((aligned(16))) unsigned char Ks[KT * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[KT * ROWB];

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
    const int64_t hstride = (int64_t)Npad * ROWB;
    const unsigned char* Qb = img + (((int64_t)0 * Bt + b) * H + h) * hstride;
    const unsigned char* Kb = img + (((int64_t)1 * Bt + b) * H + h) * hstride;
    const unsigned char* Vb = img + (((int64_t)2 * Bt + b) * H + h) * hstride;

    // Q fragments: lane (q = l31, half hi), d-step s: Q[q][16 s + 8 hi .. +7] of each plane
    const int q_row = qb * (NW * 32) + wave * 32 + l31;
    bf16x8 qf[4][3];
    {
        const unsigned char* src = Qb + (int64_t)(q_row < N ? q_row : N - 1) * ROWB + hi * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) qf[s][p] = *reinterpret_cast<const bf16x8*>(src + p * 128 + s * 32);
    }

    auto dma = [&](const unsigned char* gsrc, unsigned char* ldst, int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc + (int64_t)kt * KT * ROWB + p * 1024 + lane * 16), LDS_PTR(ldst + p * 1024), 16, 0, 0);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = NEG, l_run = 0.f;

    const int nkt = (N + KT - 1) / KT;
    dma(Kb, Ks, 0);
    dma(Vb, Vs, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    // K row reads: key = 32 kb + l31, chunk (2 s + hi) ^ ((key>>1)&7)
    const int ksw = (l31 >> 1) & 7;                 // same for key and key + 32
    const int k_rd = l31 * ROWB;
    // V transposed reads: 16-lane group g = lane>>4: d columns 16 (g&1) .. +15 of a 32-d block, keys 4 hi + (i>>2) (+8)
    const int i16 = lane & 15, cb = (lane >> 4) & 1;
    const int v_q = i16 >> 2, v_p = i16 & 3;
    // byte offset inside a row for (d block db): chunk = 4 db + 2 cb + (v_p>>1), + 8 (v_p&1); swizzle by key row applied per read

    const bool active = qb * (NW * 32) + wave * 32 < n_query;     // wave-uniform: a wave whose 32 rows are all padding only loads
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
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
        {
            bf16x8 kf[2][2][3];                       // [buffer][key block][plane]: fragments of d-step s+1 are read under step s
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int ch = ((0 + hi) ^ ksw) << 4;
                kf[0][0][p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + p * 128 + ch);
                kf[0][1][p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + 32 * ROWB + p * 128 + ch);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s < 3) {
                    const int ch = ((2 * (s + 1) + hi) ^ ksw) << 4;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        kf[(s + 1) & 1][0][p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + p * 128 + ch);
                        kf[(s + 1) & 1][1][p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + 32 * ROWB + p * 128 + ch);
                    }
                }
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s & 1][0][PA[t]], qf[s][PB[t]], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s & 1][1][PA[t]], qf[s][PB[t]], s1, 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma(Kb, Ks, kt + 1);

        if (!more && (N & (KT - 1))) {
            const int kbase = kt * KT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + mfma32_row(r, hi);
                if (key >= N) s0[r] = NEG;
                if (key + 32 >= N) s1[r] = NEG;
            }
        }
        float mt = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s1[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
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

        // O^T += V^T P^T: k-step (kb, t) covers keys 32 kb + 16 t + {4 hi + (j&3) + 8 (j>>2)}, j = element of the fragment.
        // The eight (kb, t, db) steps are software-pipelined: the V fragments of step i+1 are read under the MFMAs of step i.
        auto load_v = [&](int step, bf16x8* vf) {
            const int kb = step >> 2, t = (step >> 1) & 1, db = step & 1;
            const int key0 = 32 * kb + 16 * t + 4 * hi + v_q;
            const int sw = ((key0 >> 1) & 1) << 2;
            const int chunk = 4 * db + 2 * cb + (v_p >> 1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)LDS_PTR(Vs + key0 * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)LDS_PTR(Vs + (key0 + 8) * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                u32x4 w;
                const u32x2 a = __builtin_bit_cast(u32x2, lo), c2 = __builtin_bit_cast(u32x2, hi4);
                w[0] = a[0]; w[1] = a[1]; w[2] = c2[0]; w[3] = c2[1];
                vf[p] = __builtin_bit_cast(bf16x8, w);
            }
        };
        bf16x8 vfb[2][3];
        load_v(0, vfb[0]);
        bf16x8 pf[3];
#pragma unroll
        for (int step = 0; step < 8; ++step) {
            const int kb = step >> 2, t = (step >> 1) & 1, db = step & 1;
            if (db == 0) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = kb ? s1[8 * t + j] : s0[8 * t + j];
                u32x4 P[3];
                split8(pv, P[0], P[1], P[2]);
#pragma unroll
                for (int p = 0; p < 3; ++p) pf[p] = __builtin_bit_cast(bf16x8, P[p]);
            }
            if (step < 7) load_v(step + 1, vfb[(step + 1) & 1]);
#pragma unroll
            for (int tt = 0; tt < 6; ++tt) {
                if (db == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[step & 1][PA[tt]], pf[PB[tt]], o0, 0, 0, 0);
                else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfb[step & 1][PA[tt]], pf[PB[tt]], o1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma(Vb, Vs, kt + 1);
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < n_query) {
        const int d = H * DH;
        float* dst = out + ((int64_t)b * N + q_row) * d + h * DH + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<f32x4*>(dst + 8 * g4) = a;
            *reinterpret_cast<f32x4*>(dst + 32 + 8 * g4) = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// ring variant: 32-key tiles, three LDS stages (K + V of a tile = 24 KiB), DMA two tiles ahead behind a counted vmcnt,
// one barrier per tile
// ---------------------------------------------------------------------------------------------------------
template <int N_> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N_ & 15) | ((N_ >> 4) << 14)); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn3r_kernel(const unsigned char* __restrict__ img, float* __restrict__ out, int Bt, int N,
                                                            int Npad, int H, int n_query, int nqb) {
    constexpr int KT2 = 32, NST = 3;
    constexpr int TILEB = KT2 * ROWB;                 // 12 KiB per operand
    constexpr int STAGE = 2 * TILEB;                  // K then V
    constexpr int PPW = 24 / NW;                      // 24 one-KiB pieces per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];

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
    const int64_t hstride = (int64_t)Npad * ROWB;
    const unsigned char* Qb = img + (((int64_t)0 * Bt + b) * H + h) * hstride;
    const unsigned char* Kb = img + (((int64_t)1 * Bt + b) * H + h) * hstride;
    const unsigned char* Vb = img + (((int64_t)2 * Bt + b) * H + h) * hstride;

    const int q_row = qb * (NW * 32) + wave * 32 + l31;
    bf16x8 qf[4][3];
    {
        const unsigned char* src = Qb + (int64_t)(q_row < N ? q_row : N - 1) * ROWB + hi * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) qf[s][p] = *reinterpret_cast<const bf16x8*>(src + p * 128 + s * 32);
    }
    auto dma = [&](int kt, int buf) {
        const int last_row = N - 1 - kt * KT2;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;                       // 0..11 K, 12..23 V
            const int pp = p < 12 ? p : p - 12;
            const int off = pp * 1024 + lane * 16;
            int row = off / ROWB;
            const int within = off - row * ROWB;
            row = row < last_row ? row : last_row;
            const unsigned char* g = (p < 12 ? Kb : Vb) + ((int64_t)kt * KT2 + row) * ROWB + within;
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(sm + buf * STAGE + p * 1024), 16, 0, 0);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = NEG, l_run = 0.f;
    const int nkt = (N + KT2 - 1) / KT2;
    dma(0, 0);
    if (nkt > 1) dma(1, 1);

    const int ksw = (l31 >> 1) & 7;
    const int k_rd = l31 * ROWB;
    const int i16 = lane & 15, cb = (lane >> 4) & 1;
    const int v_q = i16 >> 2, v_p = i16 & 3;
    const bool active = qb * (NW * 32) + wave * 32 < n_query;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

    int cur = 0, nxt = 2;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) wait_vm<PPW>(); else wait_vm<0>();
        asm volatile("s_barrier" ::: "memory");
        if (kt + 2 < nkt) dma(kt + 2, nxt);
        const unsigned char* Ks = sm + cur * STAGE;
        const unsigned char* Vs = Ks + TILEB;
        cur = cur == 2 ? 0 : cur + 1;
        nxt = nxt == 2 ? 0 : nxt + 1;
        if (!active) continue;

        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 ka[3];
            const int ch = ((2 * s + hi) ^ ksw) << 4;
#pragma unroll
            for (int p = 0; p < 3; ++p) ka[p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + p * 128 + ch);
#pragma unroll
            for (int t = 0; t < 6; ++t) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[PA[t]], qf[s][PB[t]], s0, 0, 0, 0);
        }
        if (kt == nkt - 1 && (N & (KT2 - 1))) {
            const int kbase = kt * KT2;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (kbase + mfma32_row(r, hi) >= N) s0[r] = NEG;
        }
        float mt = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s0[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            ps += s0[r];
        }
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        l_run += ps;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = s0[8 * t + j];
            u32x4 P[3];
            split8(pv, P[0], P[1], P[2]);
            bf16x8 pf[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) pf[p] = __builtin_bit_cast(bf16x8, P[p]);
            const int key0 = 16 * t + 4 * hi + v_q;
            const int sw = ((key0 >> 1) & 1) << 2;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int chunk = 4 * db + 2 * cb + (v_p >> 1);
                bf16x8 vf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)LDS_PTR(Vs + key0 * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                    const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)LDS_PTR(Vs + (key0 + 8) * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                    const u32x2 a = __builtin_bit_cast(u32x2, lo), c2 = __builtin_bit_cast(u32x2, up);
                    const u32x4 w = {a[0], a[1], c2[0], c2[1]};
                    vf[p] = __builtin_bit_cast(bf16x8, w);
                }
#pragma unroll
                for (int tt = 0; tt < 6; ++tt) {
                    if (db == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[PA[tt]], pf[PB[tt]], o0, 0, 0, 0);
                    else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[PA[tt]], pf[PB[tt]], o1, 0, 0, 0);
                }
            }
        }
    }
    if (!active) return;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q_row < n_query) {
        const int d = H * DH;
        float* dst = out + ((int64_t)b * N + q_row) * d + h * DH + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<f32x4*>(dst + 8 * g4) = a;
            *reinterpret_cast<f32x4*>(dst + 32 + 8 * g4) = c;
        }
    }
}

template <int NW>
static float run_ring(const unsigned char* img, float* dout, int Bt, int N, int H, int iters) {
    const int Npad = (N + 63) / 64 * 64;
    const int nqb = (N + 32 * NW - 1) / (32 * NW);
    constexpr int lds = 3 * 2 * 32 * ROWB;
    auto kern = attn3r_kernel<NW>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nqb * H * Bt), dim3(64 * NW), lds, 0, img, dout, Bt, N, Npad, H, N, nqb);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(nqb * H * Bt), dim3(64 * NW), lds, 0, img, dout, Bt, N, Npad, H, N, nqb);
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

template <int NW>
static float run(const float* dqkv, unsigned char* img, float* dout, int Bt, int N, int H, int iters, bool with_prep) {
    const int Npad = (N + 63) / 64 * 64;
    const int64_t total = (int64_t)Bt * N * 3 * H * 8;
    const int nqb = (N + 32 * NW - 1) / (32 * NW);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dqkv, img, Bt, N, Npad, H, 0.125f * LOG2E);
    hipLaunchKernelGGL(attn3_kernel<NW>, dim3(nqb * H * Bt), dim3(64 * NW), 0, 0, img, dout, Bt, N, Npad, H, N, nqb);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) {
        if (with_prep) hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, dqkv, img, Bt, N, Npad, H, 0.125f * LOG2E);
        hipLaunchKernelGGL(attn3_kernel<NW>, dim3(nqb * H * Bt), dim3(64 * NW), 0, 0, img, dout, Bt, N, Npad, H, N, nqb);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main(int argc, char** argv) {
    const bool check = argc > 1;
    const int Bt = check ? 2 : 64, N = check ? 133 : 421, H = check ? 3 : 8, d = H * DH;
    const int Npad = (N + 63) / 64 * 64;
    std::vector<float> h((size_t)Bt * N * 3 * d);
    fill(h, 7, check ? 1.5f : 1.0f);
    float *dqkv, *dout; unsigned char* img;
    const size_t img_bytes = (size_t)3 * Bt * H * Npad * ROWB;
    CK(hipMalloc(&dqkv, h.size() * 4)); CK(hipMalloc(&dout, (size_t)Bt * N * d * 4)); CK(hipMalloc(&img, img_bytes));
    CK(hipMemset(img, 0, img_bytes));
    CK(hipMemcpy(dqkv, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    if (check) {
        if (argv[1][0] == '4') run<4>(dqkv, img, dout, Bt, N, H, 1, true); else run<2>(dqkv, img, dout, Bt, N, H, 1, true);
        if (argv[1][0] == 'r') { CK(hipMemset(dout, 0xff, (size_t)Bt * N * d * 4)); run_ring<4>(img, dout, Bt, N, H, 1); }
        std::vector<float> o((size_t)Bt * N * d);
        CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
        double max_err = 0, max_ref = 0;
        std::vector<double> sc(N);
        for (int b = 0; b < Bt; ++b)
            for (int hh = 0; hh < H; ++hh)
                for (int q = 0; q < N; ++q) {
                    const float* qp = &h[((size_t)b * N + q) * 3 * d + hh * DH];
                    double mx = -1e300;
                    for (int k = 0; k < N; ++k) {
                        const float* kp = &h[((size_t)b * N + k) * 3 * d + d + hh * DH];
                        double s = 0;
                        for (int e = 0; e < DH; ++e) s += (double)qp[e] * kp[e];
                        sc[k] = s * 0.125;
                        mx = fmax(mx, sc[k]);
                    }
                    double den = 0;
                    for (int k = 0; k < N; ++k) { sc[k] = exp(sc[k] - mx); den += sc[k]; }
                    for (int e = 0; e < DH; ++e) {
                        double acc = 0;
                        for (int k = 0; k < N; ++k) acc += sc[k] * h[((size_t)b * N + k) * 3 * d + 2 * d + hh * DH + e];
                        acc /= den;
                        const double got = o[((size_t)b * N + q) * d + hh * DH + e];
                        max_err = fmax(max_err, fabs(got - acc));
                        if (got != got) max_err = 1e30;
                        max_ref = fmax(max_ref, fabs(acc));
                    }
                }
        printf("check Bt=%d N=%d H=%d: max|err| %.3e  max|ref| %.3e  rel %.3e\n", Bt, N, H, max_err, max_ref, max_err / max_ref);
        const bool ok = max_err / max_ref < 5e-6;
        printf(ok ? "CHECK OK\n" : "CHECK FAILED\n");
        return ok ? 0 : 1;
    }
    run<2>(dqkv, img, dout, Bt, N, H, 200, false);   // warm clocks
    const double fl = 4.0 * Bt * H * (double)N * N * DH;
    for (int rep = 0; rep < 3; ++rep) {
        const float t = run<2>(dqkv, img, dout, Bt, N, H, 100, false);
        const float tp = run<2>(dqkv, img, dout, Bt, N, H, 100, true);
        const float t4 = run<4>(dqkv, img, dout, Bt, N, H, 100, false);
        const float tr = run_ring<4>(img, dout, Bt, N, H, 100);
        const float tr2 = run_ring<2>(img, dout, Bt, N, H, 100);
        printf("ring(KT=32, 3 stages): NW=4 %7.1f us %6.1f TF-eq | NW=2 %7.1f us %6.1f TF-eq\n", tr * 1e3, fl / tr / 1e9, tr2 * 1e3, fl / tr2 / 1e9);
        printf("attn3 B=%d N=%d H=%d: NW=2 %7.1f us  %6.1f TF-eq   (with prep kernel %7.1f us) | NW=4 %7.1f us %6.1f TF-eq\n", Bt, N, H, t * 1e3,
               fl / t / 1e9, tp * 1e3, t4 * 1e3, fl / t4 / 1e9);
    }
    return 0;
}
