// Lab: fp32-accurate GEMM on the bf16 matrix pipe.  Each fp32 operand is split exactly into three bf16 planes
// (x = h + m + l, 8 significant bits each); a product keeps the six terms down to 2^-16 (hh, hm, mh, hl, lh, mm) and drops
// ml, lm, ll (<= 2^-24 relative), so the result carries fp32-level error while v_mfma_f32_32x32x16_bf16 runs at 16x the
// fp32 MFMA rate: 6 bf16 MFMAs (32 cycles each) replace 8 fp32 MFMAs (64 cycles each) per k=16.
//
// Operand format "bf16x3-g16": element (r, k, plane p) at byte r*6K + (k/16)*96 + p*32 + (k%16)*2, i.e. per row, groups of
// 16 k-values hold [h:32 B | m:32 B | l:32 B].  One K-tile of 16 is then 96 contiguous bytes per row.
//
// Usage: split_lab [check]     (no argument: timing at the C3 shapes)
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

__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// ---------------------------------------------------------------------------------------------------------
// split: fp32 [rows][K] -> bf16x3-g16
// ---------------------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int64_t rows, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one thread = 8 consecutive k of one row
    const int per_row = K / 8;
    if (i >= rows * per_row) return;
    const int64_t r = i / per_row;
    const int k = (int)(i % per_row) * 8;
    float v[8];
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(x + r * K + k);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(x + r * K + k + 4);
    u32x4 H, Mi, Lo;
    split8(v, H, Mi, Lo);
    unsigned char* dst = out + r * 6 * (int64_t)K + (k >> 4) * 96 + ((k >> 3) & 1) * 16;
    *reinterpret_cast<u32x4*>(dst) = H;
    *reinterpret_cast<u32x4*>(dst + 32) = Mi;
    *reinterpret_cast<u32x4*>(dst + 64) = Lo;
}

// ---------------------------------------------------------------------------------------------------------
// GEMM: C[M][N] = A[M][K] * W[N][K]^T + bias, A and W in bf16x3-g16, C fp32
// ---------------------------------------------------------------------------------------------------------
struct SArgs {
    const unsigned char* A;
    const unsigned char* W;
    const float* bias;
    float* C;
    int64_t M;
    int N, K, nbn, sm, sn;
};

template <int N> __device__ __forceinline__ void wait_vm();
template <> __device__ __forceinline__ void wait_vm<0>() { __builtin_amdgcn_s_waitcnt(0x0f70); }
template <> __device__ __forceinline__ void wait_vm<4>() { __builtin_amdgcn_s_waitcnt(0x0f74); }
template <> __device__ __forceinline__ void wait_vm<5>() { __builtin_amdgcn_s_waitcnt(0x0f75); }
template <> __device__ __forceinline__ void wait_vm<6>() { __builtin_amdgcn_s_waitcnt(0x0f76); }
template <> __device__ __forceinline__ void wait_vm<8>() { __builtin_amdgcn_s_waitcnt(0x0f78); }
template <> __device__ __forceinline__ void wait_vm<10>() { __builtin_amdgcn_s_waitcnt(0x0f7a); }
template <> __device__ __forceinline__ void wait_vm<12>() { __builtin_amdgcn_s_waitcnt(0x0f7c); }
template <> __device__ __forceinline__ void wait_vm<15>() { __builtin_amdgcn_s_waitcnt(0x0f7f); }
template <> __device__ __forceinline__ void wait_vm<18>() { __builtin_amdgcn_s_waitcnt(0x4f72); }

template <int BM, int BN, int WM, int WN, int NST, int WPS, int TERMS, int MODE = 0>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, WPS) void sgemm_kernel(SArgs g) {
    constexpr int ROWB = 96;                              // bytes per tile row (3 planes x 16 bf16)
    constexpr int WAVES_N = BN / WN;
    constexpr int NW = (BM / WM) * WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int STAGE = (BM + BN) * ROWB;               // bytes
    constexpr int PIECES = STAGE / 1024;
    static_assert(STAGE % 1024 == 0, "stage is whole DMA pieces");
    constexpr int PPW = (PIECES + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    int bm, bn;
    if (g.sn > 0) {
        // super-tiles of sm x sn blocks (64 co-resident blocks of an XCD share sm A panels and sn W panels)
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
    const int64_t rowbytes = 6 * (int64_t)g.K;

    // DMA: piece q covers LDS bytes [1024 q, 1024 q + 1024); lane -> 16 bytes at 1024 q + 16 lane
    const unsigned char* src[PPW];
    int dst[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        int q = wave * PPW + i;
        q = q < PIECES ? q : PIECES - 1;
        const int L = q * 1024 + lane * 16;
        const int row = L / ROWB, slot = (L % ROWB) >> 4;
        const int chunk = slot ^ ((row >> 3) & 1);
        if (row < BM) {
            int64_t m = (int64_t)bm * BM + row;
            m = m < g.M ? m : g.M - 1;
            src[i] = g.A + m * rowbytes + chunk * 16;
        } else {
            int n = bn * BN + row - BM;
            n = n < g.N ? n : g.N - 1;
            src[i] = g.W + n * rowbytes + chunk * 16;
        }
        dst[i] = q * 1024;
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + (int64_t)kt * ROWB), LDS_PTR(smem + buf * STAGE + dst[i]), 16, 0, 0);
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

    const int nk = g.K / 16;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue(s, s);

    int cur = 0, nxt = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NST - 1 <= nk) wait_vm<(NST - 2) * PPW>(); else wait_vm<0>();
        asm volatile("s_barrier" ::: "memory");          // no fence: a fence would drain vmcnt and with it the tiles in flight
        if (MODE != 1 && kt + NST - 1 < nk) issue(kt + NST - 1, nxt);
        const unsigned char* st = smem + cur * STAGE;
        if (MODE == 2) { cur = cur + 1 == NST ? 0 : cur + 1; nxt = nxt + 1 == NST ? 0 : nxt + 1; continue; }
        bf16x8 af[TM][3], bf[TN][3];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(st + a_off[i] + 32 * p);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(st + b_off[j] + 32 * p);
        // small terms first
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 6 - TERMS; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[t]], bf[j][PB[t]], acc[i][j], 0, 0, 0);
        cur = cur + 1 == NST ? 0 : cur + 1;
        nxt = nxt + 1 == NST ? 0 : nxt + 1;
    }
    __syncthreads();

    // epilogue: park the wave tile in LDS, stream out 16-byte row segments (+bias)
    constexpr int CLD = WN + 4;
    float* slab = reinterpret_cast<float*>(smem) + wave * WM * CLD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    constexpr int LPR = WN / 4, RPI = 64 / LPR, NIT = WM / RPI;
    const int cr = lane / LPR, cc = (lane % LPR) * 4;
    const int n = bn * BN + wn * WN + cc;
    if (n >= g.N) return;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
    const int64_t mbase = (int64_t)bm * BM + wm * WM + cr;
    float* cptr = g.C + mbase * g.N + n;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(slab + (cr + it * RPI) * CLD + cc);
        v += bv;
        if (mbase + (int64_t)it * RPI < g.M) *reinterpret_cast<f32x4*>(cptr + (int64_t)it * RPI * g.N) = v;
    }
}

template <int BM, int BN, int WM, int WN, int NST, int WPS, int TERMS, int MODE = 0>
static float run_gemm(const SArgs& a0, int iters, int super = 0) {
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int stage_lds = NST * (BM + BN) * 96, epi_lds = NW * WM * (WN + 4) * 4;
    constexpr int lds = stage_lds > epi_lds ? stage_lds : epi_lds;
    auto kern = sgemm_kernel<BM, BN, WM, WN, NST, WPS, TERMS, MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    SArgs a = a0;
    a.nbn = (a.N + BN - 1) / BN;
    unsigned nwg = (unsigned)(((a.M + BM - 1) / BM) * a.nbn);
    a.sm = a.sn = 0;
    if (super > 0) {
        int sn = 8;
        while (a.nbn % sn) sn >>= 1;
        a.sn = sn;
        a.sm = super / sn;
        const int nbm = (int)((a.M + BM - 1) / BM);
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

struct Dev {
    float *A, *W, *bias, *C;
    unsigned char *As, *Ws;
};

static void do_split(const float* x, unsigned char* out, int64_t rows, int K) {
    const int64_t n = rows * (K / 8);
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, out, rows, K);
}

template <int TERMS>
static int check(int M, int N, int K, float scaleA) {
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hb(N), hC((size_t)M * N);
    fill(hA, 1, scaleA); fill(hW, 2, 1.f / sqrtf((float)K)); fill(hb, 3, 1.f);
    Dev d;
    CK(hipMalloc(&d.A, hA.size() * 4)); CK(hipMalloc(&d.W, hW.size() * 4)); CK(hipMalloc(&d.bias, N * 4));
    CK(hipMalloc(&d.C, hC.size() * 4)); CK(hipMalloc(&d.As, hA.size() * 6)); CK(hipMalloc(&d.Ws, hW.size() * 6));
    CK(hipMemcpy(d.A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d.bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    do_split(d.A, d.As, M, K); do_split(d.W, d.Ws, N, K);
    SArgs a{d.As, d.Ws, d.bias, d.C, M, N, K, 0, 0, 0};
    int bad = 0;
    for (int cfg = 0; cfg < 4; ++cfg) {
        CK(hipMemset(d.C, 0xff, hC.size() * 4));
        if (cfg == 0) run_gemm<128, 128, 64, 64, 3, 2, TERMS>(a, 1);
        else if (cfg == 1) run_gemm<128, 64, 64, 32, 4, 2, TERMS>(a, 1);
        else if (cfg == 2) run_gemm<128, 128, 64, 64, 3, 2, TERMS>(a, 1, 64);
        else run_gemm<256, 128, 64, 64, 4, 1, TERMS>(a, 1, 32);
        CK(hipMemcpy(hC.data(), d.C, hC.size() * 4, hipMemcpyDeviceToHost));
        double max_err = 0, max_ref = 0, sum_err2 = 0, max_f32 = 0;
        for (int m = 0; m < M; ++m)
            for (int n = 0; n < N; ++n) {
                double ref = hb[n];
                float f32 = 0.f;
                for (int k = 0; k < K; ++k) {
                    ref += (double)hA[(size_t)m * K + k] * (double)hW[(size_t)n * K + k];
                    f32 = fmaf(hA[(size_t)m * K + k], hW[(size_t)n * K + k], f32);
                }
                f32 += hb[n];
                const double e = fabs((double)hC[(size_t)m * N + n] - ref);
                max_err = fmax(max_err, e); max_ref = fmax(max_ref, fabs(ref)); sum_err2 += e * e;
                max_f32 = fmax(max_f32, fabs((double)f32 - ref));
            }
        printf("check TERMS=%d cfg=%d M=%d N=%d K=%d scaleA=%g: max|err| %.3e (fp32 FMA chain %.3e)  rms err %.3e  max|ref| %.3e  rel %.3e\n",
               TERMS, cfg, M, N, K, scaleA, max_err, max_f32, sqrt(sum_err2 / ((double)M * N)), max_ref, max_err / max_ref);
        if (!(max_err / max_ref < (TERMS == 6 ? 2e-6 : 1e-2))) bad = 1;
    }
    CK(hipFree(d.A)); CK(hipFree(d.W)); CK(hipFree(d.bias)); CK(hipFree(d.C)); CK(hipFree(d.As)); CK(hipFree(d.Ws));
    return bad;
}

int main(int argc, char** argv) {
    if (argc > 1) {
        int bad = 0;
        bad |= check<6>(200, 192, 512, 1.f);
        bad |= check<6>(333, 320, 2048, 100.f);
        bad |= check<3>(200, 192, 512, 1.f);
        bad |= check<1>(200, 192, 512, 1.f);
        printf(bad ? "CHECK FAILED\n" : "CHECK OK\n");
        return bad;
    }
    const int64_t M = 64 * 421;
    const int KMAX = 2048, NMAX = 2048;
    Dev d;
    CK(hipMalloc(&d.A, (size_t)M * KMAX * 4)); CK(hipMalloc(&d.W, (size_t)NMAX * KMAX * 4)); CK(hipMalloc(&d.bias, NMAX * 4));
    CK(hipMalloc(&d.C, (size_t)M * NMAX * 4)); CK(hipMalloc(&d.As, (size_t)M * KMAX * 6)); CK(hipMalloc(&d.Ws, (size_t)NMAX * KMAX * 6));
    {
        std::vector<float> h((size_t)M * KMAX);
        fill(h, 5, 1.f);
        CK(hipMemcpy(d.A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d.W, h.data(), (size_t)NMAX * KMAX * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d.bias, h.data(), NMAX * 4, hipMemcpyHostToDevice));
    }
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 1536, 512}, {"out_proj", 512, 512}, {"fc1", 2048, 512}, {"fc2", 512, 2048}};
    // warm the clocks
    {
        do_split(d.A, d.As, M, 512); do_split(d.W, d.Ws, 2048, 512);
        SArgs a{d.As, d.Ws, d.bias, d.C, M, 2048, 512, 0, 0, 0};
        run_gemm<128, 128, 64, 64, 3, 2, 6>(a, 300);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (auto& s : shapes) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            do_split(d.A, d.As, M, s.K);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float sms; CK(hipEventElapsedTime(&sms, e0, e1));
            do_split(d.W, d.Ws, s.N, s.K);
            SArgs a{d.As, d.Ws, d.bias, d.C, M, s.N, s.K, 0, 0, 0};
            const double fl = 2.0 * (double)M * s.N * s.K;
            const int it = 100;
            float t;
            printf("%-9s N=%4d K=%4d  split(A) %.1f us |", s.name, s.N, s.K, sms * 1e3);
            t = run_gemm<128, 128, 64, 64, 3, 2, 6>(a, it); printf(" 128x128 s3: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<128, 128, 64, 64, 3, 2, 6>(a, it, 64); printf(" +super64: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<128, 128, 64, 64, 3, 2, 6>(a, it, 32); printf(" +super32: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<256, 128, 64, 64, 4, 1, 6>(a, it); printf(" 256x128 s4: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<256, 128, 64, 64, 4, 1, 6>(a, it, 32); printf(" +super32: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<256, 128, 64, 64, 3, 1, 6>(a, it, 32); printf(" s3+super32: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<128, 128, 64, 64, 3, 2, 1>(a, it, 64); printf(" [1-term 128x128 s3 super64: %6.1f us %6.1f TF]", t * 1e3, fl / t / 1e9);
            t = run_gemm<128, 128, 64, 64, 3, 2, 6, 1>(a, it, 64); printf("\n      128x128 noDMA: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<128, 128, 64, 64, 3, 2, 6, 2>(a, it, 64); printf(" DMAonly: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<256, 128, 64, 64, 4, 1, 6, 1>(a, it, 32); printf(" 256x128 noDMA: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            t = run_gemm<256, 128, 64, 64, 4, 1, 6, 2>(a, it, 32); printf(" DMAonly: %6.1f us %6.1f TF |", t * 1e3, fl / t / 1e9);
            printf("\n");
            fflush(stdout);
        }
    return 0;
}
