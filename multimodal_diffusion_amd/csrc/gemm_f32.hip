// fp32 Linear (+bias)(+act)(+residual) on the gfx950 f32-input matrix cores.
//
//   C[M,N] = act(A[M,K] · W[N,K]^T + bias) + residual
//
// Stands under torch.nn.Linear as the reference uses it on the denoising path
// (avdiff/models/mmdt.py:60,77-83; heads/noise_heads.py:206-223; infer/sample_clip.py:54-56).
//
// Design (MI355X):
//  * v_mfma_f32_32x32x2_f32: exact fp32 multiply-accumulate, 64 FLOP/clk/SIMD (the fp32 roof, 157 TF).
//    A 32x32 tile costs ONE A and ONE B VGPR per MFMA, so LDS bandwidth is never the limiter; the kernels are
//    built to keep the matrix pipe issuing back to back.
//  * both operands are row-major with K contiguous (activations [M,K], torch weights [N,K]).
//  * the MFMA sums over k in a permuted order: lane half h of MFMA step j inside an 8-wide k group reads
//    k = 8*g + 4*h + j for BOTH operands, so one ds_read_b128 feeds four MFMAs with no shuffles.
//  * block ids are remapped so each XCD (private L2) gets a contiguous range of tiles that share A panels.
//
// Two kernels:
//  gemm_f32_dma_kernel — the hot path (K % 32 == 0, plain row-major C/R).  K tiles go global -> LDS directly
//    (global_load_lds_dwordx4, no VGPR staging, no ds_write): one wave instruction lands 8 rows x 128 B = 1 KiB
//    lane-linearly, so the LDS image is unpadded [rows][32] floats and bank conflicts are removed by XOR-ing the
//    16-byte chunk index with (row>>1)&7 on the per-lane SOURCE address and again on the fragment read.
//    Two LDS stages; tile k+1 is in flight under tile k's MFMAs.  The epilogue parks each wave's tile in LDS and
//    streams it out as whole 16-byte row segments with bias / exact-erf GELU / residual applied on float4s.
//  gemm_f32_reg_kernel — generic fallback (ragged K, segmented C/R row maps, SiLU, N = 32 outputs):
//    register-staged double buffering into [rows][32+4]-padded LDS, runtime-checked scalar epilogue.
#include "avd_common.h"

#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace avd {

constexpr int GEMM_BK = 32;
constexpr int GEMM_LD = GEMM_BK + 4;

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RES = 2 };

struct GemmArgs {
    const float* A;
    RowMap am;
    const float* W;
    const float* bias;
    const float* R;
    RowMap rm;
    float* C;
    RowMap cm;
    int64_t M;
    int N, K;
    int act;
    int nbn;
    // RMSNorm folding (dma kernels only): ss_in -> every output row is multiplied by 1 / (sqrt(sum_j ss_in[row][j]) / sqrt_d + eps)
    // before the bias (A is then the UN-normalised input and W carries the norm's scale); ss_out -> the epilogue also writes
    // the sum of squares of its final values per (row, 32-column chunk), the partials the next folded GEMM reads
    const float* ss_in;
    int ss_in_cols;
    float ss_sqrt_d, ss_eps;
    float* ss_out;
    TubeGather tg;          // gemm_f32_reg_kernel<..., GATHER = true> only: A rows are tube tokens of the latent at A
    int ldw;                // row stride of W in floats (= K, or the whole K of a split-K launch whose K field is one slice's length)
#ifdef AVD_GEMM_STAMPS      // diagnostic build only (tools/micro/gemm_stamps.py): per-block phase stamps, never in the product library
    unsigned long long* dbg;
#endif
};

#ifdef AVD_GEMM_STAMPS
#define AVD_STAMP(i)                                                                                           \
    do {                                                                                                        \
        if (threadIdx.x == 0) {                                                                                 \
            g.dbg[(size_t)blockIdx.x * 16 + 2 * (i)] = __builtin_amdgcn_s_memtime();                            \
            g.dbg[(size_t)blockIdx.x * 16 + 2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();                    \
        }                                                                                                       \
    } while (0)
#else
#define AVD_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

#define AVD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AVD_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// =====================================================================================================
// hot path: LDS-DMA staged main loop + LDS-staged float4 epilogue
// =====================================================================================================
// vmcnt(n) lgkmcnt(0) / vmcnt(n) alone as s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
constexpr int waitcnt_vm_lgkm0(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0070; }
constexpr int waitcnt_vm(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0f70; }

template <int BM, int BN, int WM, int WN, int EPI, int WPS, int NST>
__global__ __launch_bounds__(256, WPS) void gemm_f32_dma_kernel(GemmArgs g) {
    constexpr int BK = GEMM_BK;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_PIECES = BM / 32, B_PIECES = BN / 32;   // 1-KiB pieces (8 rows x 128 B) per wave
    constexpr int STAGE = (BM + BN) * BK;                   // floats per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];

    AVD_STAMP(0);
#ifdef AVD_GEMM_STAMPS
    if (threadIdx.x == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g.dbg[(size_t)blockIdx.x * 16 + 8] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // one tile per block, block ids remapped so an XCD gets a contiguous tile range (a persistent tile queue and a
    // first-generation stagger were measured in round 2, within +-1 % — DESIGN 4.2 — and removed from the product in round 3)
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int bm = wg / g.nbn, bn = wg % g.nbn;

    // split-K launches (gemm_f32_splitk): blockIdx.y = K slice z; g.K is ONE slice's length, rows keep their full strides (am.ld, ldw),
    // slice z writes its partial sums to C + z * M * ldc
    const int64_t koff = (int64_t)blockIdx.y * g.K;
    if (gridDim.y > 1) g.C += (int64_t)blockIdx.y * g.M * g.cm.ld;

    // ---- DMA source addresses: piece p = tile rows 8p..8p+7; lane -> row 8p + lane/8, PHYSICAL chunk lane%8 ----
    const int r8 = lane >> 3, pc = lane & 7;
    const float* a_src[A_PIECES];
    const float* b_src[B_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int trow = (wave + 4 * i) * 8 + r8;
        int64_t row = (int64_t)bm * BM + trow;
        row = row < g.M ? row : g.M - 1;
        a_src[i] = g.A + g.am.off(row) + koff + ((pc ^ ((trow >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
        const int trow = (wave + 4 * i) * 8 + r8;
        int n = bn * BN + trow;
        n = n < g.N ? n : g.N - 1;
        b_src[i] = g.W + (int64_t)n * g.ldw + koff + ((pc ^ ((trow >> 1) & 7)) << 2);
    }
    auto stage = [&](int kt, int buf) {
        float* as = smem + buf * STAGE;
        float* bs = as + BM * BK;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(a_src[i] + k0), AVD_LDS_PTR(as + (wave + 4 * i) * 8 * BK), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(b_src[i] + k0), AVD_LDS_PTR(bs + (wave + 4 * i) * 8 * BK), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = g.K / BK;
    // NST-stage ring: tiles 0 .. NST-2 are in flight before the loop, tile kt + NST - 1 is issued during tile kt
    constexpr int NPIECE_ = A_PIECES + B_PIECES;
    stage(0, 0);
    if (NST > 2 && nk > 1) {
        stage(1, 1);
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(NST > 2 ? NPIECE_ : 0));   // tile 0 landed, tile 1 may still fly
    } else {
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    }
    __builtin_amdgcn_s_barrier();

    // fragment read offsets (floats): row*32 + ((2kk+hi) ^ ((row>>1)&7))*4 — conflict-free ds_read_b128
    int a_row[TM], a_sw[TM], b_row[TN], b_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        a_row[i] = r * BK;
        a_sw[i] = (r >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * WN + j * 32 + l31;
        b_row[j] = r * BK;
        b_sw[j] = (r >> 1) & 7;
    }

    // Main loop schedule (per wave, per K tile of 32 = four k-groups of 8):
    //  * fragment reads run one k-group ahead of the MFMAs that consume them, across the K-tile boundary too, so a wave never
    //    waits on an LDS read it has just issued: one wave alone keeps its SIMD's matrix pipe issuing back to back and the
    //    co-resident block only has to cover the barrier.  hipcc waits lgkmcnt(0) (never a counted wait) while an LDS-DMA is
    //    in flight, so inside a k-group the order is: first MFMA step on the current registers (its wait finds the reads
    //    issued a whole group ago), THEN the reads for the next group, then the remaining steps.
    //  * the next stage's DMA pieces are issued one at a time behind single MFMAs (an LDS-DMA costs the wave ~60 issue cycles,
    //    one 64-cycle MFMA hides it) instead of as a block at the top of the tile.
    //  * sched_barrier(0) pins all of this: left alone, hipcc sinks every read down to its consumer and waits on it there.
    // Two named fragment register sets, static indexing throughout; the last K tile is a second instantiation without DMA.
    constexpr int MPS = TM * TN;                    // MFMAs per k step
    constexpr int NPIECE = A_PIECES + B_PIECES;     // DMA pieces per wave per stage
    static_assert(NPIECE <= 11 * MPS, "DMA pieces must all be issued before the tile's last k-group");
#define AVD_SB() __builtin_amdgcn_sched_barrier(0)
    auto ld_frag = [&](const float* as, const float* bs, int kk, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
#ifdef AVD_LAB_NOLDS       // diagnostic build: fragments stay whatever the registers hold
        if (kk >= 0) { asm volatile("" : "+v"(af[0]), "+v"(bf[0])); return; }
#endif
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + a_row[i] + (((2 * kk + hi) ^ a_sw[i]) << 2));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + b_row[j] + (((2 * kk + hi) ^ b_sw[j]) << 2));
    };
    f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    int cur = 0, dbuf = NST - 1;                  // ring slots of tile kt and of tile kt + NST - 1
    // mode 2: issue tile kt+NST-1's DMA, then wait for tile kt+1 leaving the newer tiles in flight; 1: nothing left to issue,
    // wait for everything; 0: last tile
    auto ktile = [&](int kt, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool DMA = MODE == 2;
        const float* as = smem + cur * STAGE;
        const float* bs = as + BM * BK;
        float* nas = smem + dbuf * STAGE;
        float* nbs = nas + BM * BK;
#ifdef AVD_LAB_SAMEK       // diagnostic build: every stage re-reads K tile 0 (all DMA traffic becomes L2 hits; results wrong by design)
        const int k1 = 0 * kt;
#else
        const int k1 = (kt + NST - 1) * BK;
#endif
        // steps [s0, s1) of k-group g on the registers (af, bf); MFMA number m of the tile is followed by DMA piece m - MPS
        auto mma = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN], int g, int s0, int s1) {
#pragma unroll
            for (int s = s0; s < s1; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
                        const int pc = (g * 4 + s) * MPS + i * TN + j - MPS;
#ifdef AVD_LAB_NODMA       // diagnostic build: no global traffic inside the main loop (results are wrong by design)
                        if (false) {
#else
                        if (DMA && pc >= 0 && pc < NPIECE) {
#endif
                            AVD_SB();
                            if (pc < A_PIECES)
                                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(a_src[pc < A_PIECES ? pc : 0] + k1),
                                                                 AVD_LDS_PTR(nas + (wave + 4 * pc) * 8 * BK), 16, 0, 0);
                            else
                                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(b_src[pc >= A_PIECES ? pc - A_PIECES : 0] + k1),
                                                                 AVD_LDS_PTR(nbs + (wave + 4 * (pc - A_PIECES)) * 8 * BK), 16, 0, 0);
                            AVD_SB();
                        }
                    }
        };
        AVD_SB(); mma(af0, bf0, 0, 0, 1); AVD_SB(); ld_frag(as, bs, 1, af1, bf1); AVD_SB(); mma(af0, bf0, 0, 1, 4);
        AVD_SB(); mma(af1, bf1, 1, 0, 1); AVD_SB(); ld_frag(as, bs, 2, af0, bf0); AVD_SB(); mma(af1, bf1, 1, 1, 4);
        AVD_SB(); mma(af0, bf0, 2, 0, 1); AVD_SB(); ld_frag(as, bs, 3, af1, bf1); AVD_SB(); mma(af0, bf0, 2, 1, 4);
        AVD_SB(); mma(af1, bf1, 3, 0, 2); AVD_SB();
        cur = cur + 1 == NST ? 0 : cur + 1;
        dbuf = dbuf + 1 == NST ? 0 : dbuf + 1;
        if constexpr (MODE != 0) {
            // every wave holds its last fragments of this stage in registers: wait for the next stage's DMA, swap
            __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(MODE == 2 ? (NST - 2) * NPIECE : 0));
            __builtin_amdgcn_s_barrier();
            const float* an = smem + cur * STAGE;
            ld_frag(an, an + BM * BK, 0, af0, bf0);
        }
        AVD_SB(); mma(af1, bf1, 3, 2, 4); AVD_SB();
    };
    AVD_STAMP(1);
    ld_frag(smem, smem + BM * BK, 0, af0, bf0);
    {
        int kt = 0;
        for (; kt + NST - 1 < nk; ++kt) ktile(kt, std::integral_constant<int, 2>{});
        for (; kt + 1 < nk; ++kt) ktile(kt, std::integral_constant<int, 1>{});
        ktile(kt, std::integral_constant<int, 0>{});
    }
#undef AVD_SB
    AVD_STAMP(2);
    __syncthreads();    // the slabs below overlay the stages: every wave must be past its last fragment read

    // ---- epilogue: park the wave's WM x WN tile in its own LDS slab, stream it out as 16-byte row segments ----
    constexpr int CLD = WN + 4;
    float* slab = smem + wave * WM * CLD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[i][j][r];
    __syncthreads();

    constexpr int LPR = WN / 4;     // lanes per output row
    constexpr int RPI = 64 / LPR;   // rows per wave instruction
    constexpr int NIT = WM / RPI;
    const int cr = lane / LPR, cc = (lane % LPR) * 4;
    const int n = bn * BN + wn * WN + cc;
    // folded RMSNorm: lane l of the wave computes the factor of the wave's row l; the store loop fetches it by shuffle
    float inv_r = 1.0f;
    if (g.ss_in) {
        const int64_t mr = (int64_t)bm * BM + wm * WM + lane;
        if (lane < WM && mr < g.M) {
            const float* sp = g.ss_in + mr * g.ss_in_cols;
            float ss = 0.f;
            if (g.ss_in_cols == 16) {            // d = 512: four independent 16-byte loads, fixed summation order
                const f32x4 p0 = *reinterpret_cast<const f32x4*>(sp), p1 = *reinterpret_cast<const f32x4*>(sp + 4);
                const f32x4 p2 = *reinterpret_cast<const f32x4*>(sp + 8), p3 = *reinterpret_cast<const f32x4*>(sp + 12);
                const f32x4 t = (p0 + p1) + (p2 + p3);
                ss = (t[0] + t[1]) + (t[2] + t[3]);
            } else {
                for (int j = 0; j < g.ss_in_cols; ++j) ss += sp[j];
            }
            inv_r = 1.0f / (sqrtf(ss) / g.ss_sqrt_d + g.ss_eps);
        }
    }
    if (n < g.N) {                  // N % 4 == 0: a float4 is wholly in or out (always true when folding: N % BN == 0 there)
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) bv = *reinterpret_cast<const f32x4*>(g.bias + n);
    const int64_t mbase = (int64_t)bm * BM + wm * WM + cr;
    const int ss_cols = g.N >> 5;
    float* cptr = g.C + mbase * g.cm.ld + n;
    const float* rptr = EPI == EPI_RES ? g.R + mbase * g.rm.ld + n : nullptr;
    constexpr int CHUNK = NIT < 8 ? NIT : 8;
#pragma unroll
    for (int c0 = 0; c0 < NIT; c0 += CHUNK) {
        f32x4 rv[CHUNK];
        if (EPI == EPI_RES) {
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
                const int64_t m = mbase + (int64_t)(c0 + u) * RPI;
                rv[u] = m < g.M ? *reinterpret_cast<const f32x4*>(rptr + (int64_t)(c0 + u) * RPI * g.rm.ld) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int it = c0 + u;
            f32x4 v = *reinterpret_cast<const f32x4*>(slab + (cr + it * RPI) * CLD + cc);
            if (g.ss_in) v *= __shfl(inv_r, cr + it * RPI, 64);
            v += bv;
            if (EPI == EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (EPI == EPI_RES) v += rv[u];
            const bool row_ok = mbase + (int64_t)it * RPI < g.M;
            if (row_ok) *reinterpret_cast<f32x4*>(cptr + (int64_t)it * RPI * g.cm.ld) = v;
            if (g.ss_out) {     // 8 consecutive lanes hold one row's 32 columns
                float sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                sq += __shfl_xor(sq, 1, 64);
                sq += __shfl_xor(sq, 2, 64);
                sq += __shfl_xor(sq, 4, 64);
                if (row_ok && (lane & 7) == 0) g.ss_out[(mbase + (int64_t)it * RPI) * ss_cols + (n >> 5)] = sq;
            }
        }
    }
    }   // n < N
    AVD_STAMP(3);
}

// =====================================================================================================
// generic fallback: register-staged double buffering, runtime-checked scalar epilogue
// =====================================================================================================
template <int BM, int BN, int WM, int WN, bool KTAIL, bool GATHER = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_reg_kernel(GemmArgs g) {
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][GEMM_LD]
    float* Bs = smem + 2 * BM * GEMM_LD;    // [2][BN][GEMM_LD]

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int bm = wg / g.nbn, bn = wg % g.nbn;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // global staging addresses: thread -> (row tid/8 + 32 i, 16-byte chunk tid%8)
    const int lrow = tid >> 3, lkc = (tid & 7) * 4;
    const float* a_src[A_IT];
    const float* b_src[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int64_t row = (int64_t)bm * BM + lrow + 32 * i;
        row = row < g.M ? row : g.M - 1;
        a_src[i] = GATHER ? g.A + g.tg.row_off(row) : g.A + g.am.off(row) + lkc;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        int n = bn * BN + lrow + 32 * i;
        n = n < g.N ? n : g.N - 1;
        b_src[i] = g.W + (int64_t)n * g.ldw + lkc;
    }
    const int st_off = lrow * GEMM_LD + lkc;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (g.K + GEMM_BK - 1) / GEMM_BK;
    f32x4 ra[A_IT], rb[B_IT];

    // KTAIL=false (K % 32 == 0): unconditional loads — a branch here makes hipcc merge the paths behind a
    // vmcnt(0) and the prefetch stops overlapping the MFMAs.  KTAIL=true: per-chunk guard (K % 4 == 0).
    auto load_tile = [&](int kt) {
        const int k0 = kt * GEMM_BK;
        // GATHER: this thread's float4 of tile kt is element k0 + lkc of the token -> (c, dt, dy, dx) of the latent
        const int64_t ka = GATHER ? g.tg.k_off((k0 + lkc) < g.K ? k0 + lkc : 0) : (int64_t)k0;
        if constexpr (!KTAIL) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_src[i] + ka);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_src[i] + k0);
        } else {
            const bool in = (k0 + lkc) < g.K;
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                ra[i] = in ? *reinterpret_cast<const f32x4*>(a_src[i] + ka) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                rb[i] = in ? *reinterpret_cast<const f32x4*>(b_src[i] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&](int buf) {
        float* as = As + buf * BM * GEMM_LD + st_off;
        float* bs = Bs + buf * BN * GEMM_LD + st_off;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<f32x4*>(as + i * 32 * GEMM_LD) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<f32x4*>(bs + i * 32 * GEMM_LD) = rb[i];
    };

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int a_rd = (wm * WM + l31) * GEMM_LD + 4 * hi;
    const int b_rd = (wn * WN + l31) * GEMM_LD + 4 * hi;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1) < nk;
        if (more) load_tile(kt + 1);
        const float* as = As + cur * BM * GEMM_LD + a_rd;
        const float* bs = Bs + cur * BN * GEMM_LD + b_rd;
#pragma unroll
        for (int kk = 0; kk < GEMM_BK / 8; ++kk) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(as + i * 32 * GEMM_LD + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bs + j * 32 * GEMM_LD + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool has_res = g.R != nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t m0 = (int64_t)bm * BM + wm * WM + i * 32;
        if (m0 >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = bn * BN + wn * WN + j * 32 + l31;
            const bool n_ok = n < g.N;
            const float bv = (g.bias != nullptr && n_ok) ? g.bias[n] : 0.f;
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + mfma32_row(r, hi);
                if (m < g.M && n_ok) {
                    float v = acc[i][j][r] + bv;
                    if (g.act == AVD_ACT_GELU) v = gelu_erf(v);
                    else if (g.act == AVD_ACT_SILU) v = silu(v);
                    if (has_res) v += g.R[g.rm.off(m) + n];
                    g.C[g.cm.off(m) + n] = v;
                }
            }
        }
    }
}

// =====================================================================================================
// host side
// =====================================================================================================

template <int BM, int BN, int WM, int WN, int EPI, int WPS, int NST>
static int launch_dma(const GemmArgs& a, hipStream_t st, int nz = 1) {
    constexpr int stage_lds = NST * (BM + BN) * GEMM_BK * 4, epi_lds = 4 * WM * (WN + 4) * 4;
    constexpr int lds = stage_lds > epi_lds ? stage_lds : epi_lds;
    static LdsAttr attr;
    auto kern = gemm_f32_dma_kernel<BM, BN, WM, WN, EPI, WPS, NST>;
    if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds, "gemm_f32_dma")) return rc;
    GemmArgs g = a;
    g.nbn = (a.N + BN - 1) / BN;
    const int64_t nwg = ((a.M + BM - 1) / BM) * g.nbn;
    AVD_REQUIRE(nwg < (1ll << 31), AVD_EUNSUPPORTED, "gemm grid too large");
    // tag = the kernel name exactly as rocprofv3 prints its template arguments (EPI: 0 bias, 1 gelu, 2 residual)
    static const int tag = prof_tag_id("gemm_f32_dma_kernel<%d, %d, %d, %d, %d, %d, %d>", BM, BN, WM, WN, EPI, WPS, NST);
    ProfScope prof(tag, 2.0 * (double)a.M * a.N * a.K * nz, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg, (unsigned)nz), dim3(256), lds, st, g);
    AVD_CHECK_LAUNCH("gemm_f32_dma");
    return AVD_OK;
}

template <int BM, int BN, int WM, int WN, int WPS, int NST>
static int launch_dma_epi(const GemmArgs& a, hipStream_t st) {
    if (a.R != nullptr) return launch_dma<BM, BN, WM, WN, EPI_RES, WPS, NST>(a, st);
    if (a.act == AVD_ACT_GELU) return launch_dma<BM, BN, WM, WN, EPI_GELU, WPS, NST>(a, st);
    return launch_dma<BM, BN, WM, WN, EPI_BIAS, WPS, NST>(a, st);
}

template <int BM, int BN, int WM, int WN, bool KTAIL, bool GATHER = false>
static int launch_reg(const GemmArgs& a, hipStream_t st) {
    constexpr int lds = 2 * (BM + BN) * GEMM_LD * 4;
    static LdsAttr attr;
    auto kern = gemm_f32_reg_kernel<BM, BN, WM, WN, KTAIL, GATHER>;
    if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), lds, "gemm_f32_reg")) return rc;
    GemmArgs g = a;
    g.nbn = (a.N + BN - 1) / BN;
    const int64_t nwg = ((a.M + BM - 1) / BM) * g.nbn;
    AVD_REQUIRE(nwg < (1ll << 31), AVD_EUNSUPPORTED, "gemm grid too large");
    static const int tag = GATHER ? prof_tag_id("gemm_f32_reg_kernel<%d, %d, %d, %d, %s, true>", BM, BN, WM, WN, KTAIL ? "true" : "false")
                                  : prof_tag_id("gemm_f32_reg_kernel<%d, %d, %d, %d, %s, false>", BM, BN, WM, WN, KTAIL ? "true" : "false");
    ProfScope prof(tag, 2.0 * (double)a.M * a.N * a.K, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, st, g);
    AVD_CHECK_LAUNCH("gemm_f32_reg");
    return AVD_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_reg_k(const GemmArgs& a, hipStream_t st) {
    return (a.K % GEMM_BK) ? launch_reg<BM, BN, WM, WN, true>(a, st) : launch_reg<BM, BN, WM, WN, false>(a, st);
}

#ifdef AVD_GEMM_STAMPS
unsigned long long* g_gemm_dbg = nullptr;
extern "C" void lab_set_dbg(unsigned long long* p) { g_gemm_dbg = p; }
extern "C" void lab_set_tile(int t);
#endif
int g_gemm_stages = getenv("AVD_GEMM_STAGES") ? atoi(getenv("AVD_GEMM_STAGES")) : 0;      // LDS ring depth of the 128x64 / 64x64 tiles: 0 = by size
int g_gemm_force_tile = getenv("AVD_GEMM_TILE") ? atoi(getenv("AVD_GEMM_TILE")) : -1;

#ifdef AVD_GEMM_STAMPS
extern "C" void lab_set_tile(int t) { g_gemm_force_tile = t; }
extern "C" void lab_set_stages(int n) { g_gemm_stages = n; }
#endif
bool gemm_f32_fold_supported(int N, int K) { return K % GEMM_BK == 0 && N % 128 == 0; }

int gemm_f32(const float* A, RowMap am, const float* W, const float* bias, const float* R, RowMap rm,
             float* C, RowMap cm, int64_t M, int N, int K, int act, hipStream_t st) {
    return gemm_f32_fold(A, am, W, bias, R, rm, C, cm, M, N, K, act, nullptr, 0, 1.f, 0.f, nullptr, st);
}

// gemm_f32 with RMSNorm folding (see GemmArgs); ss_in / ss_out may each be null.  ss_out must hold M * (N / 32) floats.
int gemm_f32_fold(const float* A, RowMap am, const float* W, const float* bias, const float* R, RowMap rm,
                  float* C, RowMap cm, int64_t M, int N, int K, int act, const float* ss_in, int ss_in_cols, float sqrt_d, float eps,
                  float* ss_out, hipStream_t st) {
    AVD_REQUIRE(A && W && C, AVD_EINVAL, "gemm: null pointer");
    AVD_REQUIRE(M >= 0 && N > 0 && K > 0, AVD_EINVAL, "gemm: bad dims M=%lld N=%d K=%d", (long long)M, N, K);
    AVD_REQUIRE(K % 4 == 0, AVD_EUNSUPPORTED, "gemm: K=%d must be a multiple of 4", K);
    AVD_REQUIRE(am.ld % 4 == 0 && (am.seg <= 0 || am.stride % 4 == 0), AVD_EUNSUPPORTED,
                "gemm: A row stride must be a multiple of 4 floats");
    AVD_REQUIRE(aligned16(A) && aligned16(W), AVD_EUNSUPPORTED, "gemm: A/W must be 16-byte aligned");
    AVD_REQUIRE(act == AVD_ACT_NONE || act == AVD_ACT_GELU || act == AVD_ACT_SILU, AVD_EINVAL, "gemm: bad act %d", act);
    if (M == 0) return AVD_OK;
    GemmArgs g{A, am, W, bias, R, rm, C, cm, M, N, K, act, 0, ss_in, ss_in_cols, sqrt_d, eps, ss_out, TubeGather{}, K};
#ifdef AVD_GEMM_STAMPS
    g.dbg = g_gemm_dbg;
#endif

    const int force = g_gemm_force_tile;      // tuning / test hook (avd_tune_set "gemm_tile"; env AVD_GEMM_TILE)
    const bool dma_ok = K % GEMM_BK == 0 && N % 4 == 0 && N > 32 && cm.seg <= 0 && cm.ld % 4 == 0 && aligned16(C) &&
                        (R == nullptr || (rm.seg <= 0 && rm.ld % 4 == 0 && aligned16(R))) &&
                        (bias == nullptr || aligned16(bias)) && (act == AVD_ACT_NONE || act == AVD_ACT_GELU) &&
                        !(R != nullptr && act != AVD_ACT_NONE) && force < 10;
    AVD_REQUIRE(!(ss_in || ss_out) || (dma_ok && gemm_f32_fold_supported(N, K) && (!ss_in || ss_in_cols > 0)), AVD_EUNSUPPORTED,
                "gemm: RMSNorm folding needs the LDS-DMA path (K %% 32 == 0, N %% 128 == 0)");
    const int64_t mb128 = (M + 127) / 128;
    if (dma_ok) {
        // 128x128 when it still gives >= 3 full rounds of 512 resident blocks; 128x64 (3 blocks/CU) for the
        // N = 512 projections so the last round is not half empty; 64x64 for small problems
        int tile = (N >= 128 && mb128 * ((N + 127) / 128) >= 1536) ? 0 : (mb128 * ((N + 63) / 64) >= 512) ? 1 : 2;
        if (force >= 0 && force <= 2) tile = force;
        // LDS ring depth (g_gemm_stages: 0 = by size, 2 / 3 = forced).  Measured: three stages pay for the 128x64 tile once the grid is
        // several rounds deep (C3: +2 % on the step) and for the 64x64 tile only on latency-bound grids of a few blocks per CU
        // (32x32 / B4: +2.5 %); in between two stages with one more resident block win (64x64 / B32: +3.6 %).
        const bool ring3_t1 = g_gemm_stages ? g_gemm_stages == 3 : M >= 16384;
        const bool ring3_t2 = g_gemm_stages ? g_gemm_stages == 3 : M <= 2048;
        if (tile == 0) return launch_dma_epi<128, 128, 64, 64, 2, 2>(g, st);
        if (tile == 1) return ring3_t1 ? launch_dma_epi<128, 64, 64, 32, 2, 3>(g, st) : launch_dma_epi<128, 64, 64, 32, 2, 2>(g, st);
        return ring3_t2 ? launch_dma_epi<64, 64, 32, 32, 3, 3>(g, st) : launch_dma_epi<64, 64, 32, 32, 4, 2>(g, st);
    }
    if (N <= 32) return launch_reg_k<128, 32, 32, 32>(g, st);
    if (N >= 128 && mb128 * ((N + 127) / 128) >= 512) return launch_reg_k<128, 128, 64, 64>(g, st);
    return launch_reg_k<64, 64, 32, 32>(g, st);
}

// ---- split-K for residual GEMMs of tiny batches (round 4) ----
// BASELINE C1 (32x32, batch 4: 344 rows) gives fc2 48 blocks of 64 x 64 that each walk K = 2,048 alone (45 us of a 0.84 ms step, eight
// times per step).  K is cut into slices (blockIdx.y) whose fp32 partial sums this kernel adds in slice order, then bias and residual as
// the fused epilogue would, writing the stream and (ss != null) its rows' sums of squares per 32-column chunk — the table a norm-folded
// GEMM reads.  Deterministic: fixed order, no atomics.  One thread per 4 columns.
__global__ __launch_bounds__(256) void splitk_reduce_f32_kernel(const float* __restrict__ part, int ns, const float* __restrict__ bias,
                                                                const float* R, int64_t ldr, float* C, int64_t ldc, float* __restrict__ ss,
                                                                int64_t M, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = N >> 2;
    const int64_t m = i / per_row;
    const int n = (int)(i % per_row) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
        v = *reinterpret_cast<const f32x4*>(part + m * N + n);
        for (int z = 1; z < ns; ++z) v += *reinterpret_cast<const f32x4*>(part + ((int64_t)z * M + m) * N + n);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (R) v += *reinterpret_cast<const f32x4*>(R + m * ldr + n);
        *reinterpret_cast<f32x4*>(C + m * ldc + n) = v;
    }
    if (ss) {       // 8 consecutive threads hold one 32-column chunk of one row (N % 32 == 0)
        float sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        sq += __shfl_xor(sq, 1, 64);
        sq += __shfl_xor(sq, 2, 64);
        sq += __shfl_xor(sq, 4, 64);
        if ((threadIdx.x & 7) == 0 && m < M) ss[m * (N >> 5) + (n >> 5)] = sq;
    }
}

int g_gemm_splitk = getenv("AVD_GEMM_SPLITK") ? atoi(getenv("AVD_GEMM_SPLITK")) : 4;     // largest slice count tried (0 off); avd_tune_set "gemm_splitk"
static int gemm_cu_count() {
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
// slices for C[M][N] = A W^T over K with the 64 x 64 tiles: the largest power of two (<= "gemm_splitk") that keeps one block per CU at
// most and a slice of >= 16 K tiles; 0 = do not split
int gemm_f32_splitk_slices(int64_t M, int N, int K) {
    if (g_gemm_splitk < 2 || N % 32 || K % 64) return 0;
    const int64_t cu = gemm_cu_count(), blocks = (M + 63) / 64 * ((N + 63) / 64);
    int ns = 1;
    while (ns * 2 <= g_gemm_splitk && ns * 2 <= kGemmSplitKMax && blocks * ns * 2 <= cu && K % (GEMM_BK * ns * 2) == 0 && K / (ns * 2) >= 16 * GEMM_BK) ns *= 2;
    return ns >= 2 ? ns : 0;
}
int64_t gemm_f32_splitk_ws_max_floats(int64_t M, int N, int K) {
    if (N % 32 || K % 64 || K / 2 < 16 * GEMM_BK) return 0;
    return (M + 63) / 64 * ((N + 63) / 64) * 2 <= gemm_cu_count() ? (int64_t)kGemmSplitKMax * M * N : 0;
}

// C = A W^T + bias + R with K cut into ns slices (partial sums in `part`, ns * M * N floats); ss_out as gemm_f32_fold's
int gemm_f32_splitk(const float* A, RowMap am, const float* W, const float* bias, const float* R, RowMap rm, float* C, RowMap cm, int64_t M,
                    int N, int K, int ns, float* part, float* ss_out, hipStream_t st) {
    AVD_REQUIRE(A && W && C && part && ns >= 2, AVD_EINVAL, "gemm_f32_splitk: null pointer");
    AVD_REQUIRE(K % (GEMM_BK * ns) == 0 && N % 32 == 0 && am.seg <= 0 && am.ld % 4 == 0 && cm.seg <= 0 && cm.ld % 4 == 0 &&
                    (R == nullptr || (rm.seg <= 0 && rm.ld % 4 == 0)), AVD_EUNSUPPORTED, "gemm_f32_splitk: shape (M=%lld N=%d K=%d slices=%d)",
                (long long)M, N, K, ns);
    AVD_REQUIRE(aligned16(A) && aligned16(W) && aligned16(C) && aligned16(part) && aligned16(bias) && aligned16(R), AVD_EUNSUPPORTED,
                "gemm_f32_splitk: pointers must be 16-byte aligned");
    const RowMap pm{N, 0, 0};
    GemmArgs g{A, am, W, nullptr, nullptr, pm, part, pm, M, N, K / ns, AVD_ACT_NONE, 0, nullptr, 0, 1.f, 0.f, nullptr, TubeGather{}, K};
#ifdef AVD_GEMM_STAMPS
    g.dbg = g_gemm_dbg;
#endif
    if (int rc = launch_dma<64, 64, 32, 32, EPI_BIAS, 3, 3>(g, st, ns)) return rc;
    const int64_t threads = M * (N >> 2);
    static const int tag = prof_tag_id("splitk_reduce_f32_kernel");
    ProfScope prof(tag, (double)M * N * 4.0 * (ns + 2), st);
    hipLaunchKernelGGL(splitk_reduce_f32_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, part, ns, bias, R, rm.ld, C, cm.ld,
                       ss_out, M, N);
    AVD_CHECK_LAUNCH("splitk_reduce_f32");
    return AVD_OK;
}

int gemm_f32_tube(const float* z, const TubeGather& tg, const float* W, const float* bias, float* C, RowMap cm, int64_t M, int N, int K,
                  hipStream_t st) {
    AVD_REQUIRE(z && W && C && M > 0 && N > 0 && K > 0, AVD_EINVAL, "gemm_tube: bad arguments");
    AVD_REQUIRE(tg.w % 4 == 0 && tg.W % 4 == 0 && K % 4 == 0 && aligned16(z) && aligned16(W), AVD_EUNSUPPORTED,
                "gemm_tube: the gathered A load needs w %% 4 == 0, W %% 4 == 0, K %% 4 == 0 and 16-byte aligned operands");
    GemmArgs g{z, RowMap{K, 0, 0}, W, bias, nullptr, cm, C, cm, M, N, K, AVD_ACT_NONE, 0, nullptr, 0, 1.f, 0.f, nullptr, tg, K};
#ifdef AVD_GEMM_STAMPS
    g.dbg = g_gemm_dbg;
#endif
    return (K % GEMM_BK) ? launch_reg<64, 64, 32, 32, true, true>(g, st) : launch_reg<64, 64, 32, 32, false, true>(g, st);
}

}  // namespace avd

extern "C" int avd_gemm_rmsfold_f32(const float* A, const float* W, const float* bias, const float* residual, float* C, int64_t M,
                                    int N, int K, int act, const float* ss_in, int ss_in_cols, float eps, float* ss_out,
                                    avd_stream_t stream) {
    using namespace avd;
    const RowMap ra{K, 0, 0}, rc{N, 0, 0};
    return gemm_f32_fold(A, ra, W, bias, residual, rc, C, rc, M, N, K, act, ss_in, ss_in_cols, (float)sqrt((double)K), eps, ss_out,
                         static_cast<hipStream_t>(stream));
}

extern "C" int avd_gemm_bias_act_f32(const float* A, int64_t lda, const float* W, const float* bias,
                                     const float* residual, int64_t ldr, float* C, int64_t ldc,
                                     int64_t M, int N, int K, int act, avd_stream_t stream) {
    using namespace avd;
    AVD_REQUIRE(lda >= K && ldc >= N && (residual == nullptr || ldr >= N), AVD_EINVAL,
                "gemm: leading dimension smaller than row length");
    return gemm_f32(A, RowMap{lda, 0, 0}, W, bias, residual, RowMap{ldr, 0, 0}, C, RowMap{ldc, 0, 0}, M, N, K, act,
                    static_cast<hipStream_t>(stream));
}
