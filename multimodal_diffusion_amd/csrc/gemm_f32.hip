// fp32 Linear (+bias)(+act)(+residual) on the gfx950 f32-input matrix cores.
//
//   C[M,N] = act(A[M,K] · W[N,K]^T + bias) + residual
//
// Stands under torch.nn.Linear as the reference uses it on the denoising path
// (avdiff/models/mmdt.py:60,77-83; heads/noise_heads.py:206-223; infer/sample_clip.py:54-56).
//
// Design (MI355X):
//  * v_mfma_f32_32x32x2_f32: exact fp32 multiply-accumulate, 64 FLOP/clk/SIMD (the fp32 roof, 157 TF).
//    A 32x32 tile costs ONE A and ONE B VGPR per MFMA, so LDS bandwidth is never the limiter; the kernel
//    is built to keep the matrix pipe issuing back to back.
//  * both operands are row-major with K contiguous (activations [M,K], torch weights [N,K]); tiles are staged
//    global -> registers -> LDS as [rows][32+4] floats.  The +4 pad makes the ds_read_b128 fragment reads
//    (16 distinct rows per lane group, 16 B each) and the ds_write_b128 stores conflict-free.
//  * the MFMA sums over k in a permuted order: lane half h of MFMA step j inside an 8-wide k group reads
//    k = 8*g + 4*h + j for BOTH operands, so one ds_read_b128 feeds four MFMAs with no shuffles.
//  * register-staged double buffering: tile k+1's global loads are issued before tile k's MFMAs, written to
//    the other LDS buffer after them; one barrier per K tile.  Full K tiles load unconditionally; a ragged last
//    tile (K % 32) takes a wave-uniform branch to a guarded load, so the steady state has no exec masking.
//  * the epilogue is specialised at compile time (bias / bias+GELU / bias+residual, plain row-major, whole
//    32x32 tiles in range) with a generic runtime-checked fallback for ragged edges and segmented rows.
//  * block ids are remapped so each XCD (private L2) gets a contiguous range of tiles that share A panels.
#include "avd_common.h"

#include <stdlib.h>

namespace avd {

constexpr int GEMM_BK = 32;
constexpr int GEMM_LD = GEMM_BK + 4;

enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RES = 2, EPI_GENERIC = 3 };

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
};

template <int BM, int BN, int WM, int WN, int EPI, bool KTAIL>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_IT = BM / 32, B_IT = BN / 32;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][GEMM_LD]
    float* Bs = smem + 2 * BM * GEMM_LD;    // [2][BN][GEMM_LD]

    // ---- XCD-aware, bijective block remap (blocks b and b+8 share an XCD) ----
    const int nwg = gridDim.x;
    int wg;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int bm = wg / g.nbn, bn = wg % g.nbn;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // ---- global staging addresses: thread -> (row tid/8 + 32 i, 16-byte chunk tid%8) ----
    const int lrow = tid >> 3, lkc = (tid & 7) * 4;
    const float* a_src[A_IT];
    const float* b_src[B_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int64_t row = (int64_t)bm * BM + lrow + 32 * i;
        row = row < g.M ? row : g.M - 1;
        a_src[i] = g.A + (EPI == EPI_GENERIC ? g.am.off(row) : row * g.am.ld) + lkc;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        int n = bn * BN + lrow + 32 * i;
        n = n < g.N ? n : g.N - 1;
        b_src[i] = g.W + (int64_t)n * g.K + lkc;
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

    // KTAIL=false (K % 32 == 0): unconditional loads — any branch here makes hipcc merge the paths behind a
    // vmcnt(0) and the prefetch no longer overlaps the MFMAs.  KTAIL=true: per-chunk guard (K % 4 == 0).
    auto load_tile = [&](int kt) {
        const int k0 = kt * GEMM_BK;
        if constexpr (!KTAIL) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const f32x4*>(a_src[i] + k0);
#pragma unroll
            for (int i = 0; i < B_IT; ++i) rb[i] = *reinterpret_cast<const f32x4*>(b_src[i] + k0);
        } else {
            const bool in = (k0 + lkc) < g.K;
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                ra[i] = in ? *reinterpret_cast<const f32x4*>(a_src[i] + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
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

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t m0 = (int64_t)bm * BM + wm * WM + i * 32;
        if (m0 >= g.M) continue;
        const bool rows_full = m0 + 32 <= g.M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n0 = bn * BN + wn * WN + j * 32;
            if (n0 >= g.N) continue;
            const int n = n0 + l31;
            if (EPI != EPI_GENERIC && rows_full && n0 + 32 <= g.N) {
                // fast path: whole 32x32 tile in range, plain row-major C (and R)
                const float bv = g.bias ? g.bias[n] : 0.f;
                float* crow = g.C + (m0 + 4 * hi) * g.cm.ld + n;
                const float* rrow = EPI == EPI_RES ? g.R + (m0 + 4 * hi) * g.rm.ld + n : nullptr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    float v = acc[i][j][r] + bv;
                    if (EPI == EPI_GELU) v = gelu_erf(v);
                    if (EPI == EPI_RES) v += rrow[ro * g.rm.ld];
                    crow[ro * g.cm.ld] = v;
                }
            } else {
                const bool n_ok = n < g.N;
                const float bv = (g.bias != nullptr && n_ok) ? g.bias[n] : 0.f;
                const int act = EPI == EPI_GENERIC ? g.act : (EPI == EPI_GELU ? AVD_ACT_GELU : AVD_ACT_NONE);
                const bool has_res = EPI == EPI_GENERIC ? (g.R != nullptr) : (EPI == EPI_RES);
#pragma unroll 1
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + mfma32_row(r, hi);
                    if (m < g.M && n_ok) {
                        float v = acc[i][j][r] + bv;
                        if (act == AVD_ACT_GELU) v = gelu_erf(v);
                        else if (act == AVD_ACT_SILU) v = silu(v);
                        if (has_res) v += g.R[g.rm.off(m) + n];
                        g.C[g.cm.off(m) + n] = v;
                    }
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int EPI, bool KTAIL>
static int launch_gemm_epi(const GemmArgs& a, hipStream_t st) {
    constexpr int lds = 2 * (BM + BN) * GEMM_LD * (int)sizeof(float);
    static bool attr_set = false;
    auto kern = gemm_f32_kernel<BM, BN, WM, WN, EPI, KTAIL>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return set_error(AVD_ELAUNCH, "gemm attr: %s", hipGetErrorString(e));
        attr_set = true;
    }
    GemmArgs g = a;
    const int64_t nbm = (a.M + BM - 1) / BM;
    g.nbn = (a.N + BN - 1) / BN;
    const int64_t nwg = nbm * g.nbn;
    AVD_REQUIRE(nwg < (1ll << 31), AVD_EUNSUPPORTED, "gemm grid too large");
    constexpr int tag = (BM == 128 && BN == 128) ? AVD_PROF_GEMM_128x128 : (BM == 128 && BN == 64) ? AVD_PROF_GEMM_128x64
                        : (BM == 64) ? AVD_PROF_GEMM_64x64 : AVD_PROF_GEMM_128x32;
    ProfScope prof(tag, 2.0 * (double)a.M * a.N * a.K, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, st, g);
    AVD_CHECK_LAUNCH("gemm_f32");
    return AVD_OK;
}

template <int BM, int BN, int WM, int WN>
static int launch_gemm(const GemmArgs& a, hipStream_t st) {
    if (a.K % GEMM_BK) return launch_gemm_epi<BM, BN, WM, WN, EPI_GENERIC, true>(a, st);
    const bool plain = a.am.seg <= 0 && a.cm.seg <= 0 && (a.R == nullptr || a.rm.seg <= 0);
    if (plain && a.act == AVD_ACT_NONE && a.R == nullptr) return launch_gemm_epi<BM, BN, WM, WN, EPI_BIAS, false>(a, st);
    if (plain && a.act == AVD_ACT_GELU && a.R == nullptr) return launch_gemm_epi<BM, BN, WM, WN, EPI_GELU, false>(a, st);
    if (plain && a.act == AVD_ACT_NONE && a.R != nullptr) return launch_gemm_epi<BM, BN, WM, WN, EPI_RES, false>(a, st);
    return launch_gemm_epi<BM, BN, WM, WN, EPI_GENERIC, false>(a, st);
}

int gemm_f32(const float* A, RowMap am, const float* W, const float* bias, const float* R, RowMap rm,
             float* C, RowMap cm, int64_t M, int N, int K, int act, hipStream_t st) {
    AVD_REQUIRE(A && W && C, AVD_EINVAL, "gemm: null pointer");
    AVD_REQUIRE(M >= 0 && N > 0 && K > 0, AVD_EINVAL, "gemm: bad dims M=%lld N=%d K=%d", (long long)M, N, K);
    AVD_REQUIRE(K % 4 == 0, AVD_EUNSUPPORTED, "gemm: K=%d must be a multiple of 4", K);
    AVD_REQUIRE(am.ld % 4 == 0 && (am.seg <= 0 || am.stride % 4 == 0), AVD_EUNSUPPORTED,
                "gemm: A row stride must be a multiple of 4 floats");
    AVD_REQUIRE(aligned16(A) && aligned16(W), AVD_EUNSUPPORTED, "gemm: A/W must be 16-byte aligned");
    AVD_REQUIRE(act == AVD_ACT_NONE || act == AVD_ACT_GELU || act == AVD_ACT_SILU, AVD_EINVAL, "gemm: bad act %d", act);
    if (M == 0) return AVD_OK;
    GemmArgs g{A, am, W, bias, R, rm, C, cm, M, N, K, act, 0};
    // tile choice: big square tiles when there is enough work to fill 256 CUs x 2 blocks, finer ones otherwise
    static const int force = getenv("AVD_GEMM_TILE") ? atoi(getenv("AVD_GEMM_TILE")) : -1;   // tuning/debug only
    if (force == 0) return launch_gemm<128, 128, 64, 64>(g, st);
    if (force == 1) return launch_gemm<128, 64, 64, 32>(g, st);
    if (force == 2) return launch_gemm<64, 64, 32, 32>(g, st);
    if (force == 3) return launch_gemm<128, 32, 32, 32>(g, st);
    const int64_t big = ((M + 127) / 128) * ((N + 127) / 128);
    if (N >= 128 && big >= 512) return launch_gemm<128, 128, 64, 64>(g, st);
    if (N >= 64 && ((M + 127) / 128) * ((N + 63) / 64) >= 384) return launch_gemm<128, 64, 64, 32>(g, st);
    if (N > 32) return launch_gemm<64, 64, 32, 32>(g, st);
    return launch_gemm<128, 32, 32, 32>(g, st);
}

}  // namespace avd

extern "C" int avd_gemm_bias_act_f32(const float* A, int64_t lda, const float* W, const float* bias,
                                     const float* residual, int64_t ldr, float* C, int64_t ldc,
                                     int64_t M, int N, int K, int act, avd_stream_t stream) {
    using namespace avd;
    AVD_REQUIRE(lda >= K && ldc >= N && (residual == nullptr || ldr >= N), AVD_EINVAL,
                "gemm: leading dimension smaller than row length");
    return gemm_f32(A, RowMap{lda, 0, 0}, W, bias, residual, RowMap{ldr, 0, 0}, C, RowMap{ldc, 0, 0}, M, N, K, act,
                    static_cast<hipStream_t>(stream));
}
