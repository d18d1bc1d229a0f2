// VideoVAE.decode on MI355X — the loop boundary right after the sampler:
//   avdiff/models/encoders/vae_video3d.py:195-214 (decode), :79-84 (_conv_block_3d = Conv3d 3x3x3 -> GELU -> GroupNorm(8)),
//   :108-119 (from_lat 1x1x1, dec_net, to_img 1x1x1), trilinear upsample align_corners=False, sigmoid/tanh output.
//
// Layout: activations are NDHWC (64 channels = 256 contiguous bytes per voxel) in a buffer padded by one zero voxel
// on every side of T, H, W.  In that layout a 3x3x3 tap is a CONSTANT address shift for every voxel of a tile, so
// the convolution is the fp32 MFMA GEMM of gemm_f32.hip with a per-K-tile scalar offset on the A operand:
//   M = voxels (128 per block), N = 64 output channels, K = 27 taps x 64 input channels = 54 K-tiles of 32,
//   A tiles by LDS-DMA straight from the padded activation (no im2col buffer, no boundary branches),
//   weights pre-arranged [out][tap][in].  Same XOR-swizzled unpadded LDS image and permuted-k fragment reads.
// Epilogue: bias + exact-erf GELU on float4 row segments, store NDHWC, and per-(sample, group) partial sums /
// sums of squares reduced with wavefront shuffles into a partials buffer (deterministic two-level reduction, no
// atomics).  GroupNorm is finished by a tiny fp64 reduction kernel and applied by an HBM-bound pass that writes the
// next conv's padded input — or, after the last block, fuses normalise + to_img (64->3) + sigmoid and writes NCDHW.
#include "avd_common.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace avd {

constexpr int VC = 64;          // channels of the decoder trunk (reference default dec_base = 64)
constexpr int VG = 8;           // GroupNorm groups = min(8, C)
constexpr int VBM = 128;        // voxels per block
constexpr int VBK = 32;

#define AVD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AVD_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

struct ConvArgs {
    const float* X;      // padded NDHWC [B, T+2, H+2, W+2, CIN]   (CIN = 64, or 4 = RGB + one zero channel)
    const float* Wt;     // [64][27][64]  |  [64][32 taps (27 real, 5 zero)][4]
    const float* bias;   // [64]
    float* Y;            // NDHWC [B, T, H, W, 64]
    float* part;         // [B*tiles][2][8][2]
    int T, H, W, tiles;  // tiles = ceil(T*H*W / 128) per sample
};

// CIN = 64: a K-tile is half of one tap's channels -> one scalar address shift per K-tile.
// CIN = 4 (first encoder conv, RGB padded to 4): a K-tile is 8 taps x 4 channels -> every 16-byte chunk of a row comes
// from its own tap, which LDS-DMA handles because the SOURCE address is per lane (only the LDS side is linear).
template <int CIN>
__global__ __launch_bounds__(256, 2) void conv3d_k3_gelu_stats_kernel(ConvArgs g) {
    constexpr int BM = VBM, BN = VC, WM = 64, WN = 32, BK = VBK;
    constexpr int KTOT = CIN == 64 ? 27 * 64 : 128;
    constexpr int TM = 2, TN = 1;
    constexpr int A_PIECES = BM / 32, B_PIECES = BN / 32;
    constexpr int STAGE = (BM + BN) * BK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int nwg = gridDim.x;
    int wg;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    const int smp = wg / g.tiles, tile = wg % g.tiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int THW = g.T * g.H * g.W;
    const int Hp = g.H + 2, Wp = g.W + 2;

    const int r8 = lane >> 3, pc = lane & 7;
    const float* a_src[A_PIECES];
    int a_c[A_PIECES];                 // logical 16-byte chunk this lane fetches (physical chunk pc, un-swizzled)
    const float* b_src[B_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int trow = (wave + 4 * i) * 8 + r8;
        int v = tile * BM + trow;
        v = v < THW ? v : THW - 1;
        const int w = v % g.W, h = (v / g.W) % g.H, t = v / (g.W * g.H);
        const int64_t pv = (((int64_t)smp * (g.T + 2) + t) * Hp + h) * Wp + w;      // tap (0,0,0) of this voxel
        a_c[i] = pc ^ ((trow >> 1) & 7);
        a_src[i] = g.X + pv * CIN + (CIN == 64 ? (a_c[i] << 2) : 0);
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
        const int trow = (wave + 4 * i) * 8 + r8;
        b_src[i] = g.Wt + (int64_t)trow * KTOT + ((pc ^ ((trow >> 1) & 7)) << 2);
    }
    auto stage = [&](int kt, int buf) {
        float* as = smem + buf * STAGE;
        float* bs = as + BM * BK;
        const int offB = kt * BK;
        if constexpr (CIN == 64) {
            const int tap = kt >> 1, half = kt & 1;
            const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            const int offA = ((dt * Hp + dh) * Wp + dw) * VC + half * BK;   // same shift for every voxel of the tile
#pragma unroll
            for (int i = 0; i < A_PIECES; ++i)
                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(a_src[i] + offA), AVD_LDS_PTR(as + (wave + 4 * i) * 8 * BK), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < A_PIECES; ++i) {
                int tap = kt * 8 + a_c[i];
                tap = tap < 27 ? tap : 0;                                    // taps 27..31 carry zero weights
                const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
                const int offA = ((dt * Hp + dh) * Wp + dw) * CIN;
                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(a_src[i] + offA), AVD_LDS_PTR(as + (wave + 4 * i) * 8 * BK), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i)
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(b_src[i] + offB), AVD_LDS_PTR(bs + (wave + 4 * i) * 8 * BK), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    constexpr int nk = KTOT / BK;   // 54 | 4
    stage(0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    int a_row[TM], a_sw[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WM + i * 32 + l31;
        a_row[i] = r * BK;
        a_sw[i] = (r >> 1) & 7;
    }
    const int br = wn * WN + l31;
    const int b_row = br * BK, b_sw = (br >> 1) & 7;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
        const float* as = smem + cur * STAGE;
        const float* bs = as + BM * BK;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 af[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(as + a_row[i] + (((2 * kk + hi) ^ a_sw[i]) << 2));
            const f32x4 bf = *reinterpret_cast<const f32x4*>(bs + b_row + (((2 * kk + hi) ^ b_sw) << 2));
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[s], acc[i][0], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }

    // ---- epilogue: slab -> bias + GELU -> NDHWC store + GroupNorm partial statistics ----
    constexpr int CLD = WN + 4;
    float* slab = smem + wave * WM * CLD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(i * 32 + mfma32_row(r, hi)) * CLD + l31] = acc[i][0][r];
    __syncthreads();

    constexpr int LPR = WN / 4, RPI = 64 / LPR, NIT = WM / RPI;     // 8 lanes per row, 8 rows per pass, 8 passes
    const int cr = lane / LPR, cc = (lane % LPR) * 4;
    const int n = wn * WN + cc;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + n);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int v = tile * BM + wm * WM + cr + it * RPI;
        f32x4 y = *reinterpret_cast<const f32x4*>(slab + (cr + it * RPI) * CLD + cc);
        y += bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = gelu_erf(y[e]);
        if (v < THW) {
            *reinterpret_cast<f32x4*>(g.Y + ((int64_t)smp * THW + v) * VC + n) = y;
            s1 += (y[0] + y[1]) + (y[2] + y[3]);
            s2 += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
        }
    }
    // lanes sharing a group: the two 4-channel chunks (lane bit 0) x the 8 rows of a pass (lane bits 3..5)
    s1 += __shfl_xor(s1, 1, 64);  s2 += __shfl_xor(s2, 1, 64);
    s1 += __shfl_xor(s1, 8, 64);  s2 += __shfl_xor(s2, 8, 64);
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    if ((lane & 0x39) == 0) {            // lanes 0,2,4,6: group wn*4 + lane/2
        float* p = g.part + ((((int64_t)smp * g.tiles + tile) * 2 + wm) * VG + wn * 4 + (lane >> 1)) * 2;
        p[0] = s1;
        p[1] = s2;
    }
}

// mean / rstd per (sample, group): fp64 reduction of the per-block partials in a fixed order, in two launches (round 5; one block of 1,024
// threads per sample read its 3 MB of partials at one CU's bandwidth: 80 us per launch at 256 x 256).  Stage 1: GN_FIN_CHUNKS blocks per
// sample, a thread reads whole 64-byte entries (the 8 groups x {sum, sum of squares} one conv wave wrote) and carries 16 fp64 sums, the
// block folds them with xor shuffles and one pass through LDS and writes 16 doubles; stage 2 adds the chunks in order.
constexpr int GN_FIN_CHUNKS = 64;
__global__ __launch_bounds__(256) void gn_finalize1_kernel(const float* __restrict__ part, double* __restrict__ fin, int tiles) {
    __shared__ double sh[4][2 * VG];
    const int chunk = blockIdx.x, smp = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[2 * VG];
#pragma unroll
    for (int k = 0; k < 2 * VG; ++k) acc[k] = 0.0;
    const int n = tiles * 2, per = (n + GN_FIN_CHUNKS - 1) / GN_FIN_CHUNKS;
    const int i1 = (chunk + 1) * per < n ? (chunk + 1) * per : n;
    const float* base = part + (int64_t)smp * n * (2 * VG);
    for (int i = chunk * per + tid; i < i1; i += 256) {
        const f32x4* e = reinterpret_cast<const f32x4*>(base + (int64_t)i * (2 * VG));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = e[q];
            acc[4 * q] += v[0]; acc[4 * q + 1] += v[1]; acc[4 * q + 2] += v[2]; acc[4 * q + 3] += v[3];
        }
    }
#pragma unroll
    for (int k = 0; k < 2 * VG; ++k) {
        double a = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        if (lane == 0) sh[wave][k] = a;
    }
    __syncthreads();
    if (tid < 2 * VG) fin[((int64_t)smp * GN_FIN_CHUNKS + chunk) * (2 * VG) + tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
}
__global__ __launch_bounds__(64) void gn_finalize2_kernel(const double* __restrict__ fin, float* __restrict__ stats, double count, float eps) {
    __shared__ double tot[2 * VG];
    const int smp = blockIdx.x, tid = threadIdx.x;
    if (tid < 2 * VG) {
        double a = 0.0;
        for (int c = 0; c < GN_FIN_CHUNKS; ++c) a += fin[((int64_t)smp * GN_FIN_CHUNKS + c) * (2 * VG) + tid];
        tot[tid] = a;
    }
    __syncthreads();
    if (tid < VG) {
        const double mean = tot[2 * tid] / count;
        double var = tot[2 * tid + 1] / count - mean * mean;     // biased, as torch.nn.GroupNorm
        if (var < 0.0) var = 0.0;
        stats[(smp * VG + tid) * 2 + 0] = (float)mean;
        stats[(smp * VG + tid) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}
static int64_t gn_fin_bytes(int B) { return (int64_t)B * GN_FIN_CHUNKS * 2 * VG * 8; }
static int gn_finalize(const float* part, double* fin, float* stats, int B, int tiles, double count, float eps, hipStream_t st) {
    hipLaunchKernelGGL(gn_finalize1_kernel, dim3(GN_FIN_CHUNKS, B), dim3(256), 0, st, part, fin, tiles);
    AVD_CHECK_LAUNCH("gn_finalize (1)");
    hipLaunchKernelGGL(gn_finalize2_kernel, dim3(B), dim3(64), 0, st, fin, stats, count, eps);
    AVD_CHECK_LAUNCH("gn_finalize (2)");
    return AVD_OK;
}

// zero the one-voxel halo of a padded activation buffer [B][T+2][H+2][W+2] voxels x rowb bytes (the interior is overwritten by its
// producer every time; a memset of the whole buffer moved 1.3 GB per decode at 256 x 256)
__global__ __launch_bounds__(256) void zero_halo_kernel(unsigned char* __restrict__ buf, int B, int T, int H, int W, int rowb) {
    const int Hp = H + 2, Wp = W + 2;
    const int64_t n_t = 2ll * Hp * Wp, n_h = 2ll * T * Wp, n_w = 2ll * T * H, per = n_t + n_h + n_w;
    const int c16 = rowb >> 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * per * c16) return;
    const int c = (int)(gid % c16);
    int64_t hidx = gid / c16;
    const int smp = (int)(hidx / per);
    hidx -= (int64_t)smp * per;
    int t, h, w;
    if (hidx < n_t) {                       // the two t faces, whole (H+2) x (W+2) planes
        t = hidx < n_t / 2 ? 0 : T + 1;
        const int64_t r = hidx % (n_t / 2);
        h = (int)(r / Wp); w = (int)(r % Wp);
    } else if (hidx < n_t + n_h) {          // the two h faces of the interior t range
        const int64_t r = hidx - n_t;
        const int64_t q = r % (n_h / 2);
        h = r < n_h / 2 ? 0 : H + 1;
        t = 1 + (int)(q / Wp); w = (int)(q % Wp);
    } else {                                // the two w faces of the interior (t, h) range
        const int64_t r = hidx - n_t - n_h;
        const int64_t q = r % (n_w / 2);
        w = r < n_w / 2 ? 0 : W + 1;
        t = 1 + (int)(q / H); h = 1 + (int)(q % H);
    }
    const int64_t pv = (((int64_t)smp * (T + 2) + t) * Hp + h) * Wp + w;
    *reinterpret_cast<u32x4*>(buf + pv * rowb + c * 16) = u32x4{0u, 0u, 0u, 0u};
}
static int zero_halo(void* buf, int B, int T, int H, int W, int rowb, hipStream_t st) {
    const int64_t per = 2ll * (H + 2) * (W + 2) + 2ll * T * (W + 2) + 2ll * T * H;
    const int64_t n = (int64_t)B * per * (rowb / 16);
    AVD_REQUIRE(rowb % 16 == 0 && (n + 255) / 256 < (1ll << 31), AVD_EUNSUPPORTED, "zero_halo: bad geometry");
    hipLaunchKernelGGL(zero_halo_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<unsigned char*>(buf), B, T, H, W, rowb);
    AVD_CHECK_LAUNCH("zero_halo");
    return AVD_OK;
}

// GroupNorm apply: Y (NDHWC) -> interior of the padded NDHWC buffer feeding the next conv
__global__ __launch_bounds__(256) void gn_apply_pad_kernel(const float* __restrict__ Y, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ Xp, int T, int H, int W, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % (VC / 4)) * 4;
    const int64_t vox = i / (VC / 4);
    const int THW = T * H * W;
    const int smp = (int)(vox / THW), v = (int)(vox % THW);
    const int w = v % W, h = (v / W) % H, t = v / (W * H);
    const float mean = stats[(smp * VG + c / 8) * 2], rstd = stats[(smp * VG + c / 8) * 2 + 1];
    const f32x4 y = *reinterpret_cast<const f32x4*>(Y + vox * VC + c);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (y[e] - mean) * rstd * gm[e] + bt[e];
    const int64_t pv = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
    *reinterpret_cast<f32x4*>(Xp + pv * VC + c) = o;
}

// last block: GroupNorm apply + to_img (1x1x1, 64 -> Cout<=4) + sigmoid/tanh, NDHWC -> NCDHW.
// A block takes 128 consecutive voxels: 16 lanes per voxel read one float4 of channels each (1 KiB per wave instruction, coalesced),
// the 64-channel dot products are finished with four xor shuffles, and the Cout values of the 128 voxels go through LDS so that
// every output plane is written as one 512-byte run (round 2 wrote them as 4-byte scattered stores: 1.9 TB/s; now the read rate).
constexpr int TOIMG_VOX = 128;
__global__ __launch_bounds__(256) void gn_apply_toimg_kernel(const float* __restrict__ Y, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ Wimg, const float* __restrict__ bimg,
                                                             float* __restrict__ out, int THW, int Cout, int use_tanh,
                                                             int64_t nvox) {
    __shared__ float s_out[4][TOIMG_VOX];
    const int tid = threadIdx.x;
    const int c = (tid & 15) * 4, vsub = tid >> 4;
    const int64_t vox0 = (int64_t)blockIdx.x * TOIMG_VOX;
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 w[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) w[o] = o < Cout ? *reinterpret_cast<const f32x4*>(Wimg + o * VC + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < TOIMG_VOX / 16; ++it) {
        const int vl = it * 16 + vsub;
        const int64_t vox = vox0 + vl;
        const int64_t vx = vox < nvox ? vox : nvox - 1;
        const int smp = (int)(vx / THW);
        const float mean = stats[(smp * VG + c / 8) * 2], rstd = stats[(smp * VG + c / 8) * 2 + 1];
        const f32x4 y = *reinterpret_cast<const f32x4*>(Y + vx * VC + c);
        float xn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xn[e] = (y[e] - mean) * rstd * gm[e] + bt[e];
#pragma unroll
        for (int o = 0; o < 4; ++o) {          // static indices keep everything in registers
            if (o < Cout) {
                float a = xn[0] * w[o][0] + xn[1] * w[o][1] + xn[2] * w[o][2] + xn[3] * w[o][3];
                a += __shfl_xor(a, 1, 64);
                a += __shfl_xor(a, 2, 64);
                a += __shfl_xor(a, 4, 64);
                a += __shfl_xor(a, 8, 64);
                if ((tid & 15) == o) s_out[o][vl] = a;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < TOIMG_VOX * Cout; idx += 256) {
        const int o = idx / TOIMG_VOX, vl = idx % TOIMG_VOX;
        const int64_t vox = vox0 + vl;
        if (vox < nvox) {
            float v = s_out[o][vl] + bimg[o];
            v = use_tanh ? tanhf(v) : 1.0f / (1.0f + expf(-v));
            const int64_t smp = vox / THW;
            out[(smp * Cout + o) * THW + (vox - smp * THW)] = v;
        }
    }
}

// from_lat: 1x1x1 conv Cv -> 64 on the latent grid, NCDHW in -> NDHWC out
__global__ __launch_bounds__(256) void fromlat_kernel(const float* __restrict__ z, const float* __restrict__ Wf,
                                                      const float* __restrict__ bf, float* __restrict__ hlow, int Cv,
                                                      int vol, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int o = (int)(i % VC);
    const int64_t vox = i / VC;
    const int smp = (int)(vox / vol), v = (int)(vox % vol);
    float a = 0.f;
    for (int c = 0; c < Cv; ++c) a += z[((int64_t)smp * Cv + c) * vol + v] * Wf[o * Cv + c];
    hlow[i] = a + bf[o];
}

// PyTorch upsample index rule for align_corners=False (area_pixel_compute_source_index)
__device__ __forceinline__ void tri_src(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l0 = 1.0f - l1;
}

// trilinear upsample of the NDHWC latent-grid features into the interior of the padded NDHWC conv input
__global__ __launch_bounds__(256) void upsample_pad_kernel(const float* __restrict__ hlow, float* __restrict__ Xp,
                                                           int Tp, int Hp_, int Wp_, int T, int H, int W, float st,
                                                           float sh, float sw, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const int c = (int)(i % (VC / 4)) * 4;
    const int64_t vox = i / (VC / 4);
    const int THW = T * H * W;
    const int smp = (int)(vox / THW), v = (int)(vox % THW);
    const int w = v % W, h = (v / W) % H, t = v / (W * H);
    int t0, t1, h0, h1, w0, w1;
    float tl0, tl1, hl0, hl1, wl0, wl1;
    tri_src(t, st, Tp, t0, t1, tl0, tl1);
    tri_src(h, sh, Hp_, h0, h1, hl0, hl1);
    tri_src(w, sw, Wp_, w0, w1, wl0, wl1);
    const float* base = hlow + (int64_t)smp * Tp * Hp_ * Wp_ * VC + c;
    auto at = [&](int tt, int hh, int ww) {
        return *reinterpret_cast<const f32x4*>(base + ((int64_t)(tt * Hp_ + hh) * Wp_ + ww) * VC);
    };
    // same association as torch's CPU kernel: t0l*(h0l*(w0l*a+w1l*b) + h1l*(w0l*c+w1l*d)) + t1l*(...)
    const f32x4 r = tl0 * (hl0 * (wl0 * at(t0, h0, w0) + wl1 * at(t0, h0, w1)) + hl1 * (wl0 * at(t0, h1, w0) + wl1 * at(t0, h1, w1))) +
                    tl1 * (hl0 * (wl0 * at(t1, h0, w0) + wl1 * at(t1, h0, w1)) + hl1 * (wl0 * at(t1, h1, w0) + wl1 * at(t1, h1, w1)));
    const int64_t pv = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
    *reinterpret_cast<f32x4*>(Xp + pv * VC + c) = r;
}

// ---- encoder side (vae_video3d.py:164-189): x [B,Cin<=4,T,H,W] NCDHW -> interior of the padded NDHWC4 buffer ----
__global__ __launch_bounds__(256) void rgb_to_ndhwc4_pad_kernel(const float* __restrict__ x, float* __restrict__ Xp4,
                                                                int Cin, int T, int H, int W, int64_t nvox) {
    const int64_t vox = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vox >= nvox) return;
    const int THW = T * H * W;
    const int smp = (int)(vox / THW), v = (int)(vox % THW);
    const int w = v % W, h = (v / W) % H, t = v / (W * H);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cin; ++c) o[c] = x[((int64_t)smp * Cin + c) * THW + v];
    const int64_t pv = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
    *reinterpret_cast<f32x4*>(Xp4 + pv * 4) = o;
}

// x [B, Cin <= 8, T, H, W] -> interior of the 96-byte-per-voxel image ([plane][16 ch], channels >= Cin zero) the packed-tap form of the
// halo-tile conv reads (the encoder's first convolution on the matrix pipe, round 5): one thread per voxel, coalesced per channel plane
__global__ __launch_bounds__(256) void rgb_lat16_kernel(const float* __restrict__ x, unsigned char* __restrict__ X16, int Cin, int T, int H, int W,
                                                        int64_t nvox) {
    const int64_t vox = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vox >= nvox) return;
    const int THW = T * H * W;
    const int smp = (int)(vox / THW), v = (int)(vox % THW);
    const int w = v % W, h = (v / W) % H, t = v / (W * H);
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = c < Cin ? x[((int64_t)smp * Cin + c) * THW + v] : 0.f;
    const int64_t pv = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
    unsigned char* dst = X16 + pv * 96;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    u32x4 Hh, Mi, Lo;
    split8(o, Hh, Mi, Lo);
    *reinterpret_cast<u32x4*>(dst) = Hh;
    *reinterpret_cast<u32x4*>(dst + 16) = zero;
    *reinterpret_cast<u32x4*>(dst + 32) = Mi;
    *reinterpret_cast<u32x4*>(dst + 48) = zero;
    *reinterpret_cast<u32x4*>(dst + 64) = Lo;
    *reinterpret_cast<u32x4*>(dst + 80) = zero;
}

// GroupNorm apply + AvgPool3d(td, sd, sd) + to_lat (1x1x1, 64 -> Cl <= 16): one wave per latent voxel.
// Pooling and the per-channel GroupNorm affine commute, so the window mean is normalised once.
__global__ __launch_bounds__(256) void gn_pool_tolat_kernel(const float* __restrict__ Y, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ Wl, const float* __restrict__ bl,
                                                            float* __restrict__ z, int T, int H, int W, int td, int sd,
                                                            int Cl, int64_t nlat) {
    const int lane = threadIdx.x & 63;
    const int64_t lv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lv >= nlat) return;
    const int Tl = T / td, Hl = H / sd, Wl_ = W / sd;
    const int vol = Tl * Hl * Wl_;
    const int smp = (int)(lv / vol), r = (int)(lv % vol);
    const int wl = r % Wl_, hl = (r / Wl_) % Hl, tl = r / (Wl_ * Hl);
    const int c = (lane & 15) * 4, sub = lane >> 4;             // 16 lanes per voxel, 4 voxels per pass
    const int win = td * sd * sd;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int j = sub; j < win; j += 4) {
        const int ww = j % sd, hh = (j / sd) % sd, tt = j / (sd * sd);
        const int64_t v = (((int64_t)smp * T + tl * td + tt) * H + hl * sd + hh) * W + wl * sd + ww;
        s += *reinterpret_cast<const f32x4*>(Y + v * VC + c);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s[e] += __shfl_xor(s[e], 16, 64);
        s[e] += __shfl_xor(s[e], 32, 64);
    }
    const float inv = 1.0f / (float)win;
    const float mean = stats[(smp * VG + c / 8) * 2], rstd = stats[(smp * VG + c / 8) * 2 + 1];
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c), bt = *reinterpret_cast<const f32x4*>(beta + c);
    float xn[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xn[e] = (s[e] * inv - mean) * rstd * gm[e] + bt[e];
    for (int o = 0; o < Cl; ++o) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(Wl + o * VC + c);
        float a = xn[0] * wv[0] + xn[1] * wv[1] + xn[2] * wv[2] + xn[3] * wv[3];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        a += __shfl_xor(a, 8, 64);
        if (lane == o) z[((int64_t)smp * Cl + o) * vol + r] = a + bl[o];
    }
}

// The same from the pooling partial sums of the halo-tile conv's OUT = 3 epilogue (t_down = 4, s_down = 8: a tile of 4 x TH x 16 output
// voxels is one pooling block in t, TH / 8 in h, two in w): a latent voxel gathers its 8 entries (4 t-slices = waves, 8 / 4 h passes), each
// [w block][64 channels], normalises the mean with the GroupNorm statistics and applies to_lat.  One wave per latent voxel, lane = channel.
template <int TH>
__global__ __launch_bounds__(256) void pool_tolat_from_partials_kernel(const float* __restrict__ P, const float* __restrict__ stats,
                                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                       const float* __restrict__ Wl, const float* __restrict__ bl,
                                                                       float* __restrict__ z, int T, int H, int W, int tiles2, int Cl, int64_t nlat) {
    const int c = threadIdx.x & 63;
    const int64_t lv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lv >= nlat) return;
    const int Tl = T / 4, Hl = H / 8, Wl_ = W / 8;
    const int vol = Tl * Hl * Wl_;
    const int smp = (int)(lv / vol), r = (int)(lv % vol);
    const int wl = r % Wl_, hl = (r / Wl_) % Hl, tl = r / (Wl_ * Hl);
    const int nw_t = (W + 15) / 16, nh_t = (H + TH - 1) / TH;
    constexpr int PASSES = TH / 4;                 // 64-voxel passes (4 h-rows x 16 w) of a wave per tile
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 8 / 4; ++k) {             // the two 4-row passes of the 8-row pooling block
        const int hrow = hl * 8 + 4 * k;           // first output row of the pass
        const int tile = (tl * nh_t + hrow / TH) * nw_t + wl / 2, ps = (hrow % TH) / 4;
#pragma unroll
        for (int wave = 0; wave < 4; ++wave) {
            const int64_t e = ((int64_t)tile * 4 + wave) * PASSES + ps;
            sum += P[((((int64_t)smp * tiles2 + e) * 2) + (wl & 1)) * VC + c];
        }
    }
    const float mean = stats[(smp * VG + c / 8) * 2], rstd = stats[(smp * VG + c / 8) * 2 + 1];
    const float xn = (sum * (1.0f / 256.0f) - mean) * rstd * gamma[c] + beta[c];
    for (int o = 0; o < Cl; ++o) {
        float a = xn * Wl[o * VC + c];
#pragma unroll
        for (int x = 1; x < 64; x <<= 1) a += __shfl_xor(a, x, 64);
        if (c == o) z[((int64_t)smp * Cl + o) * vol + r] = a + bl[o];
    }
}

// =========================================================================================================
// bf16x3 decoder path: the two 3x3x3 convolutions on the bf16 matrix pipe with exactly split operands (x = h + m + l, six
// product terms, fp32 accumulation — rationale and error analysis in gemm_bf16x3.hip).  The zero-haloed activation buffer
// then holds, per voxel, 384 B = [plane h | m | l][64 channels bf16] ("act3"); the producers (trilinear upsample, GroupNorm
// apply) write it directly, so no fp32 copy of the conv inputs exists.  Weights become a per-(tap, half-tap) image
// [54][3 planes][64 out][32 ch] with its 16-byte chunks pre-swizzled for the LDS reads.
// =========================================================================================================
constexpr int A3_ROWB = 384;                   // bytes per voxel of the act3 buffer
constexpr int W3_STAGE = 3 * 64 * 32;          // weight bytes per (16-channel slab, tap) stage: [plane][64 out][32 B]
constexpr int W3_BYTES = 108 * W3_STAGE;

struct Conv3Args {
    const unsigned char* X3;   // act3, padded [B, T+2, H+2, W+2] voxels x 384 B
    const unsigned char* W3;   // weight image
    const float* bias;
    float* Y;                  // NDHWC fp32 [B, T, H, W, 64]
    float* part;
    int T, H, W, tiles;
    // f16x2 (TERMS 3, two fp16 planes at a power-of-two scale, avd_common.h): the sums are multiplied by ab_inv = 1 / (s_act s_w);
    // when the activation scale was derived on the device (decoder block 0) a_inv_dev points at 1 / s_act and ab_inv is 1 / s_w
    float ab_inv;
    const float* a_inv_dev;
    // latent-composed first decoder conv (LAT instantiations, see upsample_lat16_kernel): per border class, what the from_lat bias
    // contributes through the taps that fall inside the volume — btab[class][64 out], class = 6 bits (t lo, t hi, h lo, h hi, w lo, w hi)
    const float* btab;
    // "folded" decoder route (round 5, bf16x3): the conv kernel's OUT modes.  OUT = 1: GELU(y) goes out as the NEXT conv's act3 image
    // (X3out, interior of the padded buffer) instead of fp32 Y — the GroupNorm between the two convs is folded into the next conv's
    // per-sample weights (conv3_weight_gn_kernel) and bias table (conv3_gn_btab_kernel).  OUT = 2 (last conv): per voxel and GroupNorm
    // group the partial to_img sums P[vox][8 groups][4] = sum_{c in g} wimg_g[o][c] y_c (toimg_from_p_kernel finishes the 1x1x1 conv).
    unsigned char* X3out;
    float* P;
    const float* wimg_g;       // [4][64] = to_img_w[o][c] gamma[c], rows >= out_ch zero
    int64_t w3_stride;         // bytes between the samples' weight images (0: one image for all samples)
    int btab_stride;           // floats between the samples' bias tables (0: one table)
    // slab-major operand images (see the kernel): bytes between the four 96-byte-per-voxel slab images of the INPUT (64-channel convs) /
    // of the image OUT = 1 writes; 0 = voxel-major act3 rows of 384 B
    int64_t x_slab_stride, out_slab_stride;
};

// scale slot in the workspace: [0] max |x|, [1] (max row norm, unused), [2] s, [3] 1 / s
__global__ void pow2_scale_kernel(float* ws) {
    const float amax = ws[0] * 1.01f;
    float s = 1.0f;
    if (amax > 0.f) {
        float e = 15.0f - ceilf(log2f(amax));
        e = fminf(fmaxf(e, -100.f), 100.f);
        s = exp2f(e);
    }
    if (!(amax == amax) || amax > 3.0e38f) s = __builtin_nanf("");      // NaN / inf input: poison the image, never saturate silently
    ws[2] = s;
    ws[3] = 1.0f / s;
}

// Wt [64 out][27 taps][64 in] fp32 -> weight image of the halo-tile kernel: stage kt = 27 * slab + tap (slab = 16 input channels)
// holds [plane][64 out][32 B]; the 16-byte chunk `half` (8 in-channels) of row `out` sits at slot half ^ ((out >> 3) & 1)
template <bool F16>
__global__ __launch_bounds__(256) void conv3_weight_kernel(const float* __restrict__ Wt, unsigned char* __restrict__ img, float sc) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // one thread = 8 in-channels of one (stage, out)
    if (i >= 108 * 64 * 2) return;
    const int half = i & 1, out = (i >> 1) & 63, kt = i >> 7;
    const int slab = kt / 27, tap = kt % 27;
    float v[8];
    const float* src = Wt + ((int64_t)out * 27 + tap) * VC + slab * 16 + half * 8;
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + 4);
    unsigned char* dst = img + (int64_t)kt * W3_STAGE + out * 32 + ((half ^ ((out >> 3) & 1)) << 4);
    if constexpr (F16) {
        u32x4 Hh, Lo;
        split8_h2(v, sc, Hh, Lo);
        *reinterpret_cast<u32x4*>(dst) = Hh;
        *reinterpret_cast<u32x4*>(dst + 2048) = Lo;
    } else {
        u32x4 Hh, Mi, Lo;
        split8(v, Hh, Mi, Lo);
        *reinterpret_cast<u32x4*>(dst) = Hh;
        *reinterpret_cast<u32x4*>(dst + 2048) = Mi;
        *reinterpret_cast<u32x4*>(dst + 4096) = Lo;
    }
}

// ---- GroupNorm folded into the NEXT convolution (round 5, three-plane mode).  conv(GN(y)) with GN(y)_c = a_c y_c + s_c per sample
// (a_c = rstd_g gamma_c, s_c = beta_c - mean_g rstd_g gamma_c) is a convolution of y itself with the weights W[o, tap, c] a_c plus what the
// shift contributes through the taps that fall inside the volume (the conv zero-pads GN's OUTPUT) — a bias per border class, as for
// from_lat's bias in the latent-composed first conv.  So the producing conv writes GELU(y) straight into the act3 image and the
// normalise pass (read 256 B + write 384 B per voxel) disappears; the price is one 648-KiB weight image and one 16-KiB table per sample.
__global__ __launch_bounds__(256) void conv3_weight_gn_kernel(const float* __restrict__ Wt, const float* __restrict__ stats,
                                                              const float* __restrict__ gamma, unsigned char* __restrict__ img) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // one thread = 8 in-channels (one group) of one (stage, out)
    if (i >= 108 * 64 * 2) return;
    const int smp = blockIdx.y;
    const int half = i & 1, out = (i >> 1) & 63, kt = i >> 7;
    const int slab = kt / 27, tap = kt % 27;
    const int c0 = slab * 16 + half * 8;
    const float rstd = stats[(smp * VG + c0 / 8) * 2 + 1];
    float v[8], gm[8];
    const float* src = Wt + ((int64_t)out * 27 + tap) * VC + c0;
    *reinterpret_cast<f32x4*>(v) = *reinterpret_cast<const f32x4*>(src);
    *reinterpret_cast<f32x4*>(v + 4) = *reinterpret_cast<const f32x4*>(src + 4);
    *reinterpret_cast<f32x4*>(gm) = *reinterpret_cast<const f32x4*>(gamma + c0);
    *reinterpret_cast<f32x4*>(gm + 4) = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= rstd * gm[e];
    unsigned char* dst = img + (int64_t)smp * W3_BYTES + (int64_t)kt * W3_STAGE + out * 32 + ((half ^ ((out >> 3) & 1)) << 4);
    u32x4 Hh, Mi, Lo;
    split8(v, Hh, Mi, Lo);
    *reinterpret_cast<u32x4*>(dst) = Hh;
    *reinterpret_cast<u32x4*>(dst + 2048) = Mi;
    *reinterpret_cast<u32x4*>(dst + 4096) = Lo;
}
// btab[smp][class][out] = sum over the taps inside the volume for that border class of sum_c W[out, tap, c] s_c
// (class bits: t-1, t+1, h-1, h+1, w-1, w+1 inside — the conv kernel's epilogue)
__global__ __launch_bounds__(256) void conv3_gn_btab_kernel(const float* __restrict__ Wt, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ btab) {
    // one block per (sample, 8 output channels): 216 dot products of 64, then the 64 classes x 8 outputs (one block per sample walked the
    // 442 KB of weights through one CU: 34 us)
    __shared__ float shift[VC];
    __shared__ float S[8][28];
    const int smp = blockIdx.x, o0 = blockIdx.y * 8, tid = threadIdx.x;
    if (tid < VC) {
        const float mean = stats[(smp * VG + tid / 8) * 2], rstd = stats[(smp * VG + tid / 8) * 2 + 1];
        shift[tid] = beta[tid] - mean * rstd * gamma[tid];
    }
    __syncthreads();
    if (tid < 8 * 27) {
        const int o = tid / 27, tap = tid - o * 27;
        const float* wr = Wt + ((int64_t)(o0 + o) * 27 + tap) * VC;
        float a = 0.f;
        for (int c = 0; c < VC; c += 4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(wr + c);
            a = fmaf(w4[0], shift[c], a); a = fmaf(w4[1], shift[c + 1], a); a = fmaf(w4[2], shift[c + 2], a); a = fmaf(w4[3], shift[c + 3], a);
        }
        S[o][tap] = a;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * 8; idx += 256) {
        const int cls = idx >> 3, o = idx & 7;
        float a = 0.f;
        for (int tap = 0; tap < 27; ++tap) {
            const int dt = tap / 9, dh = (tap / 3) % 3, dw = tap % 3;
            const bool in = (dt != 0 || (cls & 1)) && (dt != 2 || (cls & 2)) && (dh != 0 || (cls & 4)) && (dh != 2 || (cls & 8)) &&
                            (dw != 0 || (cls & 16)) && (dw != 2 || (cls & 32));
            if (in) a += S[o][tap];
        }
        btab[((int64_t)smp * 64 + cls) * VC + o0 + o] = a;
    }
}
// to_img after the last GroupNorm, from the conv's partial sums: out[o] = sum_g rstd_g P[g][o] + K[o],
// K[o] = to_img_b[o] + sum_c to_img_w[o][c] (beta_c - mean_g rstd_g gamma_c); consts[smp] = {rstd[8], K[4]}
__global__ __launch_bounds__(64) void toimg_consts_kernel(const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ Wimg,
                                                          const float* __restrict__ bimg, float* __restrict__ consts, int Cout) {
    const int smp = blockIdx.x, c = threadIdx.x;
    const float mean = stats[(smp * VG + c / 8) * 2], rstd = stats[(smp * VG + c / 8) * 2 + 1];
    const float sh = beta[c] - mean * rstd * gamma[c];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        float a = o < Cout ? Wimg[o * VC + c] * sh : 0.f;
#pragma unroll
        for (int x = 32; x > 0; x >>= 1) a += __shfl_xor(a, x, 64);
        if (c == 0) consts[smp * 12 + VG + o] = a + (o < Cout ? bimg[o] : 0.f);
    }
    if ((c & 7) == 0) consts[smp * 12 + c / 8] = rstd;
}
// wimg_g[o][c] = to_img_w[o][c] gamma[c] (rows >= Cout zero): what the last conv's epilogue multiplies GELU(y) with
__global__ __launch_bounds__(256) void toimg_wg_kernel(const float* __restrict__ Wimg, const float* __restrict__ gamma, float* __restrict__ wg, int Cout) {
    const int i = threadIdx.x, o = i >> 6, c = i & 63;
    wg[i] = o < Cout ? Wimg[o * VC + c] * gamma[c] : 0.f;
}
constexpr int TOIMG_P_VOX = 256;
// one thread per voxel: its 8 groups x 4 partial sums are one 128-byte line (a wave reads 8 KiB contiguous), the 32 multiply-adds need no
// cross-lane traffic, and consecutive threads write consecutive positions of each output plane
__global__ __launch_bounds__(256) void toimg_from_p_kernel(const float* __restrict__ P, const float* __restrict__ consts, float* __restrict__ out,
                                                           int THW, int Cout, int use_tanh, int64_t nvox) {
    const int64_t vox = (int64_t)blockIdx.x * TOIMG_P_VOX + threadIdx.x;
    if (vox >= nvox) return;
    const int64_t smp = vox / THW;
    const float* cs = consts + smp * 12;
    const f32x4* p = reinterpret_cast<const f32x4*>(P + vox * (VG * 4));
    f32x4 a = {cs[VG], cs[VG + 1], cs[VG + 2], cs[VG + 3]};
#pragma unroll
    for (int g = 0; g < VG; ++g) a += p[g] * cs[g];
    const int64_t v = vox - smp * THW;
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (o < Cout) {
            const float x = a[o];
            out[(smp * Cout + o) * THW + v] = use_tanh ? tanhf(x) : 1.0f / (1.0f + expf(-x));
        }
}

// slab_stride != 0: the slab-major form (conv3d_k3_bf16x3_kernel) — four images of 96 B per voxel = [plane][16 ch] (two planes: 64 B, so that
// every byte of a row is written: rows with holes cost the memory side a read-modify-write), slab c8 / 2, its channels 8 (c8 % 2) .. + 7;
// 0: voxel-major rows of 384 B = [plane][64 ch]
template <bool F16>
__device__ __forceinline__ void store_act3(unsigned char* X3, int64_t pv, int c8, const float* v, float sc, int64_t slab_stride = 0) {
    unsigned char* dst = slab_stride ? X3 + (int64_t)(c8 >> 1) * slab_stride + pv * (F16 ? 64 : 96) + (c8 & 1) * 16 : X3 + pv * A3_ROWB + c8 * 16;
    const int pls = slab_stride ? 32 : 128;
    if constexpr (F16) {
        u32x4 Hh, Lo;
        split8_h2(v, sc, Hh, Lo);
        *reinterpret_cast<u32x4*>(dst) = Hh;
        *reinterpret_cast<u32x4*>(dst + pls) = Lo;
    } else {
        u32x4 Hh, Mi, Lo;
        split8(v, Hh, Mi, Lo);
        *reinterpret_cast<u32x4*>(dst) = Hh;
        *reinterpret_cast<u32x4*>(dst + pls) = Mi;
        *reinterpret_cast<u32x4*>(dst + 2 * pls) = Lo;
    }
}

// trilinear upsample into the interior of the act3 buffer (same arithmetic as upsample_pad_kernel, 8 channels per thread)
template <bool F16>      // F16: the image scale is read from the device (sc_dev[0], written by pow2_scale_kernel)
__global__ __launch_bounds__(256) void upsample_pad3_kernel(const float* __restrict__ hlow, unsigned char* __restrict__ X3,
                                                            int Tp, int Hp_, int Wp_, int T, int H, int W, float st,
                                                            float sh, float sw, int64_t total8, const float* __restrict__ sc_dev, int64_t slab_stride) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total8) return;
    const int c8 = (int)(i & 7);
    const int64_t vox = i >> 3;
    const int THW = T * H * W;
    const int smp = (int)(vox / THW), v = (int)(vox % THW);
    const int w = v % W, h = (v / W) % H, t = v / (W * H);
    int t0, t1, h0, h1, w0, w1;
    float tl0, tl1, hl0, hl1, wl0, wl1;
    tri_src(t, st, Tp, t0, t1, tl0, tl1);
    tri_src(h, sh, Hp_, h0, h1, hl0, hl1);
    tri_src(w, sw, Wp_, w0, w1, wl0, wl1);
    float o[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float* base = hlow + (int64_t)smp * Tp * Hp_ * Wp_ * VC + c8 * 8 + q * 4;
        auto at = [&](int tt, int hh, int ww) {
            return *reinterpret_cast<const f32x4*>(base + ((int64_t)(tt * Hp_ + hh) * Wp_ + ww) * VC);
        };
        const f32x4 r = tl0 * (hl0 * (wl0 * at(t0, h0, w0) + wl1 * at(t0, h0, w1)) + hl1 * (wl0 * at(t0, h1, w0) + wl1 * at(t0, h1, w1))) +
                        tl1 * (hl0 * (wl0 * at(t1, h0, w0) + wl1 * at(t1, h0, w1)) + hl1 * (wl0 * at(t1, h1, w0) + wl1 * at(t1, h1, w1)));
        o[4 * q] = r[0]; o[4 * q + 1] = r[1]; o[4 * q + 2] = r[2]; o[4 * q + 3] = r[3];
    }
    const int64_t pv = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
    store_act3<F16>(X3, pv, c8, o, F16 ? sc_dev[0] : 0.f, slab_stride);
}

// GroupNorm apply: Y (NDHWC fp32) -> interior of the act3 buffer feeding the next conv.  A block takes 64 consecutive voxels: one thread
// = one (voxel, 16-channel slab) = two GroupNorm groups reads 64 bytes (four neighbouring threads: one voxel's 256), normalises, splits, and
// parks its row of the slab-major image — [plane][16 ch], 96 / 64 bytes — in LDS as [slab][voxel][piece]; then the four slab images are
// written out 16 bytes per lane in row order: 64 consecutive voxels of a slab are one 6 / 4 KiB run (rows of one w-line are contiguous).
// (Written straight from the registers, a store instruction scattered 16-byte pieces over four images: 0.30 -> 0.46 ms for two planes.)
template <bool F16>
__global__ __launch_bounds__(256) void gn_apply_pad3_kernel(const float* __restrict__ Y, const float* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            unsigned char* __restrict__ X3, int T, int H, int W, int64_t nvox, float sc, int64_t slab_stride) {
    constexpr int NP = F16 ? 4 : 6, ROWB = NP * 16;            // 16-byte pieces / bytes of a slab row: (plane, 8-channel half)
    __shared__ u32x4 stage[4][64][NP];
    __shared__ int64_t spv[64];
    const int tid = threadIdx.x, vl = tid >> 2, slab = tid & 3;
    const int64_t vox0 = (int64_t)blockIdx.x * 64, vox = vox0 + vl;
    const int THW = T * H * W;
    if (vox < nvox) {
        const int smp = (int)(vox / THW), v = (int)(vox % THW);
        const int w = v % W, h = (v / W) % H, t = v / (W * H);
        if (slab == 0) spv[vl] = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int c8 = slab * 2 + half;
            const float mean = stats[(smp * VG + c8) * 2], rstd = stats[(smp * VG + c8) * 2 + 1];
            float y[8], gm[8], bt[8], o[8];
            *reinterpret_cast<f32x4*>(y) = *reinterpret_cast<const f32x4*>(Y + vox * VC + c8 * 8);
            *reinterpret_cast<f32x4*>(y + 4) = *reinterpret_cast<const f32x4*>(Y + vox * VC + c8 * 8 + 4);
            *reinterpret_cast<f32x4*>(gm) = *reinterpret_cast<const f32x4*>(gamma + c8 * 8);
            *reinterpret_cast<f32x4*>(gm + 4) = *reinterpret_cast<const f32x4*>(gamma + c8 * 8 + 4);
            *reinterpret_cast<f32x4*>(bt) = *reinterpret_cast<const f32x4*>(beta + c8 * 8);
            *reinterpret_cast<f32x4*>(bt + 4) = *reinterpret_cast<const f32x4*>(beta + c8 * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (y[e] - mean) * rstd * gm[e] + bt[e];
            if constexpr (F16) {
                u32x4 Hh, Lo;
                split8_h2(o, sc, Hh, Lo);
                stage[slab][vl][half] = Hh;
                stage[slab][vl][2 + half] = Lo;
            } else {
                u32x4 Hh, Mi, Lo;
                split8(o, Hh, Mi, Lo);
                stage[slab][vl][half] = Hh;
                stage[slab][vl][2 + half] = Mi;
                stage[slab][vl][4 + half] = Lo;
            }
        }
    }
    __syncthreads();
    const int nv = nvox - vox0 < 64 ? (int)(nvox - vox0) : 64;
    for (int idx = tid; idx < 4 * 64 * NP; idx += 256) {
        const int sl = idx / (64 * NP), rem = idx - sl * (64 * NP), vv = rem / NP, pc = rem - vv * NP;
        if (vv < nv) {
            unsigned char* dst = slab_stride ? X3 + (int64_t)sl * slab_stride + spv[vv] * ROWB + pc * 16
                                             : X3 + spv[vv] * A3_ROWB + (pc >> 1) * 128 + (sl * 2 + (pc & 1)) * 16;
            *reinterpret_cast<u32x4*>(dst) = stage[sl][vv][pc];
        }
    }
}

// ---- input image of the latent-composed first decoder conv: per padded voxel 96 B = [plane h | m | l][16 ch bf16] (f16x2: planes h, l at
// the device-derived scale), channels 0 .. Cv-1 = trilinear upsample of the LATENT z itself (NCDHW, Cv <= 16), the rest zero.  One thread
// per output voxel: 8 neighbours x Cv channels of a latent that is a few hundred KB per sample (L2-resident), 96 contiguous bytes out.
constexpr int L16_ROWB = 96;
template <bool F16>
__global__ __launch_bounds__(256) void upsample_lat16_kernel(const float* __restrict__ z, unsigned char* __restrict__ X16, int Cv, int Tp,
                                                             int Hp_, int Wp_, int T, int H, int W, float st, float sh, float sw,
                                                             int64_t nvox, const float* __restrict__ sc_dev) {
    const int64_t vox = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vox >= nvox) return;
    const int THW = T * H * W;
    const int smp = (int)(vox / THW), v = (int)(vox % THW);
    const int w = v % W, h = (v / W) % H, t = v / (W * H);
    int t0, t1, h0, h1, w0, w1;
    float tl0, tl1, hl0, hl1, wl0, wl1;
    tri_src(t, st, Tp, t0, t1, tl0, tl1);
    tri_src(h, sh, Hp_, h0, h1, hl0, hl1);
    tri_src(w, sw, Wp_, w0, w1, wl0, wl1);
    const int vol = Tp * Hp_ * Wp_;
    const int o00 = (t0 * Hp_ + h0) * Wp_, o01 = (t0 * Hp_ + h1) * Wp_, o10 = (t1 * Hp_ + h0) * Wp_, o11 = (t1 * Hp_ + h1) * Wp_;
    float o[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        o[c] = 0.f;
        if (c < Cv) {
            const float* b = z + ((int64_t)smp * Cv + c) * vol;
            // the association of torch's CPU kernel (and of upsample_pad3_kernel)
            o[c] = tl0 * (hl0 * (wl0 * b[o00 + w0] + wl1 * b[o00 + w1]) + hl1 * (wl0 * b[o01 + w0] + wl1 * b[o01 + w1])) +
                   tl1 * (hl0 * (wl0 * b[o10 + w0] + wl1 * b[o10 + w1]) + hl1 * (wl0 * b[o11 + w0] + wl1 * b[o11 + w1]));
        }
    }
    const int64_t pv = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
    unsigned char* dst = X16 + pv * L16_ROWB;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if constexpr (F16) {
            u32x4 Hh, Lo;
            split8_h2(o + 8 * half, sc_dev[0], Hh, Lo);
            *reinterpret_cast<u32x4*>(dst + half * 16) = Hh;
            *reinterpret_cast<u32x4*>(dst + 32 + half * 16) = Lo;
        } else {
            u32x4 Hh, Mi, Lo;
            split8(o + 8 * half, Hh, Mi, Lo);
            *reinterpret_cast<u32x4*>(dst + half * 16) = Hh;
            *reinterpret_cast<u32x4*>(dst + 32 + half * 16) = Mi;
            *reinterpret_cast<u32x4*>(dst + 64 + half * 16) = Lo;
        }
    }
}

// The same for lat_ch <= 8 from a channel-LAST copy of the latent (lat_cl8_kernel: [B, vol, 8], channels >= Cv zero): a thread's 8 neighbours
// are 8 x two 16-byte loads instead of 64 four-byte ones, channels 8 .. 15 of the image are stored as zeros without going through the split.
__global__ __launch_bounds__(256) void lat_cl8_kernel(const float* __restrict__ z, float* __restrict__ zt, int Cv, int vol, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;      // one thread = one (sample, latent voxel)
    if (i >= total) return;
    const int64_t smp = i / vol, v = i - smp * vol;
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = c < Cv ? z[(smp * Cv + c) * vol + v] : 0.f;
    *reinterpret_cast<f32x4*>(zt + i * 8) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(zt + i * 8 + 4) = f32x4{o[4], o[5], o[6], o[7]};
}
template <bool F16>
__global__ __launch_bounds__(256) void upsample_lat8_kernel(const float* __restrict__ zt, unsigned char* __restrict__ X16, int Tp, int Hp_, int Wp_,
                                                            int T, int H, int W, float st, float sh, float sw, int64_t nvox,
                                                            const float* __restrict__ sc_dev) {
    const int64_t vox = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vox >= nvox) return;
    const int THW = T * H * W;
    const int smp = (int)(vox / THW), v = (int)(vox % THW);
    const int w = v % W, h = (v / W) % H, t = v / (W * H);
    int t0, t1, h0, h1, w0, w1;
    float tl0, tl1, hl0, hl1, wl0, wl1;
    tri_src(t, st, Tp, t0, t1, tl0, tl1);
    tri_src(h, sh, Hp_, h0, h1, hl0, hl1);
    tri_src(w, sw, Wp_, w0, w1, wl0, wl1);
    const float* b = zt + (int64_t)smp * Tp * Hp_ * Wp_ * 8;
    const int o00 = (t0 * Hp_ + h0) * Wp_, o01 = (t0 * Hp_ + h1) * Wp_, o10 = (t1 * Hp_ + h0) * Wp_, o11 = (t1 * Hp_ + h1) * Wp_;
    float o[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        auto at = [&](int row, int ww) { return *reinterpret_cast<const f32x4*>(b + (int64_t)(row + ww) * 8 + 4 * q); };
        // the association of torch's CPU kernel (and of upsample_pad3_kernel)
        const f32x4 r = tl0 * (hl0 * (wl0 * at(o00, w0) + wl1 * at(o00, w1)) + hl1 * (wl0 * at(o01, w0) + wl1 * at(o01, w1))) +
                        tl1 * (hl0 * (wl0 * at(o10, w0) + wl1 * at(o10, w1)) + hl1 * (wl0 * at(o11, w0) + wl1 * at(o11, w1)));
        o[4 * q] = r[0]; o[4 * q + 1] = r[1]; o[4 * q + 2] = r[2]; o[4 * q + 3] = r[3];
    }
    const int64_t pv = (((int64_t)smp * (T + 2) + t + 1) * (H + 2) + h + 1) * (W + 2) + w + 1;
    unsigned char* dst = X16 + pv * L16_ROWB;
    const u32x4 zero = {0u, 0u, 0u, 0u};
    if constexpr (F16) {
        u32x4 Hh, Lo;
        split8_h2(o, sc_dev[0], Hh, Lo);
        *reinterpret_cast<u32x4*>(dst) = Hh;
        *reinterpret_cast<u32x4*>(dst + 16) = zero;
        *reinterpret_cast<u32x4*>(dst + 32) = Lo;
        *reinterpret_cast<u32x4*>(dst + 48) = zero;
    } else {
        u32x4 Hh, Mi, Lo;
        split8(o, Hh, Mi, Lo);
        *reinterpret_cast<u32x4*>(dst) = Hh;
        *reinterpret_cast<u32x4*>(dst + 16) = zero;
        *reinterpret_cast<u32x4*>(dst + 32) = Mi;
        *reinterpret_cast<u32x4*>(dst + 48) = zero;
        *reinterpret_cast<u32x4*>(dst + 64) = Lo;
        *reinterpret_cast<u32x4*>(dst + 80) = zero;
    }
}

// conv 3x3x3 64 -> 64 + bias + GELU + GroupNorm partial statistics on the 16-bit matrix pipe — HALO-TILE kernel (round 3).
// Rounds 1-2 ran this as an implicit GEMM that re-fetched the block's A tile from global memory for every tap (27 x), 12-24 MFMAs
// per wave between barriers on a 64 x 32 wave tile: 0.30 of the matrix peak, a K stage of ~2,150 cycles of which ~770 were MFMA.
// Here a block owns an OUTPUT tile of 4 x TH x 16 voxels (TH = 8 for two planes, 4 for three) and, per slab of 16 input channels,
// stages the INPUT tile with its one-voxel halo — 6 x (TH+2) x 18 voxels — in LDS ONCE; the 27 taps are then 27 MFMA k-steps whose A
// fragments are read from that tile at constant row shifts, and only the 4-6 KiB weight stage of a (slab, tap) is moved per step
// (ring of three).  Global -> LDS traffic per MFMA drops ~9x; a wave owns one t-slice: TH x 16 voxels x 64 output channels
// (TM = TH / 2 row tiles of 2 x 16 voxels, 2 column tiles), 24 MFMAs per step, fragments one step ahead in registers.
// LDS image of the halo tile: [voxel (tz, hy, wx)][2 NPL chunks of 16 B = (plane, k-half)], chunk c of a voxel at slot
//   two planes:   c ^ ((wx >> 1) & 3)          three planes:   (c + 3 ((wx >> 1) & 1)) mod 6
// — found by exhaustive search: every ds_read_b128 of a 2 x 16-voxel fragment is bank-conflict free at all 27 tap shifts.
// Weight stage: [plane][64 out][32 B], k-half at slot half ^ ((out >> 3) & 1) (conv3_weight_kernel).
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }
constexpr int HT_T = 4, HT_W = 16, HT_HW = HT_W + 2;
template <int TERMS> struct HaloCfg {
    static constexpr int NPL = s3_planes(TERMS);
    static constexpr int TH = NPL == 2 ? 8 : 4, HH = TH + 2;
    static constexpr int TM = TH / 2;                                  // 32-voxel row tiles per wave
    static constexpr int NCH = 2 * NPL, RB = 16 * NCH;                // chunks / bytes per voxel of one 16-channel slab
    static constexpr int HVOX = (HT_T + 2) * HH * HT_HW;              // 1,080 / 648 voxels
    static constexpr int NPIECE = (HVOX * NCH + 63) / 64;             // 1-KiB DMA pieces of one halo tile: 68 / 61
    static constexpr int HALO_B = NPIECE * 1024;
    static constexpr int WST = 64 * 32 * NPL;                         // live bytes of a weight stage: 4 / 6 KiB
    static constexpr int NWS = 3;                                     // weight ring
    static constexpr int SLAB_B = 4 * 64 * 68 * 4;                    // epilogue slabs (4 waves x 64 rows), overlay everything
    static constexpr int LDS = HALO_B + NWS * WST > SLAB_B ? HALO_B + NWS * WST : SLAB_B;
    static constexpr int VOX = HT_T * TH * HT_W;                      // output voxels per block: 512 / 256
};

// NSLAB: 16-channel slabs of the input.  4 = the 64 -> 64 convolutions (act3 rows of 384 B).  1 (LAT) = the decoder's FIRST convolution
// composed with what precedes it (round 5): its input is upsample(from_lat(z)) = from_lat_w . upsample(z) + from_lat_b — upsampling is
// linear, channel-wise and its weights sum to one — so conv(u) = (conv_w . from_lat_w) * upsample(z) + a bias term, a convolution with Cv = 8
// (<= 16) input channels instead of 64: ONE slab of 27 taps instead of four, a quarter of the MFMAs (the composite weights are built once per
// parameter version by the host; the input image holds upsample(z) in rows of 96 B = [plane][16 ch]).  The from_lat bias reaches an output
// voxel through the taps that fall inside the volume only (the conv zero-pads u, not from_lat's bias): a table per border class, added in
// the epilogue.  Same operator up to fp32 rounding (fewer roundings than the reference's order: u is never rounded).
template <int TERMS, int NSLAB = 4, int OUT = 0>     // TERMS 6: bf16x3; 3: f16x2 (planes h, l of the same buffers; the third plane is neither moved nor read)
__global__ __launch_bounds__(256, 2) void conv3d_k3_bf16x3_kernel(Conv3Args g) {
    static_assert(OUT != 1 || TERMS == 6, "image output: the three-plane mode (an fp16 image needs a bound on the values before they exist)");
    static_assert(OUT >= 0 && OUT <= 3, "0 fp32 NDHWC, 1 operand image, 2 to_img partial sums, 3 pooling partial sums");
    using Cf = HaloCfg<TERMS>;
    constexpr int NPL = Cf::NPL, TH = Cf::TH, HH = Cf::HH, TM = Cf::TM, NCH = Cf::NCH, RB = Cf::RB, NWS = Cf::NWS, WST = Cf::WST;
    constexpr bool F16 = TERMS == 3;
    // NSLAB 0: the latent-composed conv with TWO taps per k-step (lat_ch <= 8): the 32x32x16 MFMA's k-half is a tap, not the upper half of a
    // 16-channel slab that is zero anyway — lanes of k-half 0 read the voxel of tap 2 s, lanes of k-half 1 the voxel of tap 2 s + 1, both
    // its channels 0 .. 7; the weight image holds [tap 2 s: 8 ch | tap 2 s + 1: 8 ch] per stage (the host packs it): 14 steps instead of 27
    constexpr bool LAT = NSLAB <= 1, PK = NSLAB == 0;
    constexpr int NSL = LAT ? 1 : NSLAB, NTAP = PK ? 14 : 27;
    // bytes per voxel / per plane of the input image, and where slab s of it starts.  Voxel-major act3 rows (384 B = [plane][64 ch]) give a
    // slab 32 bytes of each 128-byte plane line: a block touches every line of its tile four times, ~27 steps apart, and the L2 does not
    // hold 64 blocks' tiles that long — FETCH_SIZE of the 64 -> 64 conv at 256 x 256: 9.6 GB per launch for a 1.2 GB image (x 2.5 halo
    // overlap x 4).  SLAB-major (round 5, x_slab_stride != 0): four images of 96 B per voxel ([plane][16 ch], the latent image's form), one per
    // 16-channel slab — a refill streams dense rows of its slab's image.
    const bool xslab = !LAT && g.x_slab_stride != 0;
    const int XROWB = LAT ? L16_ROWB : xslab ? NPL * 32 : A3_ROWB, XPLS = LAT || xslab ? 32 : 128;
    auto slab_off = [&](int slab) -> int64_t { return xslab ? (int64_t)slab * g.x_slab_stride : (int64_t)slab * 32; };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    unsigned char* halo = smem3;
    unsigned char* wring = smem3 + Cf::HALO_B;

    const int nwg = gridDim.x;
    int wg;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = b & 7;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
    }
    // tiles of one sample: nt x nh x nw, w fastest (neighbouring blocks of an XCD share halo voxels in its L2)
    const int nw_t = (g.W + HT_W - 1) / HT_W, nh_t = (g.H + TH - 1) / TH, nt_t = (g.T + HT_T - 1) / HT_T;
    const int per_smp = nw_t * nh_t * nt_t;
    const int smp = wg / per_smp, tile = wg % per_smp;
    const int w0 = (tile % nw_t) * HT_W, h0 = ((tile / nw_t) % nh_t) * TH, t0 = (tile / (nw_t * nh_t)) * HT_T;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = the t-slice this wave owns
    const int l31 = lane & 31, hi = lane >> 5;
    const int Hp = g.H + 2, Wp = g.W + 2;
    const int THW = g.T * g.H * g.W;

    // ---- halo fill.  Three planes (round 5): by ROWS of the tile — a row (tz, hy) is 18 voxels x 6 chunks = 108 consecutive 16-byte chunks of the
    // LDS image = two DMA instructions (64 + 44 lanes) whose lane -> (voxel, chunk) -> source offset map does not depend on the row: two
    // per-lane offsets computed once, a scalar base per row (9 rows per wave, scalar unit), nothing else per refill.  (Rounds 3-4 cut the image into
    // 1-KiB pieces whatever the rows: ~40 vector instructions of index arithmetic per piece, 16 pieces per wave and slab — with the
    // per-step address arithmetic a third of the kernel's issue slots, VALUBusy 29 %.)
    constexpr bool ROWFILL = NPL == 3;
    constexpr int HROWS = (HT_T + 2) * HH, RPW = (HROWS + 3) / 4, ROWCH = HT_HW * NCH;      // 36 rows, 9 per wave, 108 chunks per row
    [[maybe_unused]] int hsrc[2];
    if constexpr (ROWFILL) {
        static_assert(ROWCH > 64 && ROWCH <= 128 && HROWS * ROWCH * 16 <= Cf::HALO_B, "two DMA instructions per halo row");
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            int ci = k2 * 64 + lane;
            ci = ci < ROWCH ? ci : ROWCH - 1;
            const int wx = ci / NCH, pc = ci - wx * NCH;
            int c = pc - 3 * ((wx >> 1) & 1);
            c = c < 0 ? c + 6 : c;
            int ww = w0 + wx;
            ww = ww < Wp ? ww : Wp - 1;             // ragged tiles: clamped to the buffer (those voxels only feed outputs that are never stored)
            hsrc[k2] = ww * XROWB + (c >> 1) * XPLS + (c & 1) * 16;
        }
    }
    constexpr int SCH = HH * HT_HW * NCH, SNI = (SCH + 63) / 64, SPW = (SNI + 3) / 4;      // chunks / DMA instructions of one t-slice, per wave
    [[maybe_unused]] int hsl[SPW];
    if constexpr (!ROWFILL) {
        static_assert((HT_T + 2) * SCH * 16 <= Cf::HALO_B && (SPW - 1) * 4 * 64 + 3 * 64 + 63 < SCH + 64, "t-slice fill: only the last instruction is partial");
#pragma unroll
        for (int k = 0; k < SPW; ++k) {
            int ci = (wave + 4 * k) * 64 + lane;
            ci = ci < SCH ? ci : SCH - 1;
            const int hv = ci / NCH, pc = ci - hv * NCH;
            const int hy = hv / HT_HW, wx = hv - hy * HT_HW;
            const int c = pc ^ ((wx >> 1) & 3);
            int hh = h0 + hy, ww = w0 + wx;        // ragged tiles: clamped to the buffer (those voxels only feed outputs that are never stored)
            hh = hh < Hp ? hh : Hp - 1;
            ww = ww < Wp ? ww : Wp - 1;
            hsl[k] = (hh * Wp + ww) * XROWB + (c >> 1) * XPLS + (c & 1) * 16;
        }
    }
    auto fill_halo = [&](int slab) {
        if constexpr (ROWFILL) {
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                const int r = wave + 4 * k;
                if (r < HROWS) {                    // wave-uniform; the row's base is scalar arithmetic, redone per refill (registers are scarce)
                    const int tz = r / HH, hy = r - tz * HH;
                    int tt = t0 + tz, hh = h0 + hy;
                    tt = tt < g.T + 2 ? tt : g.T + 1;
                    hh = hh < Hp ? hh : Hp - 1;
                    const unsigned char* src = g.X3 + ((((int64_t)smp * (g.T + 2) + tt) * Hp + hh) * Wp) * XROWB + slab_off(slab);
                    __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(src + hsrc[0]), AVD_LDS_PTR(halo + r * (ROWCH * 16)), 16, 0, 0);
                    if (lane < ROWCH - 64)
                        __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(src + hsrc[1]), AVD_LDS_PTR(halo + r * (ROWCH * 16) + 1024), 16, 0, 0);
                }
            }
        } else {
            // two planes: by t-slices of the tile — HH x 18 voxels x 4 chunks = 720 consecutive chunks = 12 DMA instructions, three per wave,
            // whose lane -> (hy, wx, chunk) -> source offset map does not depend on the slice: three per-lane offsets, a scalar base per slice
#pragma unroll
            for (int tz = 0; tz < HT_T + 2; ++tz) {
                int tt = t0 + tz;
                tt = tt < g.T + 2 ? tt : g.T + 1;
                const unsigned char* src = g.X3 + (((int64_t)smp * (g.T + 2) + tt) * Hp * Wp) * XROWB + slab_off(slab);
#pragma unroll
                for (int k = 0; k < SPW; ++k) {
                    const int j = wave + 4 * k;
                    if (j < SNI && (k + 1 < SPW || j * 64 + lane < SCH))
                        __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(src + hsl[k]), AVD_LDS_PTR(halo + (tz * SCH + j * 64) * 16), 16, 0, 0);
                }
            }
        }
    };
    // ---- weight stage kt = 27 slab + tap: NPL pieces of 2 KiB... 1-KiB pieces 2 NPL, dealt over the four waves
    const unsigned char* w3 = g.W3 + (int64_t)smp * g.w3_stride;
    auto fill_w = [&](int kt, int stage) {
        unsigned char* dst = wring + stage * WST;
#pragma unroll
        for (int i = 0; i < (2 * NPL + 3) / 4; ++i) {
            const int pce = wave + 4 * i;
            if (pce < 2 * NPL)
                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(w3 + (int64_t)kt * W3_STAGE + pce * 1024 + lane * 16), AVD_LDS_PTR(dst + pce * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addressing.  A: lane (l31, hi) of row tile i reads voxel (tz = wave + dt, hy = 2 i + (l31 >> 4) + dh, wx = (l31 & 15) + dw),
    // chunk (plane, hi) at its swizzled slot; B: out channel 32 j + l31, k-half hi
    const int wl = l31 & 15, h2 = l31 >> 4;
    const int a_vox0 = (wave * HH + h2) * HT_HW + wl;                 // tap (0,0,0), row tile 0
    const int b_off0 = l31 * 32 + ((hi ^ ((l31 >> 3) & 1)) << 4);      // column tile 0 (tile 1: + 1024; (out >> 3) & 1 is the same)
    struct Frag { bf16x8 a[TM][NPL]; bf16x8 b[2][NPL]; };
    // The 27 taps of a slab are unrolled (round 5): a tap's row shift, its weight stage (27 = 0 mod 3: stage = tap mod 3) and the row tiles'
    // strides are immediates of the ds_reads; what stays in registers is one byte offset per (tap column dw, plane) — the chunk swizzle
    // follows the voxel's wx = wl + dw — and the W fragments' lane offset.
    int aoff[3][NPL];
#pragma unroll
    for (int dw = 0; dw < 3; ++dw) {
        const int wx = wl + dw;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            int pc;
            if constexpr (NPL == 2) pc = (2 * p + hi) ^ ((wx >> 1) & 3);
            else { pc = 2 * p + hi + 3 * ((wx >> 1) & 1); pc = pc >= 6 ? pc - 6 : pc; }
            aoff[dw][p] = (a_vox0 + dw) * RB + pc * 16;
        }
    }
    // packed taps: step S carries taps 2 S (k-half 0) and 2 S + 1 (k-half 1).  The second tap is the next one in the row (types 0, 1), the first
    // of the next row (type 2) or of the next t-slice (type 3): per type and plane one per-lane offset, hi ? (tap B's) : (tap A's), both on
    // chunk (plane, k-half 0) of their voxel; the immediate is tap A's row
    // (type 4: tap 26 has no partner — its k-half 1 meets zero weights, and reads tap 26's own voxel: finite data, never LDS nobody wrote)
    [[maybe_unused]] int pb[5][NPL];
    if constexpr (PK) {
#pragma unroll
        for (int ty = 0; ty < 5; ++ty) {
            const int dwa = ty < 2 ? ty : 2, dwb = ty < 2 ? ty + 1 : ty == 4 ? 2 : 0;
            const int rows = ty == 2 ? 1 : ty == 3 ? HH - 2 : 0;            // tap B's row relative to tap A's
            const int dwl = hi ? dwb : dwa, wx = wl + dwl;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                int pc;
                if constexpr (NPL == 2) pc = (2 * p) ^ ((wx >> 1) & 3);
                else { pc = 2 * p + 3 * ((wx >> 1) & 1); pc = pc >= 6 ? pc - 6 : pc; }
                pb[ty][p] = (a_vox0 + dwl + (hi ? rows * HT_HW : 0)) * RB + pc * 16;
            }
        }
    }
    auto load_frags = [&](Frag& f, auto tap_c) {
        constexpr int TAP = decltype(tap_c)::value;
        const unsigned char* wst = wring + (TAP % NWS) * WST;
        constexpr int TA = PK ? 2 * TAP : TAP, dt = TA / 9, dh = (TA / 3) % 3, dw = TA % 3;
        constexpr int ty = dw < 2 ? dw : TA == 26 ? 4 : (dh < 2 ? 2 : 3);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (PK) f.a[i][p] = *reinterpret_cast<const bf16x8*>(halo + pb[ty][p] + ((dt * HH + dh) * HT_HW + 2 * i * HT_HW) * RB);
                else f.a[i][p] = *reinterpret_cast<const bf16x8*>(halo + aoff[dw][p] + ((dt * HH + dh) * HT_HW + 2 * i * HT_HW) * RB);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) f.b[j][p] = *reinterpret_cast<const bf16x8*>(wst + p * 2048 + j * 1024 + b_off0);
        }
    };
    constexpr int NT = F16 ? 3 : 6;
    constexpr int PA[6] = {F16 ? 0 : 2, F16 ? 1 : 0, F16 ? 0 : 1, 1, 0, 0};      // f16x2: hl, lh, hh; bf16x3: small terms first
    constexpr int PB[6] = {F16 ? 1 : 0, F16 ? 0 : 2, F16 ? 0 : 1, 0, 1, 0};
    auto mma_step = [&](const Frag& f) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma16<F16>(f.a[i][PA[t]], f.b[j][PB[t]], acc[i][j]);
    };

    // prologue: halo of slab 0, weight stages 0 and 1; fragments of step 0
    fill_halo(0);
    fill_w(0, 0);
    fill_w(1, 1);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    // three planes: 64 accumulator registers leave room for the fragments of TWO steps (those of step q+1 are read under the MFMAs
    // of step q); two planes: 128 accumulator registers, one fragment set, read at the top of its step (the co-resident block covers)
    constexpr bool PREFETCH = NPL == 3;
    static_assert(NWS == 3, "27 taps = 0 mod the weight ring: a tap's stage is a constant");
    Frag f[2];
    using Tap0 = std::integral_constant<int, 0>;
    if constexpr (PREFETCH) load_frags(f[0], Tap0{});

    // step q = 27 slab + tap.  Top of step q: weight stage q+1 has landed (issued a step ago) -> barrier -> stage q+2 is issued into the
    // slot stage q-1 vacated (its last read was the prefetch of step q-1's fragments, during step q-2).  The MFMAs of step q run on
    // registers; the fragments of step q+1 are read meanwhile.  On the last tap of a slab the halo tile is refilled for the next slab:
    // every wave has finished reading it (its last reads were the prefetch of THIS step), the refill is issued behind the barrier, the
    // step's MFMAs run while it is in flight, and only then do the waves wait and read the next step's fragments from the new tile.
#pragma unroll 1
    for (int slab_i = 0; slab_i < NSL; ++slab_i) {
        // an opaque copy of the slab index for the address arithmetic: with the 27 steps unrolled, loop strength reduction otherwise keeps
        // one 64-bit induction pointer per DMA site alive across the loop (+ ~110 registers: spills)
        int slab = slab_i;
        asm volatile("" : "+s"(slab));
        const bool more = slab_i + 1 < NSL;                      // wave-uniform
        static_for<NTAP>([&](auto tap_c) {
            constexpr int TAP = decltype(tap_c)::value;
            using Next = std::integral_constant<int, (TAP + 1) % NTAP>;
            __builtin_amdgcn_s_waitcnt(0x0070);                 // vmcnt(0) lgkmcnt(0): this wave's DMA landed, its fragment reads returned
            __builtin_amdgcn_s_barrier();
            if (TAP + 2 < NTAP || more) fill_w(slab * NTAP + TAP + 2, (TAP + 2) % NWS);
            if constexpr (PREFETCH) {
                Frag& cur = f[TAP & 1];
                Frag& nxt = f[(TAP & 1) ^ 1];
                if constexpr (TAP == NTAP - 1) {
                    if (more) {
                        fill_halo(slab + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        mma_step(cur);
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_waitcnt(0x0f70);
                        __builtin_amdgcn_s_barrier();
                        load_frags(f[0], Tap0{});               // (27 is odd: the next slab starts on set 0 again, read after this step's MFMAs)
                    } else {
                        __builtin_amdgcn_sched_barrier(0);
                        mma_step(cur);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    load_frags(nxt, Next{});
                    mma_step(cur);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {       // no prefetch: the step's fragments are read at its top
                load_frags(f[0], tap_c);
                if constexpr (TAP == NTAP - 1) {
                    if (more) {                  // last tap of the slab: once every wave holds its fragments the tile is refilled
                        __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0)
                        __builtin_amdgcn_s_barrier();
                        fill_halo(slab + 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                mma_step(f[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    }
    __syncthreads();      // slabs overlay the halo tile and the weight ring

    // ---- epilogue: 64-row passes through a per-wave LDS slab -> bias + GELU -> NDHWC store + GroupNorm partial statistics ----
    constexpr int CLD = 64 + 4;
    float* slab_f = reinterpret_cast<float*>(smem3) + wave * 64 * CLD;
    const int cr = lane >> 4, cc = (lane & 15) * 4;          // 16 lanes per row (4 channels each), 4 rows per wave instruction
    const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + cc);
    const float ab_inv = F16 ? g.ab_inv * (g.a_inv_dev ? g.a_inv_dev[0] : 1.0f) : 1.0f;
    const int tt = t0 + wave;
    if constexpr (OUT != 0) {
        // 8 lanes per voxel row, 8 channels (one GroupNorm group) each, 8 rows per wave instruction
        const int r8 = lane >> 3, c8 = lane & 7;
        float b8[8];
        *reinterpret_cast<f32x4*>(b8) = *reinterpret_cast<const f32x4*>(g.bias + c8 * 8);
        *reinterpret_cast<f32x4*>(b8 + 4) = *reinterpret_cast<const f32x4*>(g.bias + c8 * 8 + 4);
        [[maybe_unused]] float wg[4][8];
        if constexpr (OUT == 2) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                *reinterpret_cast<f32x4*>(wg[o]) = *reinterpret_cast<const f32x4*>(g.wimg_g + o * VC + c8 * 8);
                *reinterpret_cast<f32x4*>(wg[o] + 4) = *reinterpret_cast<const f32x4*>(g.wimg_g + o * VC + c8 * 8 + 4);
            }
        }
        const float* bt = g.btab ? g.btab + (int64_t)smp * g.btab_stride : nullptr;
#pragma unroll
        for (int ps = 0; ps < TM / 2; ++ps) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) slab_f[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[ps * 2 + i][j][r];
            float s1 = 0.f, s2 = 0.f;
            // OUT = 3 (the encoder's last conv): sums of GELU(y) over this pass's 4 h-rows x 8 w of each of its two 8-wide w blocks (it even /
            // odd) — what AvgPool3d(4, 8, 8) -> to_lat needs of this conv's output (GroupNorm in between is affine per channel, so it
            // commutes with the mean); the fp32 activations are not written
            [[maybe_unused]] float pool[2][8];
            if constexpr (OUT == 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { pool[0][e] = 0.f; pool[1][e] = 0.f; }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = r8 + it * 8;
                const int hh = h0 + 2 * (ps * 2 + (row >> 5)) + ((row >> 4) & 1), ww = w0 + (row & 15);
                float y[8];
                *reinterpret_cast<f32x4*>(y) = *reinterpret_cast<const f32x4*>(slab_f + row * CLD + c8 * 8);
                *reinterpret_cast<f32x4*>(y + 4) = *reinterpret_cast<const f32x4*>(slab_f + row * CLD + c8 * 8 + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = F16 ? y[e] * ab_inv + b8[e] : y[e] + b8[e];
                if (bt) {        // the folded GroupNorm shift (or from_lat's bias) through the taps inside the volume (class 63 = interior)
                    const int cls = (tt >= 1) | ((tt <= g.T - 2) << 1) | ((hh >= 1) << 2) | ((hh <= g.H - 2) << 3) | ((ww >= 1) << 4) | ((ww <= g.W - 2) << 5);
                    const f32x4 t0v = *reinterpret_cast<const f32x4*>(bt + cls * VC + c8 * 8), t1v = *reinterpret_cast<const f32x4*>(bt + cls * VC + c8 * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { y[e] += t0v[e]; y[4 + e] += t1v[e]; }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = gelu_erf(y[e]);
                if (tt < g.T && hh < g.H && ww < g.W) {
                    if constexpr (OUT == 1) {
                        const int64_t pv = (((int64_t)smp * (g.T + 2) + tt + 1) * Hp + hh + 1) * Wp + ww + 1;
                        store_act3<false>(g.X3out, pv, c8, y, 0.f, g.out_slab_stride);
                    } else if constexpr (OUT == 3) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) pool[it & 1][e] += y[e];
                    } else {
                        const int64_t v = ((int64_t)tt * g.H + hh) * g.W + ww;
                        f32x4 pp;
#pragma unroll
                        for (int o = 0; o < 4; ++o) {
                            float a = 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e) a = fmaf(y[e], wg[o][e], a);
                            pp[o] = a;
                        }
                        *reinterpret_cast<f32x4*>(g.P + (((int64_t)smp * THW + v) * VG + c8) * 4) = pp;
                    }
                    s1 += ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
                    s2 += ((y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3])) + ((y[4] * y[4] + y[5] * y[5]) + (y[6] * y[6] + y[7] * y[7]));
                }
            }
            s1 += __shfl_xor(s1, 8, 64);  s2 += __shfl_xor(s2, 8, 64);
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if constexpr (OUT == 3) {
#pragma unroll
                for (int wb = 0; wb < 2; ++wb)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float a = pool[wb][e];
                        a += __shfl_xor(a, 8, 64);
                        a += __shfl_xor(a, 16, 64);
                        a += __shfl_xor(a, 32, 64);
                        pool[wb][e] = a;
                    }
            }
            if (lane < VG) {
                const int64_t e = ((int64_t)tile * 4 + wave) * (TM / 2) + ps;
                float* pp = g.part + (((int64_t)smp * g.tiles * 2 + e) * VG + lane) * 2;
                pp[0] = s1;
                pp[1] = s2;
                if constexpr (OUT == 3) {       // pooling partials: [sample][entry][w block][64 channels]
                    float* q = g.P + (((int64_t)smp * g.tiles * 2 + e) * 2) * VC + lane * 8;
#pragma unroll
                    for (int wb = 0; wb < 2; ++wb) {
                        *reinterpret_cast<f32x4*>(q + wb * VC) = f32x4{pool[wb][0], pool[wb][1], pool[wb][2], pool[wb][3]};
                        *reinterpret_cast<f32x4*>(q + wb * VC + 4) = f32x4{pool[wb][4], pool[wb][5], pool[wb][6], pool[wb][7]};
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ps = 0; ps < TM / 2; ++ps) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab_f[(i * 32 + mfma32_row(r, hi)) * CLD + j * 32 + l31] = acc[ps * 2 + i][j][r];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = cr + it * 4;                      // slab row: row tile ps*2 + row/32, voxel (h2, wl) = ((row%32)/16, row%16)
            const int hh = h0 + 2 * (ps * 2 + (row >> 5)) + ((row >> 4) & 1), ww = w0 + (row & 15);
            f32x4 y = *reinterpret_cast<const f32x4*>(slab_f + row * CLD + cc);
            if constexpr (F16) y *= ab_inv;
            y += bv;
            if constexpr (LAT) {      // from_lat's bias through the taps inside the volume (class 63 = interior)
                const int cls = (tt >= 1) | ((tt <= g.T - 2) << 1) | ((hh >= 1) << 2) | ((hh <= g.H - 2) << 3) | ((ww >= 1) << 4) | ((ww <= g.W - 2) << 5);
                y += *reinterpret_cast<const f32x4*>(g.btab + cls * VC + cc);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = gelu_erf(y[e]);
            if (tt < g.T && hh < g.H && ww < g.W) {
                const int64_t v = ((int64_t)tt * g.H + hh) * g.W + ww;
                *reinterpret_cast<f32x4*>(g.Y + ((int64_t)smp * THW + v) * VC + cc) = y;
                s1 += (y[0] + y[1]) + (y[2] + y[3]);
                s2 += (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
            }
        }
        // lanes 2k, 2k+1 of a row hold the 8 channels of group k; rows: lane bits 4, 5
        s1 += __shfl_xor(s1, 1, 64);  s2 += __shfl_xor(s2, 1, 64);
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if ((lane & 0x31) == 0) {
            // one partials entry per 64 stored-or-not voxels: entry index = (block x 4 waves + wave) x (TM / 2) + ps
            const int64_t e = ((int64_t)tile * 4 + wave) * (TM / 2) + ps;
            float* pp = g.part + (((int64_t)smp * g.tiles * 2 + e) * VG + (lane >> 1)) * 2;
            pp[0] = s1;
            pp[1] = s2;
        }
    }
}

// launch of one 64 -> 64 convolution on the 16-bit matrix pipe (terms 0 / 6: bf16x3, 3: f16x2)
static LdsAttr g_conv3_attr[2];      // dynamic-LDS limit of conv3d_k3_bf16x3_kernel<6> / <3> (per device)
// blocks per sample of the halo-tile kernel, and the number of 128-voxel "tiles" its GroupNorm partials amount to (one partials
// entry per 64 voxels of every block, whether the tile is ragged or not): what gn_finalize_kernel sums over for this conv
template <int TERMS> static int conv3_blocks(int T, int H, int W) {
    using Cf = HaloCfg<TERMS>;
    return ((T + HT_T - 1) / HT_T) * ((H + Cf::TH - 1) / Cf::TH) * ((W + HT_W - 1) / HT_W);
}
static int conv3_tiles(int terms, int T, int H, int W) {
    return terms == 3 ? conv3_blocks<3>(T, H, W) * (HaloCfg<3>::VOX / 128) : conv3_blocks<6>(T, H, W) * (HaloCfg<6>::VOX / 128);
}
int g_vae_lat = getenv("AVD_VAE_LAT") ? atoi(getenv("AVD_VAE_LAT")) : 1;      // avd_tune_set "vae_lat": 0 = from_lat -> upsample -> 64-channel first conv
static LdsAttr g_conv3_lat_attr[2];
// avd_tune_set "vae_fold" (AVD_VAE_FOLD): 1 (default) = the three-plane decoder with two conv blocks and a latent-composed first conv writes
// conv 0's output straight into conv 1's operand image, folds the GroupNorm between them into conv 1's per-sample weights, and finishes
// to_img from per-group partial sums of conv 1's epilogue; 0 = fp32 activations between the kernels (rounds 1-4)
int g_vae_fold = getenv("AVD_VAE_FOLD") ? atoi(getenv("AVD_VAE_FOLD")) : 1;
static LdsAttr g_conv3_fold_attr[3];
template <int TERMS, int NSLAB, int OUT>
static int conv3_launch_as(const Conv3Args& a3, int B, double flops, hipStream_t st, LdsAttr& attr, const char* what) {
    auto kern = conv3d_k3_bf16x3_kernel<TERMS, NSLAB, OUT>;
    if (int rc = attr.ensure(reinterpret_cast<const void*>(kern), HaloCfg<TERMS>::LDS, what)) return rc;
    static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<%d, %d, %d>", TERMS, NSLAB, OUT);
    ProfScope prof(tag, flops, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * conv3_blocks<TERMS>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<TERMS>::LDS, st, a3);
    AVD_CHECK_LAUNCH(what);
    return AVD_OK;
}
static LdsAttr g_conv3_pk_attr[3];
static int conv3_launch(Conv3Args a3, int terms, int B, double flops, hipStream_t st, bool lat = false, int out_mode = 0, bool packed = false) {
    a3.tiles = conv3_tiles(terms, a3.T, a3.H, a3.W);
    if (lat && packed) {       // two taps per k-step (conv3d_k3_bf16x3_kernel<.., 0, ..>): 14 / 27 of the MFMA work
        AVD_REQUIRE((a3.btab || out_mode == 1) && out_mode != 2 && (out_mode == 0 || (terms != 3 && a3.X3out)), AVD_EINVAL,
                    "conv3d (latent-composed, packed taps): bad arguments");
        if (terms == 3) return conv3_launch_as<3, 0, 0>(a3, B, flops * 14.0 / 27.0, st, g_conv3_pk_attr[0], "conv3d f16x2 (latent, packed taps)");
        if (out_mode == 1) return conv3_launch_as<6, 0, 1>(a3, B, flops * 14.0 / 27.0, st, g_conv3_pk_attr[1], "conv3d bf16x3 (latent, packed taps, image out)");
        return conv3_launch_as<6, 0, 0>(a3, B, flops * 14.0 / 27.0, st, g_conv3_pk_attr[2], "conv3d bf16x3 (latent, packed taps)");
    }
    if (out_mode == 3) {       // the encoder's last conv: pooling partial sums instead of fp32 activations
        AVD_REQUIRE(!lat && a3.P, AVD_EINVAL, "conv3d (pooling partials): bad arguments");
        static LdsAttr attr[2];
        if (terms == 3) return conv3_launch_as<3, 4, 3>(a3, B, flops, st, attr[0], "conv3d f16x2 (pooling partials out)");
        return conv3_launch_as<6, 4, 3>(a3, B, flops, st, attr[1], "conv3d bf16x3 (pooling partials out)");
    }
    if (out_mode != 0) {
        AVD_REQUIRE((out_mode == 1 && terms != 3 && lat && a3.X3out) || (out_mode == 2 && !lat && a3.P && a3.wimg_g), AVD_EINVAL, "conv3d (folded route): bad arguments");
        if (out_mode == 2 && terms == 3) {
            if (int rc = g_conv3_fold_attr[2].ensure(reinterpret_cast<const void*>(conv3d_k3_bf16x3_kernel<3, 4, 2>), HaloCfg<3>::LDS, "conv3d f16x2 (to_img partials out)")) return rc;
            static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<3, 4, 2>");
            ProfScope prof(tag, flops, st);
            hipLaunchKernelGGL((conv3d_k3_bf16x3_kernel<3, 4, 2>), dim3((unsigned)(B * conv3_blocks<3>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<3>::LDS, st, a3);
        } else if (out_mode == 1) {
            if (int rc = g_conv3_fold_attr[0].ensure(reinterpret_cast<const void*>(conv3d_k3_bf16x3_kernel<6, 1, 1>), HaloCfg<6>::LDS, "conv3d bf16x3 (latent, image out)")) return rc;
            static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<6, 1, 1>");
            ProfScope prof(tag, flops, st);
            hipLaunchKernelGGL((conv3d_k3_bf16x3_kernel<6, 1, 1>), dim3((unsigned)(B * conv3_blocks<6>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<6>::LDS, st, a3);
        } else {
            if (int rc = g_conv3_fold_attr[1].ensure(reinterpret_cast<const void*>(conv3d_k3_bf16x3_kernel<6, 4, 2>), HaloCfg<6>::LDS, "conv3d bf16x3 (to_img partials out)")) return rc;
            static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<6, 4, 2>");
            ProfScope prof(tag, flops, st);
            hipLaunchKernelGGL((conv3d_k3_bf16x3_kernel<6, 4, 2>), dim3((unsigned)(B * conv3_blocks<6>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<6>::LDS, st, a3);
        }
        AVD_CHECK_LAUNCH("conv3d (folded route)");
        return AVD_OK;
    }
    if (lat) {
        AVD_REQUIRE(a3.btab, AVD_EINVAL, "conv3d (latent-composed): null bias table");
        if (terms == 3) {
            if (int rc = g_conv3_lat_attr[1].ensure(reinterpret_cast<const void*>(conv3d_k3_bf16x3_kernel<3, 1>), HaloCfg<3>::LDS, "conv3d f16x2 (latent)")) return rc;
            static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<3, 1, 0>");
            ProfScope prof(tag, flops, st);
            hipLaunchKernelGGL((conv3d_k3_bf16x3_kernel<3, 1>), dim3((unsigned)(B * conv3_blocks<3>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<3>::LDS, st, a3);
        } else {
            if (int rc = g_conv3_lat_attr[0].ensure(reinterpret_cast<const void*>(conv3d_k3_bf16x3_kernel<6, 1>), HaloCfg<6>::LDS, "conv3d bf16x3 (latent)")) return rc;
            static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<6, 1, 0>");
            ProfScope prof(tag, flops, st);
            hipLaunchKernelGGL((conv3d_k3_bf16x3_kernel<6, 1>), dim3((unsigned)(B * conv3_blocks<6>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<6>::LDS, st, a3);
        }
        AVD_CHECK_LAUNCH("conv3d (latent-composed)");
        return AVD_OK;
    }
    if (terms == 3) {
        if (int rc = g_conv3_attr[1].ensure(reinterpret_cast<const void*>(conv3d_k3_bf16x3_kernel<3, 4>), HaloCfg<3>::LDS, "conv3d f16x2")) return rc;
        static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<3, 4, 0>");
        ProfScope prof(tag, flops, st);
        hipLaunchKernelGGL((conv3d_k3_bf16x3_kernel<3, 4>), dim3((unsigned)(B * conv3_blocks<3>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<3>::LDS, st, a3);
    } else {
        if (int rc = g_conv3_attr[0].ensure(reinterpret_cast<const void*>(conv3d_k3_bf16x3_kernel<6, 4>), HaloCfg<6>::LDS, "conv3d bf16x3")) return rc;
        static const int tag = prof_tag_id("conv3d_k3_bf16x3_kernel<6, 4, 0>");
        ProfScope prof(tag, flops, st);
        hipLaunchKernelGGL((conv3d_k3_bf16x3_kernel<6, 4>), dim3((unsigned)(B * conv3_blocks<6>(a3.T, a3.H, a3.W))), dim3(256), HaloCfg<6>::LDS, st, a3);
    }
    AVD_CHECK_LAUNCH("conv3d (split operands)");
    return AVD_OK;
}
static int check_conv_terms(int terms, const float* w_scale, const float* a_scale, int n_blocks, int first, bool first_runtime) {
    AVD_REQUIRE(terms == 0 || terms == 6 || terms == 3, AVD_EINVAL, "vae: conv_terms must be 0 / 6 (bf16x3) or 3 (f16x2), got %d", terms);
    if (terms != 3) return AVD_OK;
    AVD_REQUIRE(w_scale && a_scale, AVD_EINVAL, "vae: conv_terms 3 (f16x2) needs conv_w_scale and conv_a_scale");
    for (int b = first; b < n_blocks; ++b) {
        AVD_REQUIRE(w_scale[b] > 0.f && w_scale[b] < __builtin_inff(), AVD_EINVAL, "vae: conv_w_scale[%d] must be positive and finite", b);
        if (!(first_runtime && b == first))
            AVD_REQUIRE(a_scale[b] > 0.f && a_scale[b] < __builtin_inff(), AVD_EINVAL, "vae: conv_a_scale[%d] must be positive and finite", b);
    }
    return AVD_OK;
}

static inline int64_t a256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// upsample(z) -> the latent-composed conv's input image: from a channel-last copy of the latent when lat_ch <= 8 (zt = scratch of B x vol x 8 floats)
static int upsample_lat16(const float* z, float* zt, unsigned char* X16, int B, int Cv, int Tp, int Hp, int Wp, int T, int H, int W, bool f16,
                          const float* sc_dev, hipStream_t st) {
    const int64_t nvox = (int64_t)B * T * H * W;
    const float fst = (float)Tp / (float)T, fsh = (float)Hp / (float)H, fsw = (float)Wp / (float)W;
    if (Cv <= 8 && zt) {
        const int vol = Tp * Hp * Wp;
        const int64_t total = (int64_t)B * vol;
        hipLaunchKernelGGL(lat_cl8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, zt, Cv, vol, total);
        AVD_CHECK_LAUNCH("lat_cl8");
        static const int tag = prof_tag_id("upsample_lat8_kernel");
        ProfScope prof(tag, (double)nvox * L16_ROWB, st);
        if (f16) hipLaunchKernelGGL(upsample_lat8_kernel<true>, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, zt, X16, Tp, Hp, Wp, T, H, W, fst, fsh, fsw, nvox, sc_dev);
        else hipLaunchKernelGGL(upsample_lat8_kernel<false>, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, zt, X16, Tp, Hp, Wp, T, H, W, fst, fsh, fsw, nvox, nullptr);
        AVD_CHECK_LAUNCH("upsample_lat8");
        return AVD_OK;
    }
    static const int tag = prof_tag_id("upsample_lat16_kernel");
    ProfScope prof(tag, (double)nvox * L16_ROWB, st);
    if (f16) hipLaunchKernelGGL(upsample_lat16_kernel<true>, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, z, X16, Cv, Tp, Hp, Wp, T, H, W, fst, fsh, fsw, nvox, sc_dev);
    else hipLaunchKernelGGL(upsample_lat16_kernel<false>, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, z, X16, Cv, Tp, Hp, Wp, T, H, W, fst, fsh, fsw, nvox, nullptr);
    AVD_CHECK_LAUNCH("upsample_lat16");
    return AVD_OK;
}



struct VaePlan {
    int T, H, W, tiles;
    bool fold_img;
    int64_t THW, pad_b, y_b, hlow_b, part_b, part_n, fold_b, stats_b, total;
};

static int vae_plan(const avd_vae_decode_desc* d, VaePlan& p) {
    AVD_REQUIRE(d, AVD_EINVAL, "vae_decode: null descriptor");
    AVD_REQUIRE(d->B > 0 && d->Cv > 0 && d->Tp > 0 && d->Hp > 0 && d->Wp > 0, AVD_EINVAL, "vae_decode: bad latent dims");
    AVD_REQUIRE(d->base == VC, AVD_EUNSUPPORTED, "vae_decode: decoder width %d unsupported (kernels are built for 64)", d->base);
    AVD_REQUIRE(d->n_blocks >= 1 && d->n_blocks <= 8, AVD_EUNSUPPORTED, "vae_decode: 1..8 conv blocks supported");
    AVD_REQUIRE(d->out_ch >= 1 && d->out_ch <= 4, AVD_EUNSUPPORTED, "vae_decode: 1..4 output channels supported");
    AVD_REQUIRE(d->T > 0 && d->H > 0 && d->W > 0, AVD_EINVAL, "vae_decode: bad output size");
    p.T = d->T; p.H = d->H; p.W = d->W;
    p.THW = (int64_t)d->T * d->H * d->W;
    AVD_REQUIRE(p.THW * VC < (1ll << 31) && (int64_t)(d->T + 2) * (d->H + 2) * (d->W + 2) * VC < (1ll << 31), AVD_EUNSUPPORTED,
                "vae_decode: one sample's activation exceeds 2^31 elements");
    p.tiles = (int)((p.THW + VBM - 1) / VBM);
    p.pad_b = a256((int64_t)d->B * (d->T + 2) * (d->H + 2) * (d->W + 2) * (d->conv_w3 ? A3_ROWB : VC * 4));
    p.y_b = a256((int64_t)d->B * p.THW * VC * 4);
    p.hlow_b = a256((int64_t)d->B * d->Tp * d->Hp * d->Wp * VC * 4);
    {
        const int t3 = conv3_tiles(3, d->T, d->H, d->W), t6 = conv3_tiles(6, d->T, d->H, d->W);
        const int tmax = p.tiles > t3 ? (p.tiles > t6 ? p.tiles : t6) : (t3 > t6 ? t3 : t6);
        p.part_n = (int64_t)d->B * tmax * 2 * VG * 2;
        p.part_b = a256(p.part_n * 4 + gn_fin_bytes(d->B));       // + gn_finalize's per-chunk fp64 sums
    }
    p.stats_b = a256((int64_t)d->B * VG * 2 * 4) + 256;       // + the scale slot of the f16x2 decoder (4 floats)
    // folded route (three planes, two blocks): per-sample weight image + bias table of conv 1, to_img constants, to_img_w . gamma
    // (to_img constants and to_img_w . gamma: every split-operand decoder; the per-sample image and table: three planes, two blocks)
    p.fold_img = d->conv_w3 && d->conv_terms != 3 && d->n_blocks == 2;
    p.fold_b = d->conv_w3 ? a256((int64_t)d->B * 64 + 4 * VC * 4 + (p.fold_img ? (int64_t)d->B * (W3_BYTES + 64 * VC * 4) : 0)) : 0;
    p.total = p.pad_b + p.y_b + p.hlow_b + p.part_b + p.fold_b + p.stats_b;
    return AVD_OK;
}

}  // namespace avd

using namespace avd;

extern "C" int64_t avd_conv3_weight_bytes(void) { return W3_BYTES; }
extern "C" int avd_conv3_weight_f32(const float* w_tap_major, void* img, avd_stream_t stream) {
    AVD_REQUIRE(w_tap_major && img && aligned16(w_tap_major) && aligned16(img), AVD_EINVAL, "conv3_weight: bad pointer");
    hipLaunchKernelGGL(conv3_weight_kernel<false>, dim3(54), dim3(256), 0, static_cast<hipStream_t>(stream), w_tap_major,
                       static_cast<unsigned char*>(img), 0.f);
    AVD_CHECK_LAUNCH("conv3_weight");
    return AVD_OK;
}
extern "C" int avd_conv3_weight_f16x2_f32(const float* w_tap_major, void* img, float scale, avd_stream_t stream) {
    AVD_REQUIRE(w_tap_major && img && aligned16(w_tap_major) && aligned16(img), AVD_EINVAL, "conv3_weight: bad pointer");
    AVD_REQUIRE(scale > 0.f && scale < __builtin_inff(), AVD_EINVAL, "conv3_weight: the f16x2 image scale must be positive and finite");
    hipLaunchKernelGGL(conv3_weight_kernel<true>, dim3(54), dim3(256), 0, static_cast<hipStream_t>(stream), w_tap_major,
                       static_cast<unsigned char*>(img), scale);
    AVD_CHECK_LAUNCH("conv3_weight");
    return AVD_OK;
}

extern "C" int64_t avd_vae_decode_workspace_bytes(const avd_vae_decode_desc* d) {
    VaePlan p;
    if (vae_plan(d, p)) return -1;
    return p.total;
}

extern "C" int avd_vae_decode_f32(const avd_vae_decode_desc* d, const float* z, float* out, void* workspace,
                                  int64_t workspace_bytes, avd_stream_t stream) {
    VaePlan p;
    if (int rc = vae_plan(d, p)) return rc;
    AVD_REQUIRE(z && out && d->from_lat_w && d->from_lat_b && d->conv_w && d->conv_b && d->gn_w && d->gn_b && d->to_img_w &&
                d->to_img_b, AVD_EINVAL, "vae_decode: null pointer");
    AVD_REQUIRE(workspace && workspace_bytes >= p.total, AVD_EWORKSPACE, "vae_decode: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)p.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(workspace);
    float* Xp = reinterpret_cast<float*>(w);
    float* Y = reinterpret_cast<float*>(w + p.pad_b);
    float* hlow = reinterpret_cast<float*>(w + p.pad_b + p.y_b);
    float* part = reinterpret_cast<float*>(w + p.pad_b + p.y_b + p.hlow_b);
    double* fin = reinterpret_cast<double*>(part + p.part_n);
    char* foldw = w + p.pad_b + p.y_b + p.hlow_b + p.part_b;
    float* stats = reinterpret_cast<float*>(foldw + p.fold_b);
    const int B = d->B;

    const bool s3 = d->conv_w3 != nullptr;                  // split-operand convolutions: Xp is the act3 buffer (384 B per voxel)
    const bool h2 = s3 && d->conv_terms == 3;               // f16x2: two fp16 planes of that buffer, scaled
    unsigned char* X3 = reinterpret_cast<unsigned char*>(w);
    float* scale_ws = reinterpret_cast<float*>(w + p.total - 256);     // {max |from_lat(z)|, -, s, 1 / s} of the f16x2 decoder's first image
    if (s3) {
        for (int blk = 0; blk < d->n_blocks; ++blk) AVD_REQUIRE(d->conv_w3[blk], AVD_EINVAL, "vae_decode: null conv_w3[%d]", blk);
        if (int rc = check_conv_terms(d->conv_terms, d->conv_w_scale, d->conv_a_scale, d->n_blocks, 0, true)) return rc;
    }
    // latent-composed first convolution (conv3d_k3_bf16x3_kernel<.., 1>): its input image is upsample(z), 96 B per voxel
    const bool lat = s3 && d->conv0_lat_w3 != nullptr && g_vae_lat;
    const int64_t padvox = (int64_t)B * (p.T + 2) * (p.H + 2) * (p.W + 2);
    const int act_rowb = d->conv_w3 && d->conv_terms == 3 ? 64 : L16_ROWB;      // bytes per voxel of one slab image: planes x 32
    const int64_t act_slab = padvox * act_rowb;       // bytes between the four slab images of a split-operand conv's input (slab-major act3)
    const int64_t nvox = (int64_t)B * p.THW;
    float* consts = reinterpret_cast<float*>(foldw);            // {rstd[8], K[4]} per sample (16 floats each), then to_img_w . gamma [4][64]
    float* wg = consts + (int64_t)B * 16;
    float* Pp = Y;                                              // the last conv's to_img partial sums live where the fp32 activations do otherwise
    // to_img from the last conv's per-group partial sums (every split-operand decoder; "vae_fold" 0 restores gn_apply_toimg)
    auto toimg_from_p = [&](int blk) -> int {
        hipLaunchKernelGGL(toimg_consts_kernel, dim3(B), dim3(64), 0, st, stats, d->gn_w[blk], d->gn_b[blk], d->to_img_w, d->to_img_b, consts, d->out_ch);
        AVD_CHECK_LAUNCH("toimg_consts");
        static const int tag = prof_tag_id("toimg_from_p_kernel");
        ProfScope prof(tag, 4.0 * ((double)nvox * VG * 4 + (double)nvox * d->out_ch), st);
        hipLaunchKernelGGL(toimg_from_p_kernel, dim3((unsigned)((nvox + TOIMG_P_VOX - 1) / TOIMG_P_VOX)), dim3(256), 0, st, Pp, consts, out, (int)p.THW,
                           d->out_ch, d->out_tanh, nvox);
        AVD_CHECK_LAUNCH("toimg_from_p");
        return AVD_OK;
    };
    if (lat && !h2 && g_vae_fold && p.fold_img && d->conv0_lat_btab && d->Cv <= (d->conv0_lat_packed ? 8 : 16) && padvox * L16_ROWB <= p.y_b) {
        // ---- folded route: upsample(z) -> L | conv 0 (L -> act3 image of GELU) | stats | conv 1 with GN 0 folded in (-> to_img partials) |
        // stats | to_img.  L and the partials P share the region the fp32 activations Y have on the other routes.
        unsigned char* Lm = reinterpret_cast<unsigned char*>(Y);
        unsigned char* wimg = reinterpret_cast<unsigned char*>(wg + 4 * VC);
        float* btab1 = reinterpret_cast<float*>(wimg + (int64_t)B * W3_BYTES);
        if (int rc = zero_halo(reinterpret_cast<float*>(Lm), B, p.T, p.H, p.W, L16_ROWB, st)) return rc;
        if (int rc = upsample_lat16(z, hlow, Lm, B, d->Cv, d->Tp, d->Hp, d->Wp, p.T, p.H, p.W, false, nullptr, st)) return rc;
        // conv 1's operand image is SLAB-major (four 96-byte-per-voxel images, one per 16 channels: conv3d_k3_bf16x3_kernel) = 4 B "samples" of halo
        const int64_t slab_stride = act_slab;
        if (int rc = zero_halo(Xp, 4 * B, p.T, p.H, p.W, L16_ROWB, st)) return rc;
        const int gn_tiles = conv3_tiles(6, p.T, p.H, p.W);
        {
            Conv3Args a3{Lm, static_cast<const unsigned char*>(d->conv0_lat_w3), d->conv_b[0], nullptr, part, p.T, p.H, p.W, p.tiles, 1.f, nullptr,
                         d->conv0_lat_btab, X3, nullptr, nullptr, 0, 0, 0, slab_stride};
            if (int rc = conv3_launch(a3, 6, B, 2.0 * (double)B * p.THW * VC * 27.0 * 16, st, true, 1, d->conv0_lat_packed != 0)) return rc;
        }
        if (int rc = gn_finalize(part, fin, stats, B, gn_tiles, (double)p.THW * (VC / VG), d->gn_eps, st)) return rc;
        {
            static const int tag = prof_tag_id("conv3_weight_gn_kernel");
            ProfScope prof(tag, (double)B * (W3_BYTES + 27.0 * VC * VC * 4), st);
            hipLaunchKernelGGL(conv3_weight_gn_kernel, dim3(54, B), dim3(256), 0, st, d->conv_w[1], stats, d->gn_w[0], wimg);
            AVD_CHECK_LAUNCH("conv3_weight_gn");
            hipLaunchKernelGGL(conv3_gn_btab_kernel, dim3(B, VC / 8), dim3(256), 0, st, d->conv_w[1], stats, d->gn_w[0], d->gn_b[0], btab1);
            AVD_CHECK_LAUNCH("conv3_gn_btab");
            hipLaunchKernelGGL(toimg_wg_kernel, dim3(1), dim3(256), 0, st, d->to_img_w, d->gn_w[1], wg, d->out_ch);
            AVD_CHECK_LAUNCH("toimg_wg");
        }
        {
            Conv3Args a3{X3, wimg, d->conv_b[1], nullptr, part, p.T, p.H, p.W, p.tiles, 1.f, nullptr, btab1, nullptr, Pp, wg, W3_BYTES, 64 * VC, slab_stride, 0};
            if (int rc = conv3_launch(a3, 6, B, 2.0 * (double)B * p.THW * VC * 27.0 * VC, st, false, 2)) return rc;
        }
        if (int rc = gn_finalize(part, fin, stats, B, gn_tiles, (double)p.THW * (VC / VG), d->gn_eps, st)) return rc;
        return toimg_from_p(1);
    }
    if (lat) {
        AVD_REQUIRE(d->conv0_lat_btab && d->Cv <= 16, AVD_EINVAL, "vae_decode: the latent-composed first conv needs its bias table and Cv <= 16");
        AVD_REQUIRE(!d->conv0_lat_packed || d->Cv <= 8, AVD_EINVAL, "vae_decode: conv0_lat_packed needs Cv <= 8");
        AVD_REQUIRE(!h2 || (d->conv0_lat_w_scale > 0.f && d->conv0_lat_w_scale < __builtin_inff()), AVD_EINVAL,
                    "vae_decode: conv0_lat_w_scale must be positive and finite");
        if (int rc = zero_halo(Xp, B, p.T, p.H, p.W, L16_ROWB, st)) return rc;
        if (h2) {   // |upsample(z)| <= max |z| (a convex combination): the image scale is derived from the data on the device
            if (int rc = weight_bounds_f32(z, (int64_t)B * d->Cv, d->Tp * d->Hp * d->Wp, scale_ws, st)) return rc;
            hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, st, scale_ws);
            AVD_CHECK_LAUNCH("pow2_scale");
        }
        if (int rc = upsample_lat16(z, hlow, X3, B, d->Cv, d->Tp, d->Hp, d->Wp, p.T, p.H, p.W, h2, scale_ws + 2, st)) return rc;
    } else {
    // zero halo (interiors are overwritten below by the upsample / GroupNorm-apply passes, the halo stays zero for every conv); the operand
    // image of the split-operand convs is slab-major: 4 B "samples" of 96-byte rows
    if (int rc = s3 ? zero_halo(Xp, 4 * B, p.T, p.H, p.W, act_rowb, st) : zero_halo(Xp, B, p.T, p.H, p.W, VC * 4, st)) return rc;
    {   // from_lat on the latent grid, then trilinear upsample into the padded conv input
        const int vol = d->Tp * d->Hp * d->Wp;
        const int64_t total = (int64_t)B * vol * VC;
        static const int tag = prof_tag_id("fromlat_kernel");
        ProfScope prof(tag, 4.0 * ((double)B * vol * d->Cv + (double)total), st);
        hipLaunchKernelGGL(fromlat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, z, d->from_lat_w,
                           d->from_lat_b, hlow, d->Cv, vol, total);
        AVD_CHECK_LAUNCH("fromlat");
    }
    if (s3) {
        if (h2) {
            // The first conv's input is a trilinear interpolation (a convex combination) of from_lat(z): its magnitude is bounded by
            // max |from_lat(z)|, which depends on the data — the power-of-two image scale is derived from it on the device
            if (int rc = weight_bounds_f32(hlow, (int64_t)B * d->Tp * d->Hp * d->Wp, VC, scale_ws, st)) return rc;
            hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, st, scale_ws);
            AVD_CHECK_LAUNCH("pow2_scale");
        }
        const int64_t total8 = (int64_t)B * p.THW * 8;
        static const int tag = prof_tag_id("upsample_pad3_kernel");
        ProfScope prof(tag, 6.0 * (double)B * p.THW * VC, st);
        if (h2)
            hipLaunchKernelGGL(upsample_pad3_kernel<true>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, hlow, X3, d->Tp, d->Hp,
                               d->Wp, p.T, p.H, p.W, (float)d->Tp / (float)p.T, (float)d->Hp / (float)p.H, (float)d->Wp / (float)p.W, total8,
                               scale_ws + 2, act_slab);
        else
            hipLaunchKernelGGL(upsample_pad3_kernel<false>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, hlow, X3, d->Tp, d->Hp,
                               d->Wp, p.T, p.H, p.W, (float)d->Tp / (float)p.T, (float)d->Hp / (float)p.H, (float)d->Wp / (float)p.W, total8,
                               nullptr, act_slab);
        AVD_CHECK_LAUNCH("upsample_pad3");
    } else {
        const int64_t total4 = (int64_t)B * p.THW * (VC / 4);
        static const int tag = prof_tag_id("upsample_pad_kernel");
        ProfScope prof(tag, 4.0 * (double)B * p.THW * VC, st);
        hipLaunchKernelGGL(upsample_pad_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, hlow, Xp, d->Tp,
                           d->Hp, d->Wp, p.T, p.H, p.W, (float)d->Tp / (float)p.T, (float)d->Hp / (float)p.H,
                           (float)d->Wp / (float)p.W, total4);
        AVD_CHECK_LAUNCH("upsample_pad");
    }
    }       // (!lat)
    constexpr int stage_lds = 2 * (VBM + VC) * VBK * 4, epi_lds = 4 * 64 * 36 * 4;
    constexpr int lds = stage_lds > epi_lds ? stage_lds : epi_lds;
    for (int blk = 0; blk < d->n_blocks; ++blk) {
        ConvArgs a{Xp, d->conv_w[blk], d->conv_b[blk], Y, part, p.T, p.H, p.W, p.tiles};
        int gn_tiles = p.tiles;           // partials entries / 2 the conv of this block writes (the halo-tile kernel has its own tiling)
        if (s3) {
            const bool lat0 = lat && blk == 0;
            Conv3Args a3{X3, static_cast<const unsigned char*>(lat0 ? d->conv0_lat_w3 : d->conv_w3[blk]), d->conv_b[blk], Y, part, p.T, p.H, p.W, p.tiles,
                         1.f, nullptr, lat0 ? d->conv0_lat_btab : nullptr};
            if (!lat0) a3.x_slab_stride = act_slab;
            if (h2) {
                a3.ab_inv = blk == 0 ? 1.0f / (lat0 ? d->conv0_lat_w_scale : d->conv_w_scale[0]) : 1.0f / (d->conv_w_scale[blk] * d->conv_a_scale[blk]);
                a3.a_inv_dev = blk == 0 ? scale_ws + 3 : nullptr;
            }
            // last conv: to_img's per-group partial sums instead of fp32 activations (toimg_from_p below)
            const bool p_out = g_vae_fold && blk + 1 == d->n_blocks && !lat0;
            if (p_out) {
                hipLaunchKernelGGL(toimg_wg_kernel, dim3(1), dim3(256), 0, st, d->to_img_w, d->gn_w[blk], wg, d->out_ch);
                AVD_CHECK_LAUNCH("toimg_wg");
                a3.Y = nullptr;
                a3.P = Pp;
                a3.wimg_g = wg;
            }
            if (int rc = conv3_launch(a3, h2 ? 3 : 6, B, 2.0 * (double)B * p.THW * VC * 27.0 * (lat0 ? 16 : VC), st, lat0, p_out ? 2 : 0,
                                      lat0 && d->conv0_lat_packed != 0)) return rc;
            gn_tiles = conv3_tiles(h2 ? 3 : 6, p.T, p.H, p.W);
            // the act3 buffer of the next conv overlays the latent image: its halo is zeroed now that conv 0 has read the latent image
            if (lat0 && blk + 1 < d->n_blocks)
                if (int rc = zero_halo(Xp, 4 * B, p.T, p.H, p.W, act_rowb, st)) return rc;
            if (p_out) {
                if (int rc = gn_finalize(part, fin, stats, B, gn_tiles, (double)p.THW * (VC / VG), d->gn_eps, st)) return rc;
                return toimg_from_p(blk);
            }
        } else {
            static const int tag = prof_tag_id("conv3d_k3_gelu_stats_kernel<64>");
            ProfScope prof(tag, 2.0 * (double)B * p.THW * VC * 27.0 * VC, st);
            hipLaunchKernelGGL(conv3d_k3_gelu_stats_kernel<64>, dim3((unsigned)(B * p.tiles)), dim3(256), lds, st, a);
            AVD_CHECK_LAUNCH("conv3d");
        }
        if (int rc = gn_finalize(part, fin, stats, B, gn_tiles, (double)p.THW * (VC / VG), d->gn_eps, st)) return rc;
        if (blk + 1 < d->n_blocks && s3) {
            const int64_t total8 = (int64_t)B * p.THW * 4;       // threads: (voxel, 16-channel slab)
            static const int tag = prof_tag_id("gn_apply_pad3_kernel");
            ProfScope prof(tag, 10.0 * (double)B * p.THW * VC, st);
            if (h2)
                hipLaunchKernelGGL(gn_apply_pad3_kernel<true>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, Y, stats, d->gn_w[blk],
                                   d->gn_b[blk], X3, p.T, p.H, p.W, total8 / 4, d->conv_a_scale[blk + 1], act_slab);
            else
                hipLaunchKernelGGL(gn_apply_pad3_kernel<false>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, Y, stats, d->gn_w[blk],
                                   d->gn_b[blk], X3, p.T, p.H, p.W, total8 / 4, 0.f, act_slab);
            AVD_CHECK_LAUNCH("gn_apply_pad3");
        } else if (blk + 1 < d->n_blocks) {
            const int64_t total4 = (int64_t)B * p.THW * (VC / 4);
            static const int tag = prof_tag_id("gn_apply_pad_kernel");
            ProfScope prof(tag, 8.0 * (double)B * p.THW * VC, st);
            hipLaunchKernelGGL(gn_apply_pad_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, Y, stats,
                               d->gn_w[blk], d->gn_b[blk], Xp, p.T, p.H, p.W, total4);
            AVD_CHECK_LAUNCH("gn_apply_pad");
        } else {
            const int64_t nvox = (int64_t)B * p.THW;
            static const int tag = prof_tag_id("gn_apply_toimg_kernel");
            ProfScope prof(tag, 4.0 * ((double)nvox * VC + (double)nvox * d->out_ch), st);
            hipLaunchKernelGGL(gn_apply_toimg_kernel, dim3((unsigned)((nvox + TOIMG_VOX - 1) / TOIMG_VOX)), dim3(256), 0, st, Y, stats,
                               d->gn_w[blk], d->gn_b[blk], d->to_img_w, d->to_img_b, out, (int)p.THW, d->out_ch,
                               d->out_tanh, nvox);
            AVD_CHECK_LAUNCH("gn_apply_toimg");
        }
    }
    return AVD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// VideoVAE.encode — vae_video3d.py:164-189 (deterministic path): enc_net -> AvgPool3d(t_down,s_down,s_down) -> to_lat
// ---------------------------------------------------------------------------------------------------------
namespace avd {
struct VaeEncPlan {
    int tiles;
    int64_t THW, pad4_b, pad_b, y_b, part_b, part_n, fold_b, stats_b, total;
};
static int vae_enc_plan(const avd_vae_encode_desc* d, VaeEncPlan& p) {
    AVD_REQUIRE(d, AVD_EINVAL, "vae_encode: null descriptor");
    AVD_REQUIRE(d->B > 0 && d->T > 0 && d->H > 0 && d->W > 0, AVD_EINVAL, "vae_encode: bad input dims");
    AVD_REQUIRE(d->in_ch >= 1 && d->in_ch <= 4, AVD_EUNSUPPORTED, "vae_encode: 1..4 input channels supported");
    AVD_REQUIRE(d->base == VC, AVD_EUNSUPPORTED, "vae_encode: encoder width %d unsupported (kernels are built for 64)", d->base);
    AVD_REQUIRE(d->n_blocks >= 1 && d->n_blocks <= 8, AVD_EUNSUPPORTED, "vae_encode: 1..8 conv blocks supported");
    AVD_REQUIRE(d->lat_ch >= 1 && d->lat_ch <= 16, AVD_EUNSUPPORTED, "vae_encode: 1..16 latent channels supported");
    AVD_REQUIRE(d->t_down > 0 && d->s_down > 0 && d->T % d->t_down == 0 && d->H % d->s_down == 0 && d->W % d->s_down == 0,
                AVD_EINVAL, "vae_encode: T,H,W must be divisible by (t_down, s_down, s_down) — crop first");
    p.THW = (int64_t)d->T * d->H * d->W;
    AVD_REQUIRE((int64_t)(d->T + 2) * (d->H + 2) * (d->W + 2) * VC < (1ll << 31), AVD_EUNSUPPORTED,
                "vae_encode: one sample's activation exceeds 2^31 elements");
    p.tiles = (int)((p.THW + VBM - 1) / VBM);
    const int64_t padvox = (int64_t)d->B * (d->T + 2) * (d->H + 2) * (d->W + 2);
    p.pad4_b = a256(padvox * 4 * 4);
    p.pad_b = d->n_blocks > 1 ? a256(padvox * (d->conv_w3 ? A3_ROWB : VC * 4)) : 0;
    p.y_b = a256((int64_t)d->B * p.THW * VC * 4);
    {
        const int t3 = conv3_tiles(3, d->T, d->H, d->W), t6 = conv3_tiles(6, d->T, d->H, d->W);
        const int tmax = p.tiles > t3 ? (p.tiles > t6 ? p.tiles : t6) : (t3 > t6 ? t3 : t6);
        p.part_n = (int64_t)d->B * tmax * 2 * VG * 2;
        p.part_b = a256(p.part_n * 4 + gn_fin_bytes(d->B));
    }
    p.stats_b = a256((int64_t)d->B * VG * 2 * 4);
    // folded route (three planes, two blocks, packed first conv): per-sample weight image + bias table of conv 1
    p.fold_b = d->conv_w3 && d->conv_terms != 3 && d->n_blocks == 2 && d->conv0_pk_w3 ? a256((int64_t)d->B * (W3_BYTES + 64 * VC * 4)) : 0;
    p.total = p.pad4_b + p.pad_b + p.y_b + p.part_b + p.fold_b + p.stats_b;
    return AVD_OK;
}
}  // namespace avd

extern "C" int64_t avd_vae_encode_workspace_bytes(const avd_vae_encode_desc* d) {
    VaeEncPlan p;
    if (vae_enc_plan(d, p)) return -1;
    return p.total;
}

extern "C" int avd_vae_encode_f32(const avd_vae_encode_desc* d, const float* x, float* z, void* workspace,
                                  int64_t workspace_bytes, avd_stream_t stream) {
    VaeEncPlan p;
    if (int rc = vae_enc_plan(d, p)) return rc;
    AVD_REQUIRE(x && z && d->conv_w && d->conv_b && d->gn_w && d->gn_b && d->to_lat_w && d->to_lat_b, AVD_EINVAL,
                "vae_encode: null pointer");
    AVD_REQUIRE(workspace && workspace_bytes >= p.total, AVD_EWORKSPACE, "vae_encode: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)p.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(workspace);
    float* Xp4 = reinterpret_cast<float*>(w);
    float* Xp = reinterpret_cast<float*>(w + p.pad4_b);
    float* Y = reinterpret_cast<float*>(w + p.pad4_b + p.pad_b);
    float* part = reinterpret_cast<float*>(w + p.pad4_b + p.pad_b + p.y_b);
    double* fin = reinterpret_cast<double*>(part + p.part_n);
    char* foldw = w + p.pad4_b + p.pad_b + p.y_b + p.part_b;
    float* stats = reinterpret_cast<float*>(foldw + p.fold_b);
    const int B = d->B, T = d->T, H = d->H, W = d->W;

    const bool s3 = d->conv_w3 != nullptr && d->n_blocks > 1;     // split operands for the 64 -> 64 convolutions (block 0 is 4 -> 64, fp32)
    const bool h2 = s3 && d->conv_terms == 3;
    unsigned char* X3 = reinterpret_cast<unsigned char*>(Xp);
    if (s3) {
        for (int blk = 1; blk < d->n_blocks; ++blk) AVD_REQUIRE(d->conv_w3[blk], AVD_EINVAL, "vae_encode: null conv_w3[%d]", blk);
        if (int rc = check_conv_terms(d->conv_terms, d->conv_w_scale, d->conv_a_scale, d->n_blocks, 1, false)) return rc;
    }
    {
        // ---- folded route (round 5; three planes, two blocks, pooling (4, 8, 8), in_ch <= 8): the first conv on the halo-tile kernel with two taps
        // per k-step from a 96-byte image of the input, its output straight into conv 1's operand image, GroupNorm 0 folded into conv 1's
        // per-sample weights, conv 1's epilogue -> pooling partial sums -> GroupNorm 1 + AvgPool + to_lat.  No fp32 activation is written.
        const int64_t padvox = (int64_t)B * (T + 2) * (H + 2) * (W + 2);
        if (s3 && !h2 && g_vae_fold && p.fold_b > 0 && d->in_ch <= 8 && d->t_down == 4 && d->s_down == 8 && padvox * L16_ROWB <= p.y_b) {
            unsigned char* Lm = reinterpret_cast<unsigned char*>(Y);
            unsigned char* wimg = reinterpret_cast<unsigned char*>(foldw);
            float* btab1 = reinterpret_cast<float*>(wimg + (int64_t)B * W3_BYTES);
            const int64_t nvox = (int64_t)B * p.THW;
            if (int rc = zero_halo(Lm, B, T, H, W, L16_ROWB, st)) return rc;
            {
                static const int tag = prof_tag_id("rgb_lat16_kernel");
                ProfScope prof(tag, (double)nvox * (L16_ROWB + 4.0 * d->in_ch), st);
                hipLaunchKernelGGL(rgb_lat16_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, x, Lm, d->in_ch, T, H, W, nvox);
                AVD_CHECK_LAUNCH("rgb_lat16");
            }
            const int64_t slab_stride = padvox * L16_ROWB;       // conv 1's operand image: slab-major (conv3d_k3_bf16x3_kernel)
            if (int rc = zero_halo(Xp, 4 * B, T, H, W, L16_ROWB, st)) return rc;
            const int gn_tiles = conv3_tiles(6, T, H, W);
            {
                Conv3Args a3{Lm, static_cast<const unsigned char*>(d->conv0_pk_w3), d->conv_b[0], nullptr, part, T, H, W, p.tiles, 1.f, nullptr, nullptr, X3,
                             nullptr, nullptr, 0, 0, 0, slab_stride};
                if (int rc = conv3_launch(a3, 6, B, 2.0 * (double)B * p.THW * VC * 27.0 * 16, st, true, 1, true)) return rc;
            }
            if (int rc = gn_finalize(part, fin, stats, B, gn_tiles, (double)p.THW * (VC / VG), d->gn_eps, st)) return rc;
            {
                static const int tag = prof_tag_id("conv3_weight_gn_kernel");
                ProfScope prof(tag, (double)B * (W3_BYTES + 27.0 * VC * VC * 4), st);
                hipLaunchKernelGGL(conv3_weight_gn_kernel, dim3(54, B), dim3(256), 0, st, d->conv_w[1], stats, d->gn_w[0], wimg);
                AVD_CHECK_LAUNCH("conv3_weight_gn");
                hipLaunchKernelGGL(conv3_gn_btab_kernel, dim3(B, VC / 8), dim3(256), 0, st, d->conv_w[1], stats, d->gn_w[0], d->gn_b[0], btab1);
                AVD_CHECK_LAUNCH("conv3_gn_btab");
            }
            {
                Conv3Args a3{X3, wimg, d->conv_b[1], nullptr, part, T, H, W, p.tiles, 1.f, nullptr, btab1, nullptr, Y, nullptr, W3_BYTES, 64 * VC, slab_stride, 0};
                if (int rc = conv3_launch(a3, 6, B, 2.0 * (double)B * p.THW * VC * 27.0 * VC, st, false, 3)) return rc;
            }
            if (int rc = gn_finalize(part, fin, stats, B, gn_tiles, (double)p.THW * (VC / VG), d->gn_eps, st)) return rc;
            const int64_t nlat = (int64_t)B * (T / 4) * (H / 8) * (W / 8);
            static const int tag = prof_tag_id("pool_tolat_from_partials_kernel");
            ProfScope prof(tag, 4.0 * (double)nlat * 8 * VC, st);
            hipLaunchKernelGGL(pool_tolat_from_partials_kernel<HaloCfg<6>::TH>, dim3((unsigned)((nlat + 3) / 4)), dim3(256), 0, st, Y, stats, d->gn_w[1],
                               d->gn_b[1], d->to_lat_w, d->to_lat_b, z, T, H, W, gn_tiles * 2, d->lat_ch, nlat);
            AVD_CHECK_LAUNCH("pool_tolat_from_partials");
            return AVD_OK;
        }
    }
    const int act_rowb = h2 ? 64 : L16_ROWB;
    const int64_t act_slab = (int64_t)B * (T + 2) * (H + 2) * (W + 2) * act_rowb;      // slab-major operand image of the split-operand convs
    if (int rc = zero_halo(Xp4, B, T, H, W, 16, st)) return rc;
    if (d->n_blocks > 1)
        if (int rc = s3 ? zero_halo(Xp, 4 * B, T, H, W, act_rowb, st) : zero_halo(Xp, B, T, H, W, VC * 4, st)) return rc;
    {
        const int64_t nvox = (int64_t)B * p.THW;
        hipLaunchKernelGGL(rgb_to_ndhwc4_pad_kernel, dim3((unsigned)((nvox + 255) / 256)), dim3(256), 0, st, x, Xp4, d->in_ch,
                           T, H, W, nvox);
        AVD_CHECK_LAUNCH("rgb_to_ndhwc4_pad");
    }
    constexpr int stage_lds = 2 * (VBM + VC) * VBK * 4, epi_lds = 4 * 64 * 36 * 4;
    constexpr int lds = stage_lds > epi_lds ? stage_lds : epi_lds;
    for (int blk = 0; blk < d->n_blocks; ++blk) {
        ConvArgs a{blk == 0 ? Xp4 : Xp, d->conv_w[blk], d->conv_b[blk], Y, part, T, H, W, p.tiles};
        int gn_tiles = p.tiles;
        // last conv on the halo-tile kernel with the shipped pooling (4, 8, 8): pooling partial sums instead of fp32 activations ("vae_fold")
        const bool pool_out = blk > 0 && s3 && blk + 1 == d->n_blocks && g_vae_fold && d->t_down == 4 && d->s_down == 8;
        if (blk > 0 && s3) {
            Conv3Args a3{X3, static_cast<const unsigned char*>(d->conv_w3[blk]), d->conv_b[blk], Y, part, T, H, W, p.tiles,
                         h2 ? 1.0f / (d->conv_w_scale[blk] * d->conv_a_scale[blk]) : 1.f, nullptr};
            a3.x_slab_stride = act_slab;
            if (pool_out) { a3.Y = nullptr; a3.P = Y; }
            if (int rc = conv3_launch(a3, h2 ? 3 : 6, B, 2.0 * (double)B * p.THW * VC * 27.0 * VC, st, false, pool_out ? 3 : 0)) return rc;
            gn_tiles = conv3_tiles(h2 ? 3 : 6, T, H, W);
        } else if (blk == 0) {
            static const int tag = prof_tag_id("conv3d_k3_gelu_stats_kernel<4>");
            ProfScope prof(tag, 2.0 * (double)B * p.THW * VC * 27.0 * d->in_ch, st);
            hipLaunchKernelGGL(conv3d_k3_gelu_stats_kernel<4>, dim3((unsigned)(B * p.tiles)), dim3(256), lds, st, a);
        } else {
            static const int tag = prof_tag_id("conv3d_k3_gelu_stats_kernel<64>");
            ProfScope prof(tag, 2.0 * (double)B * p.THW * VC * 27.0 * VC, st);
            hipLaunchKernelGGL(conv3d_k3_gelu_stats_kernel<64>, dim3((unsigned)(B * p.tiles)), dim3(256), lds, st, a);
        }
        AVD_CHECK_LAUNCH("conv3d(enc)");
        if (int rc = gn_finalize(part, fin, stats, B, gn_tiles, (double)p.THW * (VC / VG), d->gn_eps, st)) return rc;
        if (blk + 1 < d->n_blocks && s3) {
            const int64_t total8 = (int64_t)B * p.THW * 4;       // threads: (voxel, 16-channel slab)
            static const int tag = prof_tag_id("gn_apply_pad3_kernel");
            ProfScope prof(tag, 10.0 * (double)B * p.THW * VC, st);
            if (h2)
                hipLaunchKernelGGL(gn_apply_pad3_kernel<true>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, Y, stats, d->gn_w[blk],
                                   d->gn_b[blk], X3, T, H, W, total8 / 4, d->conv_a_scale[blk + 1], act_slab);
            else
                hipLaunchKernelGGL(gn_apply_pad3_kernel<false>, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, st, Y, stats, d->gn_w[blk],
                                   d->gn_b[blk], X3, T, H, W, total8 / 4, 0.f, act_slab);
            AVD_CHECK_LAUNCH("gn_apply_pad3");
        } else if (blk + 1 < d->n_blocks) {
            const int64_t total4 = (int64_t)B * p.THW * (VC / 4);
            static const int tag = prof_tag_id("gn_apply_pad_kernel");
            ProfScope prof(tag, 8.0 * (double)B * p.THW * VC, st);
            hipLaunchKernelGGL(gn_apply_pad_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, Y, stats,
                               d->gn_w[blk], d->gn_b[blk], Xp, T, H, W, total4);
            AVD_CHECK_LAUNCH("gn_apply_pad");
        } else if (pool_out) {
            const int64_t nlat = (int64_t)B * (T / 4) * (H / 8) * (W / 8);
            static const int tag = prof_tag_id("pool_tolat_from_partials_kernel");
            ProfScope prof(tag, 4.0 * (double)nlat * 8 * VC, st);
            if (h2)
                hipLaunchKernelGGL(pool_tolat_from_partials_kernel<HaloCfg<3>::TH>, dim3((unsigned)((nlat + 3) / 4)), dim3(256), 0, st, Y, stats, d->gn_w[blk],
                                   d->gn_b[blk], d->to_lat_w, d->to_lat_b, z, T, H, W, gn_tiles * 2, d->lat_ch, nlat);
            else
                hipLaunchKernelGGL(pool_tolat_from_partials_kernel<HaloCfg<6>::TH>, dim3((unsigned)((nlat + 3) / 4)), dim3(256), 0, st, Y, stats, d->gn_w[blk],
                                   d->gn_b[blk], d->to_lat_w, d->to_lat_b, z, T, H, W, gn_tiles * 2, d->lat_ch, nlat);
            AVD_CHECK_LAUNCH("pool_tolat_from_partials");
        } else {
            const int64_t nlat = (int64_t)B * (T / d->t_down) * (H / d->s_down) * (W / d->s_down);
            static const int tag = prof_tag_id("gn_pool_tolat_kernel");
            ProfScope prof(tag, 4.0 * (double)B * p.THW * VC, st);
            hipLaunchKernelGGL(gn_pool_tolat_kernel, dim3((unsigned)((nlat + 3) / 4)), dim3(256), 0, st, Y, stats, d->gn_w[blk],
                               d->gn_b[blk], d->to_lat_w, d->to_lat_b, z, T, H, W, d->t_down, d->s_down, d->lat_ch, nlat);
            AVD_CHECK_LAUNCH("gn_pool_tolat");
        }
    }
    return AVD_OK;
}
