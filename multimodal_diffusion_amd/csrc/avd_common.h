// Internal helpers shared by the gfx950 kernels of libavdiff_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/avdiff_hip.h"

namespace avd {

int set_error(int code, const char* fmt, ...);

#define AVD_REQUIRE(cond, code, ...)                       \
    do {                                                   \
        if (!(cond)) return avd::set_error((code), __VA_ARGS__); \
    } while (0)

#define AVD_CHECK_LAUNCH(name)                                                          \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess)                                                          \
            return avd::set_error(AVD_ELAUNCH, "%s: %s", (name), hipGetErrorString(e__)); \
    } while (0)

// Dynamic-LDS limit of one kernel, raised once per device (hipFuncSetAttribute acts on the current device's copy of the
// function).  Idempotent: two host threads racing here only repeat the call.
struct LdsAttr {
    std::atomic<uint64_t> done{0};
    int ensure(const void* kern, int lds_bytes, const char* what);
};

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

// row index of accumulator register r for the lane half hi in a 32x32 MFMA C/D tile
__device__ __forceinline__ int mfma32_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// GELU(x) = 0.5 x (1 + erf(x/sqrt2)) with erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 absolute, branch-free:
// one v_rcp, one v_exp, five FMAs).  In fp32 the resulting GELU is within 4.7e-7 absolute of the fp64 value over
// [-12, 12] — the same as an fp32 evaluation through a correctly-rounded erf (4.5e-7) — at about a third of the
// instruction count of libm's two-branch erff, which matters in the fc1 GEMM epilogue.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = x * 0.70710678118654752440f;
    const float a = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-a * a * 1.4426950408889634f);
    const float er = copysignf(fmaf(-p, e, 1.0f), z);      // explicit fma: the same rounding in every kernel that inlines this
    return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if constexpr (ACT == AVD_ACT_GELU) return gelu_erf(x);
    else if constexpr (ACT == AVD_ACT_SILU) return silu(x);
    else return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Segmented row addressing: logical row r of a [rows, ld] matrix whose rows come in groups of `seg`
// consecutive rows, groups `stride` floats apart (seg <= 0 means plain row-major).
struct RowMap {
    int64_t ld;
    int64_t seg;
    int64_t stride;
    __host__ __device__ __forceinline__ int64_t off(int64_t r) const {
        return seg > 0 ? (r / seg) * stride + (r % seg) * ld : r * ld;
    }
};

// A operand of a GEMM gathered straight from a video latent [B, C, T, H, W] in tube-token order (ops.py:100-127
// tube_patch_video: token n = ((T/t index) * H/h + H/h index) * W/w + W/w index, element k = ((c * t + dt) * h + dy) * w + dx):
// logical A[row = b * Nt + n][k] = z[b][c][nt*t + dt][ny*h + dy][nx*w + dx].  w % 4 == 0 and W % 4 == 0 keep every float4 along k
// one aligned 16-byte run of the latent.
struct TubeGather {
    int T, H, W, t, h, w, Ht, Wt, Nt;
    int64_t per;       // C*T*H*W
    __host__ __device__ __forceinline__ int64_t row_off(int64_t row) const {
        const int64_t b = row / Nt;
        const int n = (int)(row - b * Nt);
        const int nx = n % Wt, ny = (n / Wt) % Ht, nt = n / (Wt * Ht);
        return b * per + ((int64_t)(nt * t) * H + ny * h) * W + nx * w;
    }
    __host__ __device__ __forceinline__ int64_t k_off(int k) const {
        const int dx = k % w, dy = (k / w) % h, dt = (k / (w * h)) % t, c = k / (w * h * t);
        return (((int64_t)c * T + dt) * H + dy) * W + dx;
    }
};

// ---- split3 operand image of the bf16x3 matmul path (layout and rationale: gemm_bf16x3.hip) ----
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int S3_CHUNK = 128 * 96;   // one (128-row tile, 16-k group) chunk: three planes of [128 rows][32 B]
constexpr int S3_PLANE = 128 * 32;

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two fp32 -> two bf16 (round to nearest even) packed in one dword: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned int p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int p) { return __uint_as_float(p & 0xffff0000u); }

// v[8] -> three 16-byte chunks of 8 bf16: h = rn(x), m = rn(x - h), l = rn(x - h - m); both residuals are exact in fp32.
// Edge of the range (rare, so behind one wave-uniform test): a finite |x| > 0x1.fep127 would round to a bf16 infinity —
// h keeps the largest finite bf16 instead and the remainder moves to m, the sum is still exact; x = +-inf gives
// (inf, 0, 0) rather than (inf, NaN, NaN).  NaN stays NaN in every plane.
// No contraction in here: hipcc's default -ffp-contract=fast fuses across inlined statements, and `a - h` behind a caller's
// `a = p * q` became fma(p, q, -h) in SOME instantiations (the fused MLP's GELU epilogue, not the fc1 kernel's) — the planes then
// summed to the unrounded product's leading bits instead of to the fp32 value a, and two kernels running the same arithmetic
// disagreed in the last bit of a step.  With the pragma the three planes sum to v[e] exactly, in every kernel.
template <bool RANGE_CHECK = true>     // false: the caller guarantees |v| far below the top of the range (softmax probabilities)
__device__ __forceinline__ void split8(const float* v, u32x4& H, u32x4& Mi, u32x4& Lo) {
#pragma clang fp contract(off)
    bool edge = false;
    if constexpr (RANGE_CHECK) {
        float amax = fabsf(v[0]);
#pragma unroll
        for (int e = 1; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));    // fmaxf ignores NaN operands
        edge = __any(amax > 3.3895313892515355e38f);                      // largest finite bf16
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = v[2 * e], b = v[2 * e + 1];
        unsigned int h = pk_bf16(a, b);
        float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
        if (edge) {
            if ((h & 0x7fffu) == 0x7f80u) {                  // low half became +-inf
                if (fabsf(a) < __builtin_inff()) { h = (h & 0xffff8000u) | 0x7f7fu; ra = a - bf16_lo(h); } else ra = 0.f;
            }
            if ((h & 0x7fff0000u) == 0x7f800000u) {          // high half became +-inf
                if (fabsf(b) < __builtin_inff()) { h = (h & 0x8000ffffu) | 0x7f7f0000u; rb = b - bf16_hi(h); } else rb = 0.f;
            }
        }
        const unsigned int m = pk_bf16(ra, rb);
        const float sa = ra - bf16_lo(m), sb = rb - bf16_hi(m);
        H[e] = h;
        Mi[e] = m;
        Lo[e] = pk_bf16(sa, sb);
    }
}

// where the 8 values (row r, columns k..k+7, k % 8 == 0) of a [rows][K] matrix go in its split3 image
template <bool RANGE_CHECK = true>
__device__ __forceinline__ void store_split8(unsigned char* img, int64_t r, int k, int K, const float* v) {
    u32x4 H, Mi, Lo;
    split8<RANGE_CHECK>(v, H, Mi, Lo);
    const int rr = (int)(r & 127), half = (k >> 3) & 1;
    unsigned char* dst = img + ((r >> 7) * (K >> 4) + (k >> 4)) * (int64_t)S3_CHUNK + rr * 32 + ((half ^ ((rr >> 4) & 1)) << 4);
    *reinterpret_cast<u32x4*>(dst) = H;
    *reinterpret_cast<u32x4*>(dst + S3_PLANE) = Mi;
    *reinterpret_cast<u32x4*>(dst + 2 * S3_PLANE) = Lo;
}

// ---- "f16x2" operand images (matmul mode with two fp16 planes and three product terms) -------------------------------------
// Same geometry as the split3 / qkv3 images; plane 0 holds h = rn_f16(s x), plane 1 holds l = rn_f16(s x - h), plane 2 is unused
// (never written, never moved).  s is a power of two chosen by the caller from a bound on |x| so that |s x| <= 2^15: fp16 has
// only 5 exponent bits, so unlike bf16x3 the image is scaled; the consumer divides the scales out of its fp32 accumulators,
// which is exact.  h + l carries 22 significant bits of s x (|error| <= 2^-22 |x|, or 2^-25 / s absolute for elements more than
// 17 binades below the top of the range), and a product keeps hh + hl + lh (ll <= 2^-22 relative is dropped).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned int pk_f16(float a, float b) {      // v_cvt_pk_f16_f32, round to nearest even
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ f32x2 unpk_f16(unsigned int p) { return __builtin_convertvector(__builtin_bit_cast(f16x2, p), f32x2); }

// A value past the fp16 range (the caller's bound was wrong, or x is inf / NaN) becomes inf in h and NaN in l: the rows it
// touches come out NaN instead of silently saturated.
__device__ __forceinline__ void split8_h2(const float* v, float s, u32x4& H, u32x4& L) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = v[2 * e] * s, b = v[2 * e + 1] * s;
        const unsigned int h = pk_f16(a, b);
        const f32x2 u = unpk_f16(h);
        H[e] = h;
        L[e] = pk_f16(a - u[0], b - u[1]);
    }
}
__device__ __forceinline__ void store_split8_h2(unsigned char* img, int64_t r, int k, int K, const float* v, float s) {
    u32x4 H, L;
    split8_h2(v, s, H, L);
    const int rr = (int)(r & 127), half = (k >> 3) & 1;
    unsigned char* dst = img + ((r >> 7) * (K >> 4) + (k >> 4)) * (int64_t)S3_CHUNK + rr * 32 + ((half ^ ((rr >> 4) & 1)) << 4);
    *reinterpret_cast<u32x4*>(dst) = H;
    *reinterpret_cast<u32x4*>(dst + S3_PLANE) = L;
}

// one 32x32x16 MFMA on 16-bit operands held as bf16x8 bit patterns: bf16 planes, or (F16) the fp16 planes of an f16x2 image
template <bool F16>
__device__ __forceinline__ f32x16 mma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// planes a matmul mode moves and reads: terms 1 -> h; 3 (f16x2) -> h, l; 6 / 9 -> h, m, l
__host__ __device__ constexpr int s3_planes(int terms) { return terms == 1 ? 1 : terms == 3 ? 2 : 3; }

// ---- qkv3 image of the bf16x3 attention (attn_bf16x3.hip): for part in {q,k,v}, sample b, head h, rows n in [0,Npad) of
// 384 B = [plane h | m | l][64 d bf16]; the 16-byte chunk c (8 d) of a row sits at slot c ^ qkv3_swizzle(part, n) ----
constexpr int QKV3_ROWB = 384;
__host__ __device__ __forceinline__ int qkv3_npad(int n) { return (n + 63) / 64 * 64; }
__host__ __device__ __forceinline__ int qkv3_swizzle(int part, int n) {
    // q: read straight from global; k: ds_read_b128 of 32 key rows; v: ds_read_b64_tr_b16 of 4-key x 16-d blocks
    // (v, round 5: bit 2 of the key joins the swizzle — conflict-free for the transposed reads of BOTH MFMA shapes; with bit 1 alone the
    // 16x16x32 kernel's reads, whose 32-lane half spans keys k and k + 4, were 2-way; brute-forced over all linear 4-bit swizzles)
    return part == 0 ? 0 : part == 1 ? (n >> 1) & 7 : (((n >> 1) & 1) << 2) | (((n >> 2) & 1) << 1);
}

// measurement hooks (see avd_prof_enable): RAII bracket around one launch
extern bool g_prof_on;
void prof_mark(int tag, double work, hipStream_t st, bool begin);
int prof_tag_id(const char* fmt, ...);   // registers a kernel name once, returns its tag
struct ProfScope {
    int tag; hipStream_t st; bool on;
    ProfScope(int t, double work, hipStream_t s) : tag(t), st(s), on(g_prof_on) { if (on) prof_mark(tag, work, st, true); }
    ~ProfScope() { if (on) prof_mark(tag, 0.0, st, false); }
};

// internal launchers used by the composites (same kernels as the public entry points)
int gemm_f32(const float* A, RowMap am, const float* W, const float* bias, const float* R, RowMap rm,
             float* C, RowMap cm, int64_t M, int N, int K, int act, hipStream_t st);
// C = A W^T + bias with A gathered from the latent z through tg (the tube patch fused into the A-operand load)
int gemm_f32_tube(const float* z, const TubeGather& tg, const float* W, const float* bias, float* C, RowMap cm, int64_t M, int N, int K,
                  hipStream_t st);

extern int g_gemm_stages;        // 2 or 3 LDS stages for the 128x64 / 64x64 tiles (avd_tune_set "gemm_stages")
extern int g_gemm_force_tile;     // -1 = automatic tile choice; 0 / 1 / 2 = 128x128 / 128x64 / 64x64 (avd_tune_set "gemm_tile")
bool gemm_f32_fold_supported(int N, int K);
int gemm_f32_fold(const float* A, RowMap am, const float* W, const float* bias, const float* R, RowMap rm, float* C, RowMap cm,
                  int64_t M, int N, int K, int act, const float* ss_in, int ss_in_cols, float sqrt_d, float eps, float* ss_out,
                  hipStream_t st);
int rowss_f32(const float* x, float* ss, int64_t rows, int d, hipStream_t st);
// split-K of the fp32 residual GEMMs for batches of a few hundred rows (gemm_f32.hip): slices to use (0 = none), workspace bound, launch
constexpr int kGemmSplitKMax = 4;
extern int g_gemm_splitk;        // largest slice count tried (0 off; avd_tune_set "gemm_splitk")
int gemm_f32_splitk_slices(int64_t M, int N, int K);
int64_t gemm_f32_splitk_ws_max_floats(int64_t M, int N, int K);
int gemm_f32_splitk(const float* A, RowMap am, const float* W, const float* bias, const float* R, RowMap rm, float* C, RowMap cm, int64_t M,
                    int N, int K, int ns, float* part, float* ss_out, hipStream_t st);

int attn_f32(const float* qkv, float* out, int B, int N, int H, int Dh, float scale, int n_query, const unsigned char* key_padding_mask,
             hipStream_t st);
int rmsnorm_f32(const float* x, RowMap xm, const float* scale, float* y, RowMap ym, int64_t rows, int d, float eps,
                hipStream_t st);
int layernorm_act_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int d, float eps,
                      int act, hipStream_t st);
int weight_bounds_f32(const float* x, int64_t rows, int d, float* out2, hipStream_t st);

// bf16x3 path (gemm_bf16x3.hip)
int64_t split3_bytes(int64_t rows, int K);
// h2_scale > 0: write the f16x2 image with that scale instead of the three bf16 planes
int split3_f32(const float* x, int64_t ld, void* out, int64_t rows, int K, hipStream_t st, float h2_scale = 0.f);
// ss != null: also the rows' sums of squares per 64-column chunk, ss[row][K / 64] (a folded RMSNorm reads them)
int split3_rows_f32(const float* x, RowMap xm, void* out, int64_t rows, int K, hipStream_t st, float h2_scale = 0.f, float* ss = nullptr);
int rmsnorm_split3_f32(const float* x, const float* scale, void* out, int64_t rows, int d, float eps, hipStream_t st, float h2_scale = 0.f);
int layernorm_act_split3_f32(const float* x, const float* gamma, const float* beta, void* out, int64_t rows, int d, float eps, int act,
                             hipStream_t st, float h2_scale = 0.f);
bool gemm_bf16x3_supported(int64_t M, int N, int K);
int attn_f32_split3(const float* qkv, void* out3, int B, int N, int H, int Dh, float scale, int n_query, hipStream_t st);
extern int g_s3_stagger;
extern int g_mlp_fused;          // 1: fc1 -> GELU -> fc2 of the six-term bf16x3 path as one launch (mlp_bf16x3.hip; default 0, avd_tune_set "mlp_fused")
bool mlp_bf16x3_supported(int d, int hidden, int terms);
int mlp_bf16x3(const void* X3, const void* W1n3, const float* b1, const void* W23, const float* b2, const float* ss_in, float eps,
               const float* R, float* C, void* C3, float* ss_out, int64_t M, int d, int hidden, hipStream_t st);
extern int g_attn_m16;           // 1: the split-operand attention of the three-plane modes on v_mfma_f32_16x16x32 (2: every split mode; avd_tune_set "attn_m16")
extern int g_attn_pipe;          // 1 (default): the split-operand attention as one software pipeline per wave (avd_tune_set "attn_pipe")
extern int g_s3_m16;             // 1 (default): bf16x3 GEMMs on v_mfma_f32_16x16x32_bf16, two terms per MFMA; 0: 32x32x16 (avd_tune_set "s3_m16")
extern int g_s3_w128;            // 1: the 8-wave bf16x3 blocks with an image epilogue run as 4 waves with a 128 x 128 wave tile (avd_tune_set "s3_w128")
extern int g_s3_rt4;             // row tiles per wave of the 4-wave image-epilogue blocks: 0 automatic, 5 .. 8 = 160 .. 256-row blocks (avd_tune_set "s3_rt4")
extern int g_s3_deep4;           // 1: residual + image launches whose 4-wave blocks fit the CUs once run one block per CU on a four-stage ring (avd_tune_set "s3_deep4")
extern int g_s3_rt;              // rows per 8-wave block of the residual + image epilogue: 0 automatic, 7 = 224 rows, 8 = 256 rows (avd_tune_set "s3_rt")
extern int g_vae_lat;            // 1 (default): the decoder's first conv composed with from_lat and the upsample when the descriptor allows (vae3d_f32.hip; avd_tune_set "vae_lat")
extern int g_vae_fold;           // 1 (default): conv 0 -> conv 1 through the operand image with the GroupNorm folded into conv 1's weights, to_img from partial sums (vae3d_f32.hip; avd_tune_set "vae_fold")
extern int g_codec_mfma;         // 1 (default): the codec's 64 -> 64 conv1d layers on the fp32 matrix pipe (codec_f32.hip; avd_tune_set "codec_mfma")
extern int g_cfg_rows;           // 1 (default): fused CFG + un-patch + DDIM through whole 128-byte lines (tokens.hip; avd_tune_set "cfg_rows")
extern int g_s3_sn, g_s3_super4, g_s3_super8;      // super-tile shape overrides of the split GEMMs, 0 = default (avd_tune_set "s3_sn" / "s3_super4" / "s3_super8")
extern int g_s3_tile;            // -1 = per epilogue; 0 / 1 = 8-wave 256x256 / 4-wave 256x128 blocks (avd_tune_set "s3_tile")
extern thread_local bool t_s3_two_streams;
// terms == 3 (f16x2): ab_scale = (A image scale) x (W image scale), c_scale = scale of the image written (if one is written)
int gemm_bf16x3_qkv3(const void* A3, const void* W3, const float* bias, void* img, int64_t M, int tokens, int heads, int K, float qscale,
                     int terms, hipStream_t st, float ab_scale = 1.f, float c_scale = 1.f, const float* ss_in = nullptr, float eps = 0.f);
int64_t qkv3_bytes(int B, int N, int H);
// fp8 attention (attn_fp8.hip): reads the same qkv3 image, needs attn_fp8_ws_bytes(B, N, H) of scratch
int64_t attn_fp8_ws_bytes(int B, int N, int H);
int attn_fp8(const void* qkv3, void* ws, int64_t ws_bytes, float* out, void* out3, int B, int N, int H, int n_query, hipStream_t st,
             int img_terms = 6, float img_scale = 1.f, float out_scale = 1.f);
// terms == 3: img_scale = scale of the qkv image, out_scale = scale of the image written to out3
// out_tokens (0 = N): rows per sample of the output — n_query <= out_tokens: the last block writes its target rows compactly
int attn_bf16x3(const void* qkv3, float* out, void* out3, int B, int N, int H, int n_query, int terms, hipStream_t st,
                float img_scale = 1.f, float out_scale = 1.f, int out_tokens = 0);
int gemm_bf16x3(const void* A3, const void* W3, const float* bias, const float* R, float* C, void* C3, int64_t M, int N, int K,
                int act, int terms, hipStream_t st, float ab_scale = 1.f, float c_scale = 1.f, const float* ss_in = nullptr, float eps = 0.f,
                float* ss_out = nullptr, const float* gamma = nullptr, int r_seg = 0, int r_stride = 0);
// r_seg > 0 (fp32 + image residual epilogue of the six-term 16x16x32 kernels): output row m adds R row (m / r_seg) * r_stride + m % r_seg
bool gemm_bf16x3_resmap_supported(int terms);
// gamma != null (f16x2 images, N == 512): C = A W^T + bias + R as fp32 AND C3 = image (scale c_scale) of RMSNorm(C; gamma, eps)
bool gemm_bf16x3_rownorm_supported(int N, int terms);
// split-K for residual GEMMs that cannot fill the chip (small batches): slices to use (0 = none), workspace, launch
extern int g_s3_splitk;          // slices tried (0 off; avd_tune_set "s3_splitk")
int gemm_bf16x3_splitk_slices(int64_t M, int N, int K, int terms);
int64_t gemm_bf16x3_splitk_ws_floats(int64_t M, int N, int ns);
constexpr int kS3SplitKMax = 4;  // upper bound of "s3_splitk" (workspaces are sized for it)
int64_t gemm_bf16x3_splitk_ws_max_floats(int64_t M, int N, int K);
int gemm_bf16x3_splitk(const void* A3, const void* W3, const float* bias, const float* R, float* C, void* C3, float* ss, int64_t M, int N,
                       int K, int terms, int ns, float* part, hipStream_t st);

}  // namespace avd
