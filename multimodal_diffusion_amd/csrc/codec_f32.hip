// AudioCodec on MI355X — the audio end of the loop boundary:
//   avdiff/models/encoders/audio_codec.py:88-133 (layers), :158-182 (_avgpool_frames), :184-199 (encode), :200-214 (decode).
// 97 k parameters, ~2 GFLOP per 3-second clip, run once per sample: a direct NCL conv1d on the vector ALUs with the
// input tile in LDS is ample (the matrix cores would sit idle behind the 1-channel ends).  One thread owns one output
// position and 16 output channels; weights are wave-uniform scalar loads; the nearest-neighbour x hop upsample of the
// decoder is folded into the first smoothing conv's input indexing, so the 64 x 48,000 upsampled signal is never stored.
#include "avd_common.h"

#include <stdlib.h>

namespace avd {

constexpr int C1_TILE = 256;    // output positions per block
constexpr int C1_OC = 16;       // output channels per thread
constexpr int C1_CC = 16;       // input channels staged per pass
constexpr int C1_KMAX = 15;

// out[b][o][l] = act(bias[o] + sum_c sum_j w[o][c][j] * xin[b][c][l + j - pad]),  xin[c][p] = x[c][p / up] for 0 <= p < Lin*up
__global__ __launch_bounds__(C1_TILE) void conv1d_ncl_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             int Cin, int Cout, int Lin, int up, int k, int act) {
    __shared__ float xs[C1_CC][C1_TILE + C1_KMAX - 1];
    const int Lout = Lin * up, pad = k / 2;
    const int l0 = blockIdx.x * C1_TILE, o0 = blockIdx.y * C1_OC, b = blockIdx.z;
    const int tid = threadIdx.x;
    float acc[C1_OC];
#pragma unroll
    for (int i = 0; i < C1_OC; ++i) acc[i] = 0.f;
    const float* xb = x + (int64_t)b * Cin * Lin;
    const int span = C1_TILE + k - 1;
    for (int c0 = 0; c0 < Cin; c0 += C1_CC) {
        const int cc = Cin - c0 < C1_CC ? Cin - c0 : C1_CC;
        for (int i = tid; i < cc * span; i += C1_TILE) {
            const int c = i / span, p = i % span;
            const int pos = l0 + p - pad;                       // position in the (virtually upsampled) input
            xs[c][p] = (pos >= 0 && pos < Lout) ? xb[(int64_t)(c0 + c) * Lin + pos / up] : 0.f;
        }
        __syncthreads();
        for (int c = 0; c < cc; ++c) {
            for (int j = 0; j < k; ++j) {
                const float xv = xs[c][tid + j];
#pragma unroll
                for (int i = 0; i < C1_OC; ++i) {
                    const int o = o0 + i;                        // wave-uniform -> scalar weight load
                    if (o < Cout) acc[i] = fmaf(w[((int64_t)o * Cin + c0 + c) * k + j], xv, acc[i]);
                }
            }
        }
        __syncthreads();
    }
    const int l = l0 + tid;
    if (l < Lout) {
#pragma unroll
        for (int i = 0; i < C1_OC; ++i) {
            const int o = o0 + i;
            if (o < Cout) {
                float v = acc[i] + (bias ? bias[o] : 0.f);
                if (act == AVD_ACT_GELU) v = gelu_erf(v);
                else if (act == AVD_ACT_TANH) v = tanhf(v);
                out[((int64_t)b * Cout + o) * Lout + l] = v;
            }
        }
    }
}

// The two 64 -> 64 layers of each direction (k = 9 in the encoder, 7 in the decoder: 98 % of the codec's FLOPs) on the fp32 matrix pipe
// (round 5: the vector kernel above reached 14 TFLOP/s on them, 13 ms per batch of 32 three-second clips — 2 % of a whole generation).
// Implicit GEMM D[co][pos] = sum_kk W[co][kk] X[kk][pos], kk = (ci, tap): v_mfma_f32_32x32x2_f32 with A = weights (rows = output channels),
// B = the shifted input (columns = positions), so an accumulator register is 32 consecutive positions of one channel: coalesced NCL stores.
// A block = 256 positions x 64 channels, a wave = 64 positions (2 column tiles) x 64 channels (2 row tiles); per pass 16 input channels:
// their input rows [16][256 + k - 1] (nearest x up folded into the indexing, as above) and their weights [16 k][64] go to LDS; row
// strides are chosen so that the two k-halves of a fragment read (lanes 0..31 / 32..63) fall into different halves of the 64 banks.
typedef float f32x16c __attribute__((ext_vector_type(16)));
template <int K>
__global__ __launch_bounds__(256) void conv1d_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int Cin, int Lin, int up, int act) {
    constexpr int CC = 16, NKK = CC * K, PAD = K / 2, SPAN = 256 + K - 1;
    constexpr int ROWP = ((SPAN - (K + 31) + 63) / 64) * 64 + K + 31;      // >= SPAN, ROWP - K + 1 = 32 mod 64
    constexpr int WROW = 96;                                               // 64 channels, = 32 mod 64
    static_assert(ROWP >= SPAN && (ROWP - K + 1) % 64 == 32 && NKK % 2 == 0, "LDS row strides");
    __shared__ float xs[CC * ROWP];
    __shared__ float ws[NKK * WROW];
    const int Lout = Lin * up;
    const int l0 = blockIdx.x * 256, b = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    f32x16c acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* xb = x + (int64_t)b * Cin * Lin;
    // fragment addresses: weights ws[2 s + hi][32 rt + l31]; input xs[ci][64 wave + 32 ct + l31 + tap] with (ci, tap) of kk = 2 s + hi —
    // kk + 1 is the next tap of the same channel (one float on) or, behind the last tap, tap 0 of the next channel (ROWP - K + 1 on)
    const float* wa = ws + hi * WROW + l31;
    const float* xa = xs + wave * 64 + l31 + hi, * xw = xs + wave * 64 + l31 + hi * (ROWP - K + 1);
    for (int c0 = 0; c0 < Cin; c0 += CC) {
        for (int i = tid; i < CC * SPAN; i += 256) {
            const int c = i / SPAN, p = i - c * SPAN;
            const int pos = l0 + p - PAD;                       // position in the (virtually upsampled) input
            xs[c * ROWP + p] = (pos >= 0 && pos < Lout) ? xb[(int64_t)(c0 + c) * Lin + pos / up] : 0.f;
        }
        for (int i = tid; i < NKK * 64; i += 256) {
            const int co = i / NKK, kk = i - co * NKK;         // consecutive threads walk (ci, tap) of one output channel: contiguous in w
            ws[kk * WROW + co] = w[((int64_t)co * Cin + c0) * K + kk];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NKK / 2; ++s) {
            const int kk0 = 2 * s, ci = kk0 / K, tap = kk0 - ci * K;
            const float* xsrc = (tap == K - 1 ? xw : xa) + ci * ROWP + tap;
            const float a0 = wa[kk0 * WROW], a1 = wa[kk0 * WROW + 32];
            const float b0 = xsrc[0], b1 = xsrc[32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = rt * 32 + mfma32_row(r, hi);
            const float bv = bias ? bias[co] : 0.f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int l = l0 + wave * 64 + ct * 32 + l31;
                float v = acc[rt][ct][r] + bv;
                if (act == AVD_ACT_GELU) v = gelu_erf(v);
                else if (act == AVD_ACT_TANH) v = tanhf(v);
                if (l < Lout) out[((int64_t)b * 64 + co) * Lout + l] = v;
            }
        }
}

// F.pad / crop to Fa*hop then avg_pool1d(kernel = stride = hop): explicit zero padding counts in the mean
__global__ void avgpool_frames_kernel(const float* __restrict__ x, float* __restrict__ out, int L, int Fa, int hop,
                                      int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % Fa);
    const int64_t row = i / Fa;                                  // b*C + c
    const float* xr = x + row * L;
    float s = 0.f;
    const int beg = f * hop;
    for (int j = 0; j < hop; ++j) {
        const int p = beg + j;
        if (p < L) s += xr[p];
    }
    out[i] = s / (float)hop;
}

// avd_tune_set "codec_mfma" (AVD_CODEC_MFMA): 1 (default) = 64 -> 64 layers with k = 7 / 9 on the fp32 matrix pipe; 0 = the vector kernel
int g_codec_mfma = getenv("AVD_CODEC_MFMA") ? atoi(getenv("AVD_CODEC_MFMA")) : 1;
int conv1d_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int Lin, int up,
               int k, int act, hipStream_t st) {
    AVD_REQUIRE(x && w && out, AVD_EINVAL, "conv1d: null pointer");
    AVD_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Lin > 0 && up > 0, AVD_EINVAL, "conv1d: bad dims");
    AVD_REQUIRE(k >= 1 && k <= C1_KMAX && (k & 1), AVD_EUNSUPPORTED, "conv1d: odd kernel size 1..%d supported (got %d)", C1_KMAX, k);
    AVD_REQUIRE(act == AVD_ACT_NONE || act == AVD_ACT_GELU || act == AVD_ACT_TANH, AVD_EINVAL, "conv1d: bad act %d", act);
    AVD_REQUIRE(B <= 65535 && (Cout + C1_OC - 1) / C1_OC <= 65535, AVD_EUNSUPPORTED, "conv1d: grid too large");
    const int64_t Lout = (int64_t)Lin * up;
    AVD_REQUIRE(Lout < (1ll << 31), AVD_EUNSUPPORTED, "conv1d: output too long");
    if (Cout == 64 && Cin % 16 == 0 && (k == 7 || k == 9) && g_codec_mfma) {
        static const int tag = prof_tag_id("conv1d_mfma_kernel");
        ProfScope prof(tag, 2.0 * (double)B * Lout * Cin * Cout * k, st);
        const dim3 grid((unsigned)((Lout + 255) / 256), 1, B);
        if (k == 7) hipLaunchKernelGGL(conv1d_mfma_kernel<7>, grid, dim3(256), 0, st, x, w, bias, out, Cin, Lin, up, act);
        else hipLaunchKernelGGL(conv1d_mfma_kernel<9>, grid, dim3(256), 0, st, x, w, bias, out, Cin, Lin, up, act);
        AVD_CHECK_LAUNCH("conv1d (mfma)");
        return AVD_OK;
    }
    static const int tag = prof_tag_id("conv1d_ncl_kernel");
    ProfScope prof(tag, 2.0 * (double)B * Lout * Cin * Cout * k, st);
    hipLaunchKernelGGL(conv1d_ncl_kernel, dim3((unsigned)((Lout + C1_TILE - 1) / C1_TILE), (Cout + C1_OC - 1) / C1_OC, B),
                       dim3(C1_TILE), 0, st, x, w, bias, out, Cin, Cout, Lin, up, k, act);
    AVD_CHECK_LAUNCH("conv1d");
    return AVD_OK;
}

}  // namespace avd

using namespace avd;

extern "C" int avd_conv1d_act_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                                  int Lin, int upsample, int k, int act, avd_stream_t stream) {
    return conv1d_f32(x, w, bias, out, B, Cin, Cout, Lin, upsample, k, act, static_cast<hipStream_t>(stream));
}

extern "C" int avd_avgpool_frames_f32(const float* x, float* out, int rows, int L, int Fa, int hop, avd_stream_t stream) {
    AVD_REQUIRE(x && out && rows > 0 && L > 0 && Fa > 0 && hop > 0, AVD_EINVAL, "avgpool_frames: bad arguments");
    const int64_t total = (int64_t)rows * Fa;
    hipLaunchKernelGGL(avgpool_frames_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, out, L, Fa, hop, total);
    AVD_CHECK_LAUNCH("avgpool_frames");
    return AVD_OK;
}
