// AudioCodec on MI355X — the audio end of the loop boundary:
//   avdiff/models/encoders/audio_codec.py:88-133 (layers), :158-182 (_avgpool_frames), :184-199 (encode), :200-214 (decode).
// 97 k parameters, ~2 GFLOP per 3-second clip, run once per sample: a direct NCL conv1d on the vector ALUs with the
// input tile in LDS is ample (the matrix cores would sit idle behind the 1-channel ends).  One thread owns one output
// position and 16 output channels; weights are wave-uniform scalar loads; the nearest-neighbour x hop upsample of the
// decoder is folded into the first smoothing conv's input indexing, so the 64 x 48,000 upsampled signal is never stored.
#include "avd_common.h"

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

int conv1d_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, int Lin, int up,
               int k, int act, hipStream_t st) {
    AVD_REQUIRE(x && w && out, AVD_EINVAL, "conv1d: null pointer");
    AVD_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Lin > 0 && up > 0, AVD_EINVAL, "conv1d: bad dims");
    AVD_REQUIRE(k >= 1 && k <= C1_KMAX && (k & 1), AVD_EUNSUPPORTED, "conv1d: odd kernel size 1..%d supported (got %d)", C1_KMAX, k);
    AVD_REQUIRE(act == AVD_ACT_NONE || act == AVD_ACT_GELU || act == AVD_ACT_TANH, AVD_EINVAL, "conv1d: bad act %d", act);
    AVD_REQUIRE(B <= 65535 && (Cout + C1_OC - 1) / C1_OC <= 65535, AVD_EUNSUPPORTED, "conv1d: grid too large");
    const int64_t Lout = (int64_t)Lin * up;
    AVD_REQUIRE(Lout < (1ll << 31), AVD_EUNSUPPORTED, "conv1d: output too long");
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
