// Sliding-window stitching for long-form generation:
//   avdiff/models/infer/stream_infer.py:85-116 (crossfade_audio), :119-143 (crossfade_video).
// Weighted overlap-add of N windows of length L placed every `hop` positions, divided by the summed weights
// (clamped at 1e-6).  Gather form, windows accumulated in increasing order with separately rounded multiply and add
// (no FMA contraction), so fp32 results are bit-identical to the reference's numpy loop.  HBM-bound, one pass.
#include "avd_common.h"

namespace avd {

__device__ __forceinline__ void window_range(int64_t p, int L, int hop, int N, int& lo, int& hi) {
    hi = (int)(p / hop);
    if (hi > N - 1) hi = N - 1;
    const int64_t q = p - L + 1;
    lo = q <= 0 ? 0 : (int)((q + hop - 1) / hop);
}

// chunks [N, L, inner] fp32, w [L], out [(N-1)*hop + L, inner]
__global__ __launch_bounds__(256) void crossfade_f32_kernel(const float* __restrict__ chunks, const float* __restrict__ w,
                                                            float* __restrict__ out, int N, int L, int hop, int64_t inner,
                                                            int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t p = i / inner, j = i % inner;
    int lo, hi;
    window_range(p, L, hop, N, lo, hi);
    float acc = 0.f, nrm = 0.f;
    for (int k = lo; k <= hi; ++k) {
        const int q = (int)(p - (int64_t)k * hop);
        const float wq = w[q];
        acc = __fadd_rn(acc, __fmul_rn(chunks[((int64_t)k * L + q) * inner + j], wq));
        nrm = __fadd_rn(nrm, wq);
    }
    out[i] = __fdiv_rn(acc, fmaxf(nrm, 1e-6f));
}

// chunks [N, L, inner] uint8 (inner = H*W*3), w [L], out uint8: (clip(sum(c/255 * w) / sum(w), 0, 1) * 255) truncated
__global__ __launch_bounds__(256) void crossfade_u8_kernel(const uint8_t* __restrict__ chunks, const float* __restrict__ w,
                                                           uint8_t* __restrict__ out, int N, int L, int hop, int64_t inner,
                                                           int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int64_t p = i / inner, j = i % inner;
    int lo, hi;
    window_range(p, L, hop, N, lo, hi);
    float acc = 0.f, nrm = 0.f;
    for (int k = lo; k <= hi; ++k) {
        const int q = (int)(p - (int64_t)k * hop);
        const float wq = w[q];
        const float c = __fdiv_rn((float)chunks[((int64_t)k * L + q) * inner + j], 255.0f);
        acc = __fadd_rn(acc, __fmul_rn(c, wq));
        nrm = __fadd_rn(nrm, wq);
    }
    float v = __fdiv_rn(acc, fmaxf(nrm, 1e-6f));
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = (uint8_t)__fmul_rn(v, 255.0f);
}

}  // namespace avd

using namespace avd;

extern "C" int avd_crossfade_f32(const float* chunks, const float* w, float* out, int N, int L, int hop, int64_t inner,
                                 avd_stream_t stream) {
    AVD_REQUIRE(chunks && w && out && N > 0 && L > 0 && hop > 0 && inner > 0, AVD_EINVAL, "crossfade: bad arguments");
    const int64_t total = ((int64_t)(N - 1) * hop + L) * inner;
    static const int tag = prof_tag_id("crossfade_f32_kernel");
    ProfScope prof(tag, 4.0 * ((double)N * L * inner + (double)total), static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(crossfade_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), chunks, w, out, N, L, hop, inner, total);
    AVD_CHECK_LAUNCH("crossfade_f32");
    return AVD_OK;
}

extern "C" int avd_crossfade_u8(const uint8_t* chunks, const float* w, uint8_t* out, int N, int L, int hop, int64_t inner,
                                avd_stream_t stream) {
    AVD_REQUIRE(chunks && w && out && N > 0 && L > 0 && hop > 0 && inner > 0, AVD_EINVAL, "crossfade: bad arguments");
    const int64_t total = ((int64_t)(N - 1) * hop + L) * inner;
    static const int tag = prof_tag_id("crossfade_u8_kernel");
    ProfScope prof(tag, (double)N * L * inner + (double)total, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(crossfade_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), chunks, w, out, N, L, hop, inner, total);
    AVD_CHECK_LAUNCH("crossfade_u8");
    return AVD_OK;
}
