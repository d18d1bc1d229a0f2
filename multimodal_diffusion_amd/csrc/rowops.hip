// Row-wise HBM-bound ops: RMSNorm (avdiff/models/mmdt.py:33-42) and LayerNorm+activation
// (avdiff/models/heads/noise_heads.py:141-147).
//
// One wave per row: 16-byte coalesced loads, the row stays in registers between the reduction and the
// normalise pass (d <= 2048), wavefront-shuffle (xor butterfly) reductions, no LDS, no atomics.
// Algorithmic traffic: read d + write d floats per row (8·d bytes/row) — the HBM roofline for these kernels.
#include "avd_common.h"

namespace avd {

constexpr int ROW_MAXV = 8;   // float4 per lane kept in registers -> d <= 2048

template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* x, RowMap xm,   // y may alias x (in-place)
                                                      const float* __restrict__ scale, float* y,
                                                      RowMap ym, int64_t rows, int d, float eps, float sqrt_d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + xm.off(row);
    f32x4 v[NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
        }
    }
    ss = wave_sum(ss);
    // reference: norm_x = ||x|| / sqrt(d); y = scale * x / (norm_x + eps)   (eps outside the sqrt)
    const float den = sqrtf(ss) / sqrt_d + eps;
    float* yr = y + ym.off(row);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c);
            f32x4 o = {s[0] * v[i][0] / den, s[1] * v[i][1] / den, s[2] * v[i][2] / den, s[3] * v[i][3] / den};
            *reinterpret_cast<f32x4*>(yr + c) = o;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_act_kernel(const float* x,          // y may alias x (in-place)
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* y,
                                                            int64_t rows, int d, float eps, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * d;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c);
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = v[i][e] - mean;
                q += t * t;
            }
        }
    }
    const float var = wave_sum(q) / (float)d;      // biased, as torch.nn.LayerNorm
    const float rstd = 1.0f / sqrtf(var + eps);
    float* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[i][e] - mean) * rstd * gm[e] + bt[e];
                if (act == AVD_ACT_GELU) t = gelu_erf(t);
                else if (act == AVD_ACT_SILU) t = silu(t);
                else if (act == AVD_ACT_RELU) t = fmaxf(t, 0.f);
                else if (act == AVD_ACT_LEAKY_RELU) t = t > 0.f ? t : 0.1f * t;       // nn.LeakyReLU(0.1), noise_heads.py:34
                o[e] = t;
            }
            *reinterpret_cast<f32x4*>(yr + c) = o;
        }
    }
}

static int nv_for(int d) { return (d + 255) / 256; }

int rmsnorm_f32(const float* x, RowMap xm, const float* scale, float* y, RowMap ym, int64_t rows, int d, float eps,
                hipStream_t st) {
    AVD_REQUIRE(x && scale && y, AVD_EINVAL, "rmsnorm: null pointer");
    AVD_REQUIRE(rows >= 0 && d > 0, AVD_EINVAL, "rmsnorm: bad dims");
    AVD_REQUIRE(d % 4 == 0 && d <= 256 * ROW_MAXV, AVD_EUNSUPPORTED, "rmsnorm: d=%d must be a multiple of 4, <= %d", d,
                256 * ROW_MAXV);
    AVD_REQUIRE(xm.ld % 4 == 0 && ym.ld % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(scale), AVD_EUNSUPPORTED,
                "rmsnorm: rows must be 16-byte aligned");
    if (rows == 0) return AVD_OK;
    const int tag = prof_tag_id("rmsnorm_kernel<%d>", nv_for(d));
    ProfScope prof(tag, 8.0 * (double)rows * d, st);
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const float isd = (float)sqrt((double)d);   // the reference divides by math.sqrt(d) rounded to fp32
    switch (nv_for(d)) {
#define AVD_CASE(NV) case NV: hipLaunchKernelGGL(rmsnorm_kernel<NV>, dim3(grid), dim3(256), 0, st, x, xm, scale, y, ym, rows, d, eps, isd); break;
        AVD_CASE(1) AVD_CASE(2) AVD_CASE(3) AVD_CASE(4) AVD_CASE(5) AVD_CASE(6) AVD_CASE(7) AVD_CASE(8)
#undef AVD_CASE
    }
    AVD_CHECK_LAUNCH("rmsnorm");
    return AVD_OK;
}

// sum of squares of every row: ss[row] (the one-column partial table a folded GEMM reads for the first block's norm1)
__global__ __launch_bounds__(256) void rowss_kernel(const float* __restrict__ x, float* __restrict__ ss, int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * d;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    if (lane == 0) ss[row] = s;
}

int rowss_f32(const float* x, float* ss, int64_t rows, int d, hipStream_t st) {
    AVD_REQUIRE(x && ss && rows > 0 && d > 0 && d % 4 == 0 && aligned16(x), AVD_EINVAL, "rowss: bad arguments");
    static const int tag = prof_tag_id("rowss_kernel");
    ProfScope prof(tag, 4.0 * (double)rows * d, st);
    hipLaunchKernelGGL(rowss_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, ss, rows, d);
    AVD_CHECK_LAUNCH("rowss");
    return AVD_OK;
}

// max |x| and max row 2-norm of a [rows][d] matrix -> out[0], out[1] (the weight bounds the f16x2 image scales are derived from).
// One wave per row; the two maxima are folded with integer atomicMax on the bit patterns (both are non-negative floats), so the
// result is exact and order-independent.  NaN anywhere makes both outputs NaN (the caller refuses non-finite bounds).
__global__ __launch_bounds__(256) void bounds_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * d;
    float s = 0.f, m = 0.f;
    bool bad = false;
    for (int c = lane; c < d; c += 64) {
        const float v = xr[c];
        s += v * v;
        m = fmaxf(m, fabsf(v));
        bad = bad || v != v;
    }
    s = wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    bad = __any(bad);
    if (lane == 0) {
        const float nrm = bad ? __builtin_nanf("") : sqrtf(s);
        const float mx = bad ? __builtin_nanf("") : m;
        // NaN's bit pattern (0x7fc00000) is above every finite non-negative float and +inf: it wins the integer maximum
        atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(mx));
        atomicMax(reinterpret_cast<unsigned int*>(out) + 1, __float_as_uint(nrm));
    }
}
__global__ void bounds_init_kernel(float* out) { out[threadIdx.x] = 0.f; }

int weight_bounds_f32(const float* x, int64_t rows, int d, float* out2, hipStream_t st) {
    AVD_REQUIRE(x && out2 && rows > 0 && d > 0, AVD_EINVAL, "weight_bounds: bad arguments");
    hipLaunchKernelGGL(bounds_init_kernel, dim3(1), dim3(2), 0, st, out2);
    hipLaunchKernelGGL(bounds_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, out2, rows, d);
    AVD_CHECK_LAUNCH("weight_bounds");
    return AVD_OK;
}

int layernorm_act_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int d, float eps,
                      int act, hipStream_t st) {
    AVD_REQUIRE(x && gamma && beta && y, AVD_EINVAL, "layernorm: null pointer");
    AVD_REQUIRE(rows >= 0 && d > 0, AVD_EINVAL, "layernorm: bad dims");
    AVD_REQUIRE(d % 4 == 0 && d <= 256 * ROW_MAXV, AVD_EUNSUPPORTED, "layernorm: d=%d must be a multiple of 4, <= %d",
                d, 256 * ROW_MAXV);
    AVD_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta), AVD_EUNSUPPORTED,
                "layernorm: pointers must be 16-byte aligned");
    if (rows == 0) return AVD_OK;
    const int tag = prof_tag_id("layernorm_act_kernel<%d>", nv_for(d));
    ProfScope prof(tag, 8.0 * (double)rows * d, st);
    const unsigned grid = (unsigned)((rows + 3) / 4);
    switch (nv_for(d)) {
#define AVD_CASE(NV) case NV: hipLaunchKernelGGL(layernorm_act_kernel<NV>, dim3(grid), dim3(256), 0, st, x, gamma, beta, y, rows, d, eps, act); break;
        AVD_CASE(1) AVD_CASE(2) AVD_CASE(3) AVD_CASE(4) AVD_CASE(5) AVD_CASE(6) AVD_CASE(7) AVD_CASE(8)
#undef AVD_CASE
    }
    AVD_CHECK_LAUNCH("layernorm_act");
    return AVD_OK;
}

}  // namespace avd

extern "C" int avd_rmsnorm_f32(const float* x, const float* scale, float* y, int64_t rows, int d, float eps,
                               avd_stream_t stream) {
    return avd::rmsnorm_f32(x, avd::RowMap{d, 0, 0}, scale, y, avd::RowMap{d, 0, 0}, rows, d, eps,
                            static_cast<hipStream_t>(stream));
}

extern "C" int avd_weight_bounds_f32(const float* w, int64_t rows, int cols, float* out2, avd_stream_t stream) {
    return avd::weight_bounds_f32(w, rows, cols, out2, static_cast<hipStream_t>(stream));
}

extern "C" int avd_layernorm_act_f32(const float* x, const float* gamma, const float* beta, float* y, int64_t rows,
                                     int d, float eps, int act, avd_stream_t stream) {
    return avd::layernorm_act_f32(x, gamma, beta, y, rows, d, eps, act, static_cast<hipStream_t>(stream));
}
