// Calibration: on ONE SIMD, does a wave of bf16 MFMAs (v_mfma_f32_32x32x16_bf16) run beside a wave of vector-ALU work
// (softmax-like mix: v_sub / v_exp / v_add / v_cvt_pk_bf16 / v_max)?   512-thread blocks = two waves per SIMD: waves 0-3 MFMA, waves 4-7 VALU
// (or swapped: `swap` = the OLDER waves take the VALU role).  mode 1 = MFMA waves only, 2 = VALU only, 3 = both; prio: s_setprio of the MFMA waves.
//   hipcc -O3 --offload-arch=gfx950 -o coissue16 coissue16.hip && ./coissue16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512) void co_kernel(float* out, const float* __restrict__ tbl, int mode, int it_mfma, int it_valu, int swap, int prio) {
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = swap ? wave >= 4 : wave < 4;
    float s = 0.f;
    if (mode == 4) {        // ONE stream per wave: 4 MFMAs, then ~28 VALU instructions, repeated (both waves of a SIMD run it)
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(tbl[e] + threadIdx.x * 1e-3f); b[e] = (__bf16)(tbl[8 + e] - threadIdx.x * 1e-3f); }
        float c[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) c[i] = tbl[i] * 1e-3f;
        float m = tbl[3];
        for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                if (it_valu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float e = __builtin_amdgcn_exp2f(c[4 * u + i] - m);
                        c[4 * u + i] = e * 0.5f + 1e-3f;
                        m = fmaxf(m * 0.999f, e * 1e-3f);
                    }
                }
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int i = 0; i < 32; ++i) s += c[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + m;
        return;
    }
    if (mfma_role) {
        if (!(mode & 1)) return;
        if (prio) __builtin_amdgcn_s_setprio(3);
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(tbl[e] + threadIdx.x * 1e-3f); b[e] = (__bf16)(tbl[8 + e] - threadIdx.x * 1e-3f); }
        for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        if (!(mode & 2)) return;
        float c[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) c[i] = tbl[i] * 1e-3f;
        float m = tbl[3];
        unsigned pk = 0;
        for (int it = 0; it < it_valu; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {                  // per element: sub, exp, add, max  (+ one cvt_pk per pair)
#ifdef NO_EXP
                const float e = (c[i] - m) * 1.0001f;
#else
                const float e = __builtin_amdgcn_exp2f(c[i] - m);
#endif
                c[i] = e * 0.5f + 1e-3f;
                m = fmaxf(m * 0.999f, e * 1e-3f);
            }
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                const f32x2 v = {c[i], c[i + 1]};
                pk ^= __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            }
        }
        for (int i = 0; i < 32; ++i) s += c[i];
        s += (float)(pk & 0xff) + m;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int run(int mode, int it_mfma, int it_valu, int swap, int prio, float* out, const float* tbl) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    co_kernel<<<256, 512>>>(out, tbl, mode, it_mfma / 2, it_valu / 2, swap, prio);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    co_kernel<<<256, 512>>>(out, tbl, mode, it_mfma, it_valu, swap, prio);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double f_m = (mode & 1) ? 256.0 * 4 * it_mfma * 32.0 * 32768.0 : mode == 4 ? 256.0 * 8 * it_mfma * 32.0 * 32768.0 : 0.0;
    printf("mode %d swap %d prio %d  mfma iters %6d valu iters %6d: %8.3f ms   mfma %7.1f TF\n", mode, swap, prio, it_mfma, it_valu, ms, f_m / ms / 1e9);
    return 0;
}

int main() {
    float *out, *tbl; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&tbl, 64 * 4));
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.f + 1e-2f * i;
    CK(hipMemcpy(tbl, h, sizeof(h), hipMemcpyHostToDevice));
    // MFMA wave: 32 MFMAs x 32 cyc = 1,024 cyc per iteration.  VALU wave: ~150 instructions per iteration
    const int im = 20000;
    // mode 4: both waves of a SIMD run 32 MFMAs + (it_valu ? 32 x (sub exp fma mul mul max) : nothing) per iteration, interleaved 4 : 24
    if (run(4, im / 2, 0, 0, 0, out, tbl)) return 1;
    if (run(4, im / 2, 1, 0, 0, out, tbl)) return 1;
    for (int rep = 0; rep < 1; ++rep) {
        if (run(1, im, 0, 0, 0, out, tbl)) return 1;
        for (int iv : {im / 2, im, 2 * im}) if (run(2, 0, iv, 0, 0, out, tbl)) return 1;
        for (int swap = 0; swap < 2; ++swap)
            for (int prio = 0; prio < 2; ++prio)
                for (int iv : {im / 2, im, 2 * im}) if (run(3, im, iv, swap, prio, out, tbl)) return 1;
    }
    return 0;
}
