// Calibration: do the fp32 matrix pipe and the fp32 vector ALU of one SIMD run at the same time?
// 512-thread blocks (two waves per SIMD): waves 0-3 run a register-only v_mfma_f32_32x32x2_f32 loop, waves 4-7 a
// register-only v_fma_f32 loop whose multiplier comes from SGPRs.  mode 1 = MFMA waves only, 2 = VALU waves only, 3 = both.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int NV = 32;

__global__ __launch_bounds__(512) void co_kernel(float* out, const float* __restrict__ tbl, int mode, int it_mfma, int it_valu,
                                                 float a0) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = a0 + threadIdx.x * 1e-3f, b = a0 - threadIdx.x * 1e-3f;
        for (int it = 0; it < it_mfma; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        if (!(mode & 2)) return;
        float c[NV], sb[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { c[i] = 0.f; sb[i] = tbl[i]; }
        float a = a0 + threadIdx.x * 1e-3f;
        for (int it = 0; it < it_valu; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NV; ++i) c[i] = __builtin_fmaf(a, sb[i], c[i]);
        }
        for (int i = 0; i < NV; ++i) s += c[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int run(int mode, int blocks, int it_mfma, int it_valu, float* out, const float* tbl) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    co_kernel<<<blocks, 512>>>(out, tbl, mode, it_mfma / 2, it_valu / 2, 1.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    co_kernel<<<blocks, 512>>>(out, tbl, mode, it_mfma, it_valu, 1.0001f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double f_m = (mode & 1) ? (double)blocks * 4 * it_mfma * 32.0 * 4096.0 : 0.0;
    const double f_v = (mode & 2) ? (double)blocks * 256.0 * it_valu * 8.0 * NV * 2.0 : 0.0;
    printf("mode %d blocks=%4d: %8.2f ms   mfma %7.1f TF   valu %7.1f TF   sum %7.1f TF\n", mode, blocks, ms, f_m / ms / 1e9,
           f_v / ms / 1e9, (f_m + f_v) / ms / 1e9);
    return 0;
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d\n", p.gcnArchName, p.multiProcessorCount);
    float *out, *tbl; CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&tbl, NV * 4));
    float h[NV]; for (int i = 0; i < NV; ++i) h[i] = 1.f + 1e-4f * i;
    CK(hipMemcpy(tbl, h, sizeof(h), hipMemcpyHostToDevice));
    // per iteration: MFMA wave 32 MFMAs x 64 cyc = 2048 cyc; VALU wave 256 v_fma x 2 cyc = 512 cyc -> 4 VALU iterations per MFMA iteration
    const int im = 30000;
    for (int rep = 0; rep < 2; ++rep) {
        for (int blocks : {256, 512}) {
            if (run(1, blocks, im, 0, out, tbl)) return 1;
            if (run(2, blocks, 0, im * 4, out, tbl)) return 1;
            if (run(3, blocks, im, im * 4, out, tbl)) return 1;
            if (run(3, blocks, im, im * 2, out, tbl)) return 1;
            if (run(3, blocks, im, im, out, tbl)) return 1;
        }
    }
    return 0;
}
