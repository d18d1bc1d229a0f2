// Calibration: what does a register-only fp32 MFMA loop sustain on this device, per MFMA shape, over a long run?
// (operands in registers, independent accumulators, 1 or 2 waves per SIMD; >= 100 ms per measurement so the clock settles)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int blocks, int iters, double flop_per_mfma, int nacc) {
    float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<blocks, 256>>>(out, iters / 4, 1.f, 1.f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<<<blocks, 256>>>(out, iters, 1.0001f, 0.9999f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * 4 * iters * 16.0 * nacc * flop_per_mfma;
    printf("%-34s blocks=%4d: %8.2f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    (void)hipFree(out);
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int rep = 0; rep < 2; ++rep) {
        run("32x32x2 f32, 4 acc, 1 wave/SIMD", k32<4>, 256, 60000, 4096.0, 4);
        run("32x32x2 f32, 4 acc, 2 waves/SIMD", k32<4>, 512, 30000, 4096.0, 4);
        run("16x16x4 f32, 8 acc, 1 wave/SIMD", k16<8>, 256, 60000, 2048.0, 8);
        run("16x16x4 f32, 8 acc, 2 waves/SIMD", k16<8>, 512, 30000, 2048.0, 8);
    }
    return 0;
}
