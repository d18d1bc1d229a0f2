// Calibration: what does a register-only v_mfma_f32_32x32x2_f32 loop sustain on this device?
// (operands in registers, 4 independent accumulators, W waves per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
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
void run(int blocks, int threads, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, 1.0001f, 0.9999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * 16.0 * NACC * 4096.0;
    printf("NACC=%d blocks=%d threads=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks, threads, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<4>(256, 256, 2000);      // 1 wave / SIMD
    run<4>(512, 256, 2000);      // 2 waves / SIMD
    run<1>(512, 256, 8000);      // single dependent chain, 2 waves/SIMD
    run<1>(256, 256, 8000);      // single dependent chain, 1 wave/SIMD
    run<4>(1024, 256, 1000);
    return 0;
}
