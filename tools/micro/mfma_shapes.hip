// Calibration: sustained bf16 MFMA rate of the two shapes under the chip's power management, register operands, random data.
// Same FLOPs per wave either way: 32 x v_mfma_f32_32x32x16_bf16 (32 cycles each) vs 128 x v_mfma_f32_16x16x32_bf16 (8 passes... 16 cycles each, 4x fewer FLOPs)
//   hipcc -O3 --offload-arch=gfx950 -o mfma_shapes mfma_shapes.hip && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SHAPE, int WPS>     // SHAPE 32: 32x32x16, 16: 16x16x32; WPS waves per SIMD (block = 256 * WPS threads)
__global__ __launch_bounds__(256 * WPS) void k(float* out, const unsigned short* __restrict__ rnd, int iters) {
    bf16x8 a[4], b[4];
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 8; ++e) {
            a[q][e] = __builtin_bit_cast(__bf16, rnd[(threadIdx.x * 64 + q * 8 + e) & 65535]);
            b[q][e] = __builtin_bit_cast(__bf16, rnd[(threadIdx.x * 64 + 32 + q * 8 + e) & 65535]);
        }
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)                                     // 2 x 32 MFMAs of 16x16x32 = the FLOPs of 4 x 8 of 32x32x16 ... x2 below
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3], b[(i >> 3) & 3], acc[i], 0, 0, 0);
#pragma unroll
            for (int u = 2; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3], b[(i >> 3) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The same comparison with every operand fragment re-read from LDS inside the loop, in the proportions of the split GEMM's wave tile
// (128 x 64, one 16-k step of six product terms): 32x32x16 -> 18 ds_read_b128 + 48 MFMAs;  16x16x32 with two terms per MFMA
// (K = 32 = [plane a | plane b] of the same 16 k) -> 28 ds_read_b128 + 96 MFMAs.  Results are meaningless; rates are not.
template <int SHAPE, int WPS>
__global__ __launch_bounds__(256 * WPS) void kl(float* out, const unsigned short* __restrict__ rnd, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[16384];       // 32 KiB of random bf16
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = rnd[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane;       // conflict-free: consecutive lanes, consecutive 16-byte chunks
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            bf16x8 a[4][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[i][p] = base[64 * ((i * 3 + p + it) & 15)];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = base[64 * ((12 + j * 3 + p + it) & 15) + 1024];
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[t]], b[j][PB[t]], acc[i * 2 + j], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            bf16x8 a[8][2], b[4][3];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p) a[i][p] = base[64 * ((i * 2 + p + it) & 15)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = base[64 * ((j * 3 + p + it) & 15) + 1024];
            constexpr int PA[3] = {0, 0, 1}, PB[3] = {0, 1, 2};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[t]], b[j][PB[t]], acc[i * 4 + j], 0, 0, 0);
        }
        for (int i = 0; i < 32; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int WPS> static int runl(float* out, const unsigned short* rnd, int iters, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 6; ++w) kl<SHAPE, WPS><<<blocks, 256 * WPS>>>(out, rnd, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int w = 0; w < 3; ++w) kl<SHAPE, WPS><<<blocks, 256 * WPS>>>(out, rnd, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double mf = SHAPE == 32 ? 48.0 * 32768.0 : 96.0 * 16384.0;       // equal FLOPs per step
    const double fl = 3.0 * blocks * 4.0 * WPS * iters * mf;
    printf("LDS-fed  shape %2d  waves/SIMD %d: %8.2f ms  %7.1f TFLOP/s bf16\n", SHAPE, WPS, ms, fl / ms / 1e9);
    return 0;
}

template <int SHAPE, int WPS> static int run(float* out, const unsigned short* rnd, int iters, int blocks) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 6; ++w) k<SHAPE, WPS><<<blocks, 256 * WPS>>>(out, rnd, iters);       // ~1 s of back-to-back launches first
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int w = 0; w < 3; ++w) k<SHAPE, WPS><<<blocks, 256 * WPS>>>(out, rnd, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // FLOPs per wave per iteration: 32 MFMAs x 32768 (32x32x16) = 128 MFMAs x 8192... (16x16x32: 16*16*32*2 = 16384 -> 128 x 16384 = 2x; so halve iters there)
    const double mf = SHAPE == 32 ? 32.0 * 32768.0 : 128.0 * 16384.0;
    const double fl = 3.0 * blocks * 4.0 * WPS * iters * mf;
    printf("shape %2d  waves/SIMD %d  blocks %4d: %8.2f ms  %7.1f TFLOP/s bf16\n", SHAPE, WPS, blocks, ms, fl / ms / 1e9);
    return 0;
}

int main() {
    float* out; unsigned short* rnd;
    CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&rnd, 65536 * 2));
    unsigned short* h = (unsigned short*)malloc(65536 * 2);
    srand(1);
    for (int i = 0; i < 65536; ++i) {       // random bf16 in [-2, 2): random sign and mantissa, exponent 124..127
        h[i] = (unsigned short)(((rand() & 1) << 15) | ((124 + (rand() & 3)) << 7) | (rand() & 127));
    }
    CK(hipMemcpy(rnd, h, 65536 * 2, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        if (run<32, 1>(out, rnd, 100000, 256)) return 1;
        if (run<16, 1>(out, rnd, 50000, 256)) return 1;
        if (run<32, 2>(out, rnd, 50000, 256)) return 1;
        if (run<16, 2>(out, rnd, 25000, 256)) return 1;
        if (runl<32, 2>(out, rnd, 30000, 256)) return 1;
        if (runl<16, 2>(out, rnd, 30000, 256)) return 1;
    }
    return 0;
}
