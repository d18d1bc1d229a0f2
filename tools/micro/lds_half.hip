// Calibration: what does an EXEC-masked half-wave ds_read_b128 cost next to MFMAs, and what would a fragment scheme built on it buy?
//
// The two-terms-per-MFMA bf16x3 GEMMs (csrc/gemm_bf16x3.hip, v_mfma_f32_16x16x32_bf16) read five fragment kinds per 16-k step:
//   A [h|l] [h|m]     B [l|h] [m|h] [h|m]          ([X|Y] = lanes 0-31 carry plane X of the 16 k, lanes 32-63 plane Y)
// = 40 ds_read_b128 per 192 MFMAs for a 128 x 128 wave tile.  But [h|l] -> [h|m] changes only the upper 32 lanes, [l|h] -> [m|h] only the
// lower 32, and [m|h] -> ([m|m], [h|h]) is one v_mov + one v_permlane32_swap per dword — with hh + mm regrouped as
// [h|m] x [h|h] + [h|m] x [m|m] = (hh + mh) + (hm + mm).  That is 16 full + 16 HALF reads per step: 24 read-equivalents, IF the LDS
// skips the lane groups EXEC masks off (a ds_read_b128 is served in four 16-lane groups, two per wave half).
// Variants (one wave per SIMD, 256-thread blocks, random operands, results meaningless):
//   0  40 full reads per step (today)         1  16 full + 16 half reads + 32 (mov, swap) pairs (the scheme)
//   2  24 full reads (what the scheme costs if a half read is half a read)      3  32 full reads (... if it is a whole one)
//   4  no reads (register operands: the ceiling)
//   hipcc -O3 --offload-arch=gfx950 -o lds_half lds_half.hip && ./lds_half
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

// four half-wave reads (lanes 32-63 when UPPER, else lanes 0-31) into the live registers of four fragments; offsets are immediates
template <bool UPPER, int O0, int O1, int O2, int O3>
__device__ __forceinline__ void ld_half4(bf16x8& f0, bf16x8& f1, bf16x8& f2, bf16x8& f3, unsigned addr) {
    const unsigned long long m = UPPER ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    asm volatile("s_mov_b64 exec, %5\n\t"
                 "ds_read_b128 %0, %4 offset:%c6\n\t"
                 "ds_read_b128 %1, %4 offset:%c7\n\t"
                 "ds_read_b128 %2, %4 offset:%c8\n\t"
                 "ds_read_b128 %3, %4 offset:%c9\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
                 : "v"(addr), "s"(m), "i"(O0), "i"(O1), "i"(O2), "i"(O3)
                 : "memory");
}

template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, const unsigned short* __restrict__ rnd, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[32768];       // 64 KiB of random bf16
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = rnd[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane;        // conflict-free: consecutive lanes, consecutive 16-byte chunks
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const void*)base;
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a[8], b[8], b2[8];
    for (int i = 0; i < 8; ++i) { a[i] = base[64 * i]; b[i] = base[64 * (8 + i)]; b2[i] = base[64 * (16 + i)]; }
    auto mm = [&](const bf16x8 (&A_)[8], const bf16x8 (&B_)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B_[j], A_[i], acc[i][j & 3], 0, 0, 0);
    };
    auto ld8 = [&](bf16x8 (&dst)[8], int slot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = base[64 * ((slot * 8 + i) & 31) + 2048 * (slot & 1)];
    };
    for (int it = 0; it < iters; ++it) {
        if constexpr (VAR == 0) {              // 40 full reads: A [h|l], [h|m]; B [l|h], [m|h], [h|m]
            ld8(a, it); ld8(b, it + 1); SB();
            mm(a, b); SB();
            ld8(a, it + 2); ld8(b, it + 3); SB();
            mm(a, b); SB();
            ld8(b, it + 4); SB();
            mm(a, b); SB();
        } else if constexpr (VAR == 1) {       // 16 full + 16 half reads, swaps for the third group
            ld8(a, it); ld8(b, it + 1); SB();
            mm(a, b); SB();
            ld_half4<true, 0, 1024, 2048, 3072>(a[0], a[1], a[2], a[3], addr + ((it & 7) << 12));
            ld_half4<true, 4096, 5120, 6144, 7168>(a[4], a[5], a[6], a[7], addr + ((it & 7) << 12));
            ld_half4<false, 8192, 9216, 10240, 11264>(b[0], b[1], b[2], b[3], addr + ((it & 7) << 12));
            ld_half4<false, 12288, 13312, 14336, 15360>(b[4], b[5], b[6], b[7], addr + ((it & 7) << 12));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SB();
            mm(a, b); SB();
            // [m|h] -> [m|m] (in b) and [h|h] (in b2): one copy + one half exchange per dword
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                u32x4 x = __builtin_bit_cast(u32x4, b[j]), y = x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(x[e], y[e], false, false);
                    x[e] = sw[0];
                    y[e] = sw[1];
                }
                b[j] = __builtin_bit_cast(bf16x8, x);
                b2[j] = __builtin_bit_cast(bf16x8, y);
            }
            SB();
            // (hh + mh) + (hm + mm): two half-size groups in the real kernel; here the same 64 MFMAs
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((i & 1) ? b[j] : b2[j], a[i], acc[i][j & 3], 0, 0, 0);
            SB();
        } else if constexpr (VAR == 2) {       // 24 full reads
            ld8(a, it); ld8(b, it + 1); SB();
            mm(a, b); SB();
            ld8(a, it + 2); SB();
            mm(a, b); SB();
            mm(a, b); SB();
        } else if constexpr (VAR == 3) {       // 32 full reads
            ld8(a, it); ld8(b, it + 1); SB();
            mm(a, b); SB();
            ld8(a, it + 2); ld8(b, it + 3); SB();
            mm(a, b); SB();
            mm(a, b); SB();
        } else {                               // registers only
            mm(a, b); SB();
            mm(a, b2); SB();
            mm(a, b); SB();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VAR> static int run(float* out, const unsigned short* rnd, int iters, int blocks, const char* what) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 4; ++w) k<VAR><<<blocks, 256>>>(out, rnd, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int w = 0; w < 3; ++w) k<VAR><<<blocks, 256>>>(out, rnd, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double fl = 3.0 * blocks * 4.0 * iters * 192.0 * 16384.0;
    printf("variant %d  %-58s %8.2f ms  %7.1f TFLOP/s bf16\n", VAR, what, ms, fl / ms / 1e9);
    return 0;
}

int main() {
    float* out; unsigned short* rnd;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&rnd, 65536 * 2));
    unsigned short* h = (unsigned short*)malloc(65536 * 2);
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (unsigned short)(((rand() & 1) << 15) | ((124 + (rand() & 3)) << 7) | (rand() & 127));
    CK(hipMemcpy(rnd, h, 65536 * 2, hipMemcpyHostToDevice));
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        if (run<0>(out, rnd, iters, 256, "40 full reads / 192 MFMAs (today)")) return 1;
        if (run<1>(out, rnd, iters, 256, "16 full + 16 half-wave reads + 32 swaps (the scheme)")) return 1;
        if (run<2>(out, rnd, iters, 256, "24 full reads (scheme, if a half read costs half)")) return 1;
        if (run<3>(out, rnd, iters, 256, "32 full reads (scheme, if a half read costs a whole one)")) return 1;
        if (run<4>(out, rnd, iters, 256, "no reads (register operands)")) return 1;
    }
    return 0;
}
