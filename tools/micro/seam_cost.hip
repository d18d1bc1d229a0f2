// Calibration (round 5): what a seam between two dependent stages costs on this chip — as a kernel boundary (back-to-back launches
// on one stream) against a grid-wide barrier inside one persistent launch (one block per CU, arrive on a global counter + spin; the cheapest
// correct form: one atomic per block, a generation number, acquire / release fences).  The small-batch step (C1: 58 dependent launches of
// ~11 us) could only gain from fusing its chain into one launch if the second were cheaper than the first.
//   hipcc -O3 --offload-arch=gfx950 -o seam_cost seam_cost.hip && ./seam_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void stage_kernel(float* buf, int n) {          // a minimal dependent stage: every block reads what the previous launch wrote
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = buf[(i + 64) % n] * 1.0001f + 1.0f;
}

__global__ void persistent_kernel(float* buf, int n, unsigned* counter, int stages) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int s = 1; s <= stages; ++s) {
        if (i < n) buf[i] = buf[(i + 64) % n] * 1.0001f + 1.0f;
        __threadfence();                                    // release this block's writes
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)s * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int blocks = p.multiProcessorCount, threads = 256, n = blocks * threads, stages = 2000;
    float* buf; unsigned* counter;
    CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4)); CK(hipMalloc(&counter, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int s = 0; s < stages; ++s) stage_kernel<<<blocks, threads>>>(buf, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms_l; CK(hipEventElapsedTime(&ms_l, e0, e1));
        CK(hipMemset(counter, 0, 4));
        CK(hipEventRecord(e0));
        persistent_kernel<<<blocks, threads>>>(buf, n, counter, stages);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms_p; CK(hipEventElapsedTime(&ms_p, e0, e1));
        printf("%d CUs, %d dependent stages of one block per CU: %7.3f us per stage as separate launches | %7.3f us per stage behind a grid-wide barrier in one launch\n",
               blocks, stages, ms_l * 1e3 / stages, ms_p * 1e3 / stages);
    }
    // the same with a HIP graph of the launches (what DenoiseEngine.run(graph=True) replays)
    hipStream_t st; CK(hipStreamCreate(&st));
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int s = 0; s < 200; ++s) stage_kernel<<<blocks, threads, 0, st>>>(buf, n);
    CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms_g; CK(hipEventElapsedTime(&ms_g, e0, e1));
    printf("as a captured graph of 200 launches, replayed: %7.3f us per stage\n", ms_g * 1e3 / 2000);
    return 0;
}
