// Calibration (round 5): the ENERGY floor of the six-term bf16x3 product stream on this chip.
// Every large kernel of the denoising step runs on the socket power cap (tools/micro/kernel_power.py), so its time is energy / cap and the
// roofline that binds is joules per FLOP, not cycles.  This program runs the bare matrix stream those kernels are built around —
// v_mfma_f32_16x16x32_bf16 from registers, random operands, one wave per SIMD (the 128 x 128 wave tile's occupancy) and two — for ~3 s per
// case while a host thread samples this GPU's shader clock and socket power from sysfs, and prints TFLOP/s, W and pJ per bf16 FLOP; x 6
// terms = pJ per fp32-equivalent FLOP, the floor a six-term GEMM cannot go under at this power cap.  A third case adds the LDS fragment
// reads of the w128 GEMM's k-step (40 ds_read_b128 per 192 MFMAs): what the operand path costs on top.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_energy mfma_energy.hip -lpthread && ./mfma_energy
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <dirent.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WPS, bool LDS>
__global__ __launch_bounds__(256 * WPS) void k(float* out, const unsigned short* __restrict__ rnd, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[LDS ? 32768 : 64];
    if constexpr (LDS) {
        for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = rnd[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[8];
    for (int q = 0; q < 8; ++q)
        for (int e = 0; e < 8; ++e) {
            a[q][e] = __builtin_bit_cast(__bf16, rnd[(threadIdx.x * 128 + q * 8 + e) & 65535]);
            b[q][e] = __builtin_bit_cast(__bf16, rnd[(threadIdx.x * 128 + 64 + q * 8 + e) & 65535]);
        }
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* base = reinterpret_cast<const bf16x8*>(lds) + lane;
    for (int it = 0; it < iters; ++it) {
        if constexpr (LDS) {       // 10 fragment reads per 48 MFMAs = the 40 per 192 of the 128 x 128 wave tile's k-step
#pragma unroll
            for (int q = 0; q < 5; ++q) { a[q] = base[64 * ((q + it) & 31)]; b[q] = base[64 * ((q + 5 + it) & 31) + 2048]; }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[(t & 3) * 8 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + t) & 7], b[(i * 3 + t) & 7], acc[(t & 3) * 8 + i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static std::string g_dir;
static bool find_dir(int dev) {
    char bdf[64];
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), dev) != hipSuccess) return false;
    for (char* p = bdf; *p; ++p) *p = (char)tolower(*p);
    g_dir = std::string("/sys/bus/pci/devices/") + bdf;
    FILE* f = fopen((g_dir + "/pp_dpm_sclk").c_str(), "r");
    if (!f) return false;
    fclose(f);
    return true;
}
static double read_sclk() {
    FILE* f = fopen((g_dir + "/pp_dpm_sclk").c_str(), "r");
    if (!f) return 0;
    char line[128];
    double mhz = 0;
    while (fgets(line, sizeof(line), f)) {
        if (strchr(line, '*')) { const char* c = strchr(line, ':'); if (c) mhz = atof(c + 1); }
    }
    fclose(f);
    return mhz;
}
static double read_watts() {
    std::string hw = g_dir + "/hwmon";
    DIR* d = opendir(hw.c_str());
    if (!d) return 0;
    double w = 0;
    while (dirent* e = readdir(d)) {
        if (strncmp(e->d_name, "hwmon", 5)) continue;
        for (const char* n : {"power1_average", "power1_input"}) {
            FILE* f = fopen((hw + "/" + e->d_name + "/" + n).c_str(), "r");
            if (f) { double v = 0; if (fscanf(f, "%lf", &v) == 1) w = v / 1e6; fclose(f); if (w > 0) break; }
        }
    }
    closedir(d);
    return w;
}
static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int WPS, bool LDS> static int run(const char* name, float* out, const unsigned short* rnd) {
    const int blocks = 256, iters = 20000;
    const double fl_launch = (double)blocks * 4 * WPS * iters * 48.0 * 16384.0;
    std::atomic<bool> stop{false};
    std::vector<double> clk, pw;
    auto t0 = std::chrono::steady_clock::now();
    std::thread th([&] {
        while (!stop) {
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const double c = read_sclk(), w = read_watts();
            if (s > 1.0) { clk.push_back(c); pw.push_back(w); }
            std::this_thread::sleep_for(std::chrono::milliseconds(50));
        }
    });
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int n = 0;
    float ms_tot = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 3.0) {
        CK(hipEventRecord(e0));
        for (int w = 0; w < 4; ++w) k<WPS, LDS><<<blocks, 256 * WPS>>>(out, rnd, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) { ms_tot += ms; n += 4; }
    }
    stop = true;
    th.join();
    const double tf = fl_launch * n / (ms_tot * 1e-3) / 1e12, W = med(pw), C = med(clk);
    printf("%-44s %7.1f TFLOP/s bf16 | sclk %6.0f MHz  socket %6.0f W | %5.3f pJ per bf16 FLOP -> %5.2f pJ per fp32-equivalent FLOP (six terms), %6.1f TFLOP/s fp32-equivalent\n",
           name, tf, C, W, W / tf, 6.0 * W / tf, tf / 6.0);
    return 0;
}

int main() {
    float* out; unsigned short* rnd;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&rnd, 65536 * 2));
    unsigned short* h = (unsigned short*)malloc(65536 * 2);
    srand(1);
    for (int i = 0; i < 65536; ++i) {      // random finite bf16 of moderate magnitude (random sign, exponent near 1, random mantissa)
        const unsigned m = rand() & 0x7f, e = 120 + (rand() & 7), s = rand() & 1;
        h[i] = (unsigned short)((s << 15) | (e << 7) | m);
    }
    CK(hipMemcpy(rnd, h, 65536 * 2, hipMemcpyHostToDevice));
    if (!find_dir(0)) printf("(no sysfs node for this device: clock / power columns read 0)\n");
    if (run<1, false>("registers only, one wave per SIMD", out, rnd)) return 1;
    if (run<2, false>("registers only, two waves per SIMD", out, rnd)) return 1;
    if (run<1, true>("+ 40 ds_read_b128 per 192 MFMAs, one wave/SIMD", out, rnd)) return 1;
    return 0;
}
