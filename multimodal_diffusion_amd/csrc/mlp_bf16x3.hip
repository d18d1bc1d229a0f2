// MLP of a Block (avdiff/models/mmdt.py:77-83, with the residual add of :98) as ONE launch in the six-term bf16x3 mode:
//     y = x + fc2( GELU( fc1( RMSNorm(x) ) ) )        fc1: d -> hidden, fc2: hidden -> d,  d = 512
// The hidden activations never leave the CU: a block owns 128 rows, walks the hidden width in chunks of 64 columns — phase 1 computes the
// chunk H = GELU(rinv * (X W1c^T) + b1) from the un-normalised stream's operand image X and the scale-carrying fc1 image (the folded
// RMSNorm of gemm_bf16x3.hip), splits it into planes and parks it in LDS as four 16-k operand tiles; phase 2 multiplies those tiles
// into the block's 128 x 512 fc2 accumulators (four waves side by side, 128 x 128 each, 256 AGPRs per lane as in gemm_bf16x3_w128_kernel)
// — and ends in the residual + image + sums-of-squares epilogue of the two-launch path.  Same operands and the same six product terms
// in the same order as fc1 (gemm_bf16x3_m16_kernel<3, 4, 8>) followed by fc2 (gemm_bf16x3_w128_kernel<6, RT>): the fp32 stream is
// bit-identical to the two launches' (tests/test_gpu_parity.py::test_fused_mlp_matches_two_launches).
//
// Why it is OFF by default (avd_tune_set "mlp_fused" 1 turns it on; DESIGN.md 4.9 has the numbers): the operand image of 128 rows of X
// is 393 KB — it does not fit the 160 KB LDS beside anything, so X is streamed again for EVERY hidden chunk (32 times for hidden = 2,048)
// and the L2 -> LDS traffic per row is 1.2x that of the two launches, whose 331 MB image round trip per layer it removes; the phase-1 wave
// tile (64 x 32) reads 14 fragments per 24 MFMAs against 28 per 96; and with the fc2 accumulators taking half of every SIMD's registers
// there is one wave per SIMD, so every DMA issue and every phase switch is exposed.  The step runs on the socket power cap: bytes staged
// per FLOP are what it pays for, and this kernel stages more of them.
#include "avd_common.h"
#include "s3_common.h"

#include <stdlib.h>

namespace avd {

constexpr int ML_BM = 128, ML_HC = 64, ML_D = 512;
constexpr int ML_H = 4 * S3_CHUNK;                          // H chunk: four 16-k tiles of [3 planes][128 rows][32 B] = 48 KiB
constexpr int ML_S1 = S3_CHUNK + 3 * ML_HC * 32;            // phase-1 stage: X [3][128][32 B] + W1 chunk [3][64][32 B] = 18 KiB
constexpr int ML_NS1 = 3;                                   // ... ring of three
constexpr int ML_S2 = 3 * ML_D * 32;                        // phase-2 stage: W2 [3 planes][512 rows][32 B] = 48 KiB, ring of two
constexpr int ML_LDS = ML_H + 2 * ML_S2;                    // 144 KiB (the phase-1 ring, 54 KiB, overlays the phase-2 ring)
static_assert(ML_NS1 * ML_S1 <= 2 * ML_S2, "phase-1 ring fits the stage area");

struct MlpArgs {
    const unsigned char* X;      // operand image of the un-normalised stream [M][512]
    const unsigned char* W1;     // image of fc1.weight * norm2.scale [hidden][512]
    const float* b1;             // [hidden]
    const unsigned char* W2;     // image of fc2.weight [512][hidden]
    const float* ss_in;          // sums of squares of X's rows per 64-column chunk, [M][8]
    int hidden;
    float sqrt_d, eps;
    S3Args epi;                  // bias = fc2.bias, R, C, C3 (may be null), ss_out (may be null), M, N = 512: the residual epilogue
};

__global__ __launch_bounds__(256, 1) void mlp_bf16x3_kernel(MlpArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int bm = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4, hiq = kq >> 1;
    const int wm = wave >> 1, wn = wave & 1;                 // phase 1: 2 x 2 waves over the 128 x 64 chunk (64 x 32 each)
    const int64_t M = g.epi.M, m0 = (int64_t)bm * ML_BM;
    const int nchunk = g.hidden / ML_HC, nk2 = g.hidden >> 4;
    constexpr int NK1 = ML_D / 16;
    const unsigned lane16 = (unsigned)lane * 16u;
    unsigned char* stg = smem + ML_H;

    // fragment planes by pair type (see gemm_bf16x3_m16_kernel): lanes of k-group pair 0 carry the first plane, pair 1 the second
    enum { T_HM = 0, T_HL = 1, T_MH = 2, T_LH = 3 };
    auto plane_of = [&](int type, int ps) { return type == T_HM ? hiq * ps : type == T_HL ? 2 * hiq * ps : type == T_MH ? ps - hiq * ps : 2 * ps - 2 * hiq * ps; };
    const int base_e = l15 * 32 + ((kq & 1) << 4), base_o = l15 * 32 + (((kq & 1) ^ 1) << 4);

    // folded RMSNorm: 1 / (rms + eps) of the two rows this lane finishes in the phase-1 epilogue (row tile 2 ip + (kq & 1) of its wave)
    float rinv[2];
#pragma unroll
    for (int ip = 0; ip < 2; ++ip) {
        const int64_t m = m0 + 64 * wm + 16 * (2 * ip + (kq & 1)) + l15;
        rinv[ip] = 1.0f;
        if (m < M) {
            const float* sp = g.ss_in + m * 8;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sp), p1 = *reinterpret_cast<const f32x4*>(sp + 4);
            const f32x4 t = p0 + p1;
            rinv[ip] = 1.0f / (sqrtf((t[0] + t[1]) + (t[2] + t[3])) / g.sqrt_d + g.eps);
        }
    }

    // ---- DMA ----
    // phase 1, k-step kk of chunk c: 12 pieces of X (the block's row group, one contiguous 12 KiB chunk) + 6 of W1 (64 rows of one plane
    // are 2 KiB); five per wave, the two surplus slots repeat the wave's previous piece (same bytes to the same place)
    const unsigned char* xsrc = g.X + (int64_t)bm * NK1 * S3_CHUNK;
    auto issue1 = [&](int c, int kk, int buf) {
        const unsigned char* w1src = g.W1 + ((int64_t)(c >> 1) * NK1 + kk) * S3_CHUNK + (c & 1) * 2048;
        unsigned char* dst = stg + buf * ML_S1;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int P = wave + 4 * i;
            if (P >= 18) P -= 4;
            if (P < 12) {
                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(xsrc + (int64_t)kk * S3_CHUNK + P * 1024 + lane16), AVD_LDS_PTR(dst + P * 1024), 16, 0, 0);
            } else {
                const int pw = P - 12, pl = pw >> 1, part = pw & 1;
                __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(w1src + pl * S3_PLANE + part * 1024 + lane16),
                                                 AVD_LDS_PTR(dst + S3_CHUNK + pl * 2048 + part * 1024), 16, 0, 0);
            }
        }
    };
    // phase 2, hidden k-step kg: W2 rows 0..511 = four row groups, 48 pieces, twelve per wave
    auto issue2 = [&](int kg, int buf) {
        unsigned char* dst = stg + buf * ML_S2;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int P = wave * 12 + i, pl = P >> 4, q = P & 15;
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(g.W2 + ((int64_t)(q >> 2) * nk2 + kg) * S3_CHUNK + pl * S3_PLANE + (q & 3) * 1024 + lane16),
                                             AVD_LDS_PTR(dst + pl * (ML_D * 32) + q * 1024), 16, 0, 0);
        }
    };

    // ---- fc2 accumulators: 128 x 128 per wave in AGPRs (asm MFMAs, see gemm_bf16x3_w128_kernel for why) ----
    f32x4t acc2[2][8][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2[h][i][j] = f32x4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(acc2[h][i][j]));
    asm volatile("s_nop 4");
    // FENCE (the last MFMA group of a chunk's phase 2): the wait states that let the AGPR writes land ride in the same asm
    // statement as the last MFMA.  The register allocator moves accumulator tiles between registers at the phase boundaries and at the
    // loop's exit when it likes, and pads nothing — the MFMAs are opaque to its hazard recogniser; whatever it puts there comes behind them
    // (a plain bool, folded after inlining: asm operands inside a GENERIC lambda do not capture — clang)
    auto mm2 = [&](const bf16x8 (&A_)[8], const bf16x8 (&B_)[8], const bool FENCE) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (FENCE && i == 7 && j == 7)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+a"(acc2[j >> 2][i][j & 3]) : "v"(B_[j]), "v"(A_[i]));
                else
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc2[j >> 2][i][j & 3]) : "v"(B_[j]), "v"(A_[i]));
            }
    };

#define ML_SB() __builtin_amdgcn_sched_barrier(0)
#define ML_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    for (int c = 0; c < nchunk; ++c) {
        // ================= phase 1: H chunk = X W1c^T over K = 512 =================
        ML_BARRIER();                                   // every wave is past its reads of the stage area (W2 stages of the previous chunk)
        issue1(c, 0, 0);
        issue1(c, 1, 1);
        f32x4t acc1[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc1[i][j] = f32x4t{0.f, 0.f, 0.f, 0.f};
        int buf = 0;
        for (int kk = 0; kk < NK1; ++kk) {
            if (kk + 1 < NK1) wait_vm<5>();             // stage kk has landed; stage kk + 1 may stay in flight
            else wait_vm<0>();
            ML_BARRIER();                               // ... for every wave, and stage kk - 1 is no longer read
            if (kk + 2 < NK1) issue1(c, kk + 2, buf >= 1 ? buf - 1 : ML_NS1 - 1);      // = (kk + 2) % 3
            const unsigned char* st = stg + buf * ML_S1;
            buf = buf + 1 == ML_NS1 ? 0 : buf + 1;
            bf16x8 ahl[4], ahm[4], blh[2], bmh[2], bhm[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ro = (64 * wm + 16 * i) * 32 + ((i & 1) ? base_o : base_e);
                ahl[i] = *reinterpret_cast<const bf16x8*>(st + plane_of(T_HL, S3_PLANE) + ro);
                ahm[i] = *reinterpret_cast<const bf16x8*>(st + plane_of(T_HM, S3_PLANE) + ro);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ro = S3_CHUNK + (32 * wn + 16 * j) * 32 + ((j & 1) ? base_o : base_e);
                blh[j] = *reinterpret_cast<const bf16x8*>(st + plane_of(T_LH, 2048) + ro);
                bmh[j] = *reinterpret_cast<const bf16x8*>(st + plane_of(T_MH, 2048) + ro);
                bhm[j] = *reinterpret_cast<const bf16x8*>(st + plane_of(T_HM, 2048) + ro);
            }
            // hl + lh, hm + mh, hh + mm: the term order of gemm_bf16x3_m16_kernel (W as the first operand: one output row per lane)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = mma16x16(blh[j], ahl[i], acc1[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = mma16x16(bmh[j], ahm[i], acc1[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc1[i][j] = mma16x16(bhm[j], ahm[i], acc1[i][j]);
        }
        ML_BARRIER();                                   // the stage area is free: the first W2 tile travels while the chunk is finished
        issue2(c * 4, 0);
        // ---- chunk epilogue: rinv, bias, GELU, split, into the H tiles (lane: row tile 2 ip + (kq & 1), 8 consecutive columns) ----
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int row = 64 * wm + 16 * (2 * ip + (kq & 1)) + l15;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc1[2 * ip][j][r]), __float_as_uint(acc1[2 * ip + 1][j][r]), false, false);
                    v[r] = __uint_as_float(sw[0]);
                    v[4 + r] = __uint_as_float(sw[1]);
                }
                const float* bp = g.b1 + c * ML_HC + 32 * wn + 16 * j + 8 * hiq;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1v = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_erf(fmaf(v[e], rinv[ip], e < 4 ? b0[e] : b1v[e - 4]));
                u32x4 Hh, Mi, Lo;
                split8<true>(v, Hh, Mi, Lo);
                unsigned char* dst = smem + (2 * wn + j) * S3_CHUNK + row * 32 + ((hiq ^ (kq & 1)) << 4);      // (row >> 4) & 1 == kq & 1
                *reinterpret_cast<u32x4*>(dst) = Hh;
                *reinterpret_cast<u32x4*>(dst + S3_PLANE) = Mi;
                *reinterpret_cast<u32x4*>(dst + 2 * S3_PLANE) = Lo;
            }
        }
        // ================= phase 2: acc2 += H chunk x W2[:, chunk]^T, four 16-k steps =================
        auto phase2_step = [&](const bool last, int kt) {
            wait_vm<0>();                                // this wave's pieces of W2 tile kt have landed
            ML_BARRIER();                                // ... every wave's, the H tiles are written, W2 tile kt - 1 is no longer read
            if (kt + 1 < 4) issue2(c * 4 + kt + 1, (kt + 1) & 1);
            const unsigned char* hs = smem + kt * S3_CHUNK;
            const unsigned char* ws = stg + (kt & 1) * ML_S2 + wave * (128 * 32);
            bf16x8 ahl[8], ahm[8], bx[8], by[8];
            auto lda = [&](bf16x8 (&dst)[8], int type) {
#pragma unroll
                for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(hs + plane_of(type, S3_PLANE) + i * 512 + ((i & 1) ? base_o : base_e));
            };
            auto ldb = [&](bf16x8 (&dst)[8], int type) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dst[j] = *reinterpret_cast<const bf16x8*>(ws + plane_of(type, ML_D * 32) + j * 512 + ((j & 1) ? base_o : base_e));
            };
            ML_SB();
            lda(ahl, T_HL); ldb(bx, T_LH); ML_SB();
            lda(ahm, T_HM); ldb(by, T_MH); ML_SB();
            mm2(ahl, bx, false); ML_SB();            // hl + lh
            ldb(bx, T_HM); ML_SB();                      // [h|m] of W2 into the dead [l|h] registers
            mm2(ahm, by, false); ML_SB();            // hm + mh
            mm2(ahm, bx, last); ML_SB();             // hh + mm
        };
        for (int kt = 0; kt < 3; ++kt) phase2_step(false, kt);
        phase2_step(true, 3);
    }
#undef ML_SB
#undef ML_BARRIER
    // ---- residual epilogue of the two-launch path (bias, + x, fp32 stream, its image, its rows' sums of squares) ----
    int lane_e = lane;
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(lane_e), "+a"(acc2[0][6][0]), "+a"(acc2[0][6][1]), "+a"(acc2[0][6][2]), "+a"(acc2[0][6][3]), "+a"(acc2[1][6][0]),
                   "+a"(acc2[1][6][1]), "+a"(acc2[1][6][2]), "+a"(acc2[1][6][3]), "+a"(acc2[0][7][0]), "+a"(acc2[0][7][1]), "+a"(acc2[0][7][2]),
                   "+a"(acc2[0][7][3]), "+a"(acc2[1][7][0]), "+a"(acc2[1][7][1]), "+a"(acc2[1][7][2]), "+a"(acc2[1][7][3]));
    s3_epilogue_img16<S3_EPI_RES_IMG, 8>(g.epi, acc2[0], m0, wave * 128, lane_e);
    s3_epilogue_img16<S3_EPI_RES_IMG, 8>(g.epi, acc2[1], m0, wave * 128 + 64, lane_e);
}

int g_mlp_fused = getenv("AVD_MLP_FUSED") ? atoi(getenv("AVD_MLP_FUSED")) : 0;

bool mlp_bf16x3_supported(int d, int hidden, int terms) { return d == ML_D && hidden > 0 && hidden % 128 == 0 && (terms == 0 || terms == 6); }

// y = R + fc2(GELU(fc1(RMSNorm(x)))) from the images described above; C3 / ss_out (both or neither): the new stream's image and sums of squares
int mlp_bf16x3(const void* X3, const void* W1n3, const float* b1, const void* W23, const float* b2, const float* ss_in, float eps,
               const float* R, float* C, void* C3, float* ss_out, int64_t M, int d, int hidden, hipStream_t st) {
    AVD_REQUIRE(X3 && W1n3 && b1 && W23 && b2 && ss_in && R && C, AVD_EINVAL, "mlp_bf16x3: null pointer");
    AVD_REQUIRE((C3 != nullptr) == (ss_out != nullptr), AVD_EINVAL, "mlp_bf16x3: the image and the sums of squares come together");
    AVD_REQUIRE(mlp_bf16x3_supported(d, hidden, 6), AVD_EUNSUPPORTED, "mlp_bf16x3: d must be 512 and hidden a multiple of 128 (d=%d hidden=%d)", d, hidden);
    AVD_REQUIRE(M > 0 && (M + ML_BM - 1) / ML_BM < (1ll << 31), AVD_EINVAL, "mlp_bf16x3: bad row count");
    AVD_REQUIRE(aligned16(X3) && aligned16(W1n3) && aligned16(W23) && aligned16(b1) && aligned16(b2) && aligned16(ss_in) && aligned16(R) &&
                    aligned16(C) && aligned16(C3), AVD_EUNSUPPORTED, "mlp_bf16x3: pointers must be 16-byte aligned");
    static LdsAttr attr;
    if (int rc = attr.ensure(reinterpret_cast<const void*>(mlp_bf16x3_kernel), ML_LDS, "mlp_bf16x3")) return rc;
    MlpArgs a{};
    a.X = static_cast<const unsigned char*>(X3);
    a.W1 = static_cast<const unsigned char*>(W1n3);
    a.b1 = b1;
    a.W2 = static_cast<const unsigned char*>(W23);
    a.ss_in = ss_in;
    a.hidden = hidden;
    a.sqrt_d = (float)sqrt((double)d);
    a.eps = eps;
    a.epi = S3Args{};
    a.epi.bias = b2;
    a.epi.R = R;
    a.epi.C = C;
    a.epi.C3 = static_cast<unsigned char*>(C3);
    a.epi.ss_out = ss_out;
    a.epi.M = M;
    a.epi.N = d;
    a.epi.K = hidden;
    a.epi.terms = 6;
    static const int tag = prof_tag_id("mlp_bf16x3_kernel");
    ProfScope prof(tag, 4.0 * (double)M * d * hidden, st);
    hipLaunchKernelGGL(mlp_bf16x3_kernel, dim3((unsigned)((M + ML_BM - 1) / ML_BM)), dim3(256), ML_LDS, st, a);
    AVD_CHECK_LAUNCH("mlp_bf16x3");
    return AVD_OK;
}

}  // namespace avd
