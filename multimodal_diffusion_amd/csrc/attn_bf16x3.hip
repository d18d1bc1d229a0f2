// Attention on the 16-bit matrix pipes with split operands — the counterpart of attn_f32.hip
// (nn.MultiheadAttention's scaled-dot-product step, avdiff/models/mmdt.py:51-61), used with gemm_bf16x3.hip.
//
// Both contractions run on v_mfma_f32_32x32x16_{bf16,f16} with split operands and fp32 accumulation (modes and error analysis in
// gemm_bf16x3.hip: TERMS 6 / 9 = three exact bf16 planes, 1 = one bf16 plane, 3 = "f16x2", two scaled fp16 planes):
//   S^T = K Q^T : K and Q planes come from the qkv3 image the in_proj GEMM epilogue wrote (q pre-multiplied by
//                 scale * log2 e, so the scores are already in the exp2 domain; f16x2: the image's scale^2 is divided out);
//   softmax     : in registers, as in attn_f32.hip (a lane holds 16 + 16 keys of ONE query column);
//   O^T = V^T P^T: P is split in registers (v_cvt_pk_bf16_f32; f16x2: at scale 2^15, p <= 1) and, because an MFMA may sum k in any
//                 order, the score accumulators feed the B operand without moving between lanes; V stays row-major [key][d] in LDS
//                 and is read transposed by ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group) in the key order P has.
// K/V tiles of 64 keys are 24 KiB contiguous, pre-swizzled pieces of the image: the global->LDS DMA is a linear copy.
// Measured (B=64, N=421, H=8): bf16x3 164 us, f16x2 120 us, against 228 us for the fp32-MFMA kernel; at N=1573 f16x2 reaches
// 263 fp32-equivalent TFLOP/s.  Staggering the two co-resident blocks, a double-buffered V tile and software-pipelined fragment
// reads were measured and change nothing: the step runs on the socket power cap.
#include "avd_common.h"

#include <stdlib.h>
#include <type_traits>

namespace avd {

constexpr int A3_DH = 64, A3_KT = 64, A3_NW = 4;
constexpr float A3_NEG = -1.0e30f;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4t __attribute__((ext_vector_type(4)));

#define AVD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AVD_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// product terms per k (see S3Terms in gemm_bf16x3.hip): 6 default, 9 strict, 1 plain bf16 operands
template <int TERMS> struct A3Terms;
template <> struct A3Terms<6> { static constexpr int N = 6; static constexpr int PA[6] = {2, 0, 1, 1, 0, 0}; static constexpr int PB[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct A3Terms<9> { static constexpr int N = 9; static constexpr int PA[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0}; static constexpr int PB[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}; };
template <> struct A3Terms<1> { static constexpr int N = 1; static constexpr int PA[1] = {0}; static constexpr int PB[1] = {0}; };
template <> struct A3Terms<3> { static constexpr int N = 3; static constexpr int PA[3] = {0, 1, 0}; static constexpr int PB[3] = {1, 0, 0}; };   // f16x2

template <bool SPLIT_OUT, int TERMS>   // SPLIT_OUT: the [B*N, H*64] result is written as a split3 image (A operand of out_proj)
__global__ __launch_bounds__(A3_NW * 64, 2) void attn_bf16x3_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                                                    int Bt, int N, int Npad, int H, int n_query, int nqb,
                                                                    float s_inv2, float v_inv, float o_scale, int out_tok) {
    // TERMS == 3 (f16x2 image of scale s): s_inv2 = 1 / s^2 takes the scores back to the exp2 domain, the probabilities are split
    // at scale 2^15 (p <= 1), v_inv = 1 / s undoes V's scale, o_scale is the scale of the image written (SPLIT_OUT)
    constexpr int NW = A3_NW, ROWB = QKV3_ROWB;
    constexpr bool F16 = TERMS == 3;
    constexpr int NPL = s3_planes(TERMS);
    constexpr int PPW = 24 / NW;                       // 1-KiB DMA pieces per wave per 24 KiB operand tile
    __shared__ __attribute__((aligned(16))) unsigned char Ks[A3_KT * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[A3_KT * ROWB];

    // XCD-aware 1-D grid: the q-blocks of one (sample, head) — which re-read the same K/V — share an L2
    int qb, h, b;
    {
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        const int w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
        qb = w % nqb;
        h = (w / nqb) % H;
        b = w / (nqb * H);
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int64_t hstride = (int64_t)Npad * ROWB;
    const unsigned char* Qb = img + (((int64_t)0 * Bt + b) * H + h) * hstride;
    const unsigned char* Kb = img + (((int64_t)1 * Bt + b) * H + h) * hstride;
    const unsigned char* Vb = img + (((int64_t)2 * Bt + b) * H + h) * hstride;

    // Q fragments: lane (q = l31, half hi), d-step s: Q[q][16 s + 8 hi .. +7] of each plane
    const int q_row = qb * (NW * 32) + wave * 32 + l31;
    bf16x8 qf[4][3];
    {
        const unsigned char* src = Qb + (int64_t)(q_row < N ? q_row : N - 1) * ROWB + hi * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < NPL; ++p) qf[s][p] = *reinterpret_cast<const bf16x8*>(src + p * 128 + s * 32);
    }

    // DMA of tile kt: linear, except that rows past the last token are fetched from token N-1 (finite filler: the scores
    // of those keys are masked, and a zero probability times a finite V is zero)
    auto dma = [&](const unsigned char* gsrc, unsigned char* ldst, int kt) {
        const int last_row = N - 1 - kt * A3_KT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            const int off = p * 1024 + lane * 16;
            int row = off / ROWB;
            const int within = off - row * ROWB;
            row = row < last_row ? row : last_row;
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(gsrc + ((int64_t)kt * A3_KT + row) * ROWB + within), AVD_LDS_PTR(ldst + p * 1024), 16,
                                             0, 0);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = A3_NEG, l_run = 0.f;

    const int nkt = (N + A3_KT - 1) / A3_KT;
    dma(Kb, Ks, 0);
    dma(Vb, Vs, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();

    // K row reads: key = 32 kb + l31, chunk (2 s + hi) ^ ((key>>1)&7)  (the swizzle is the same for key and key + 32)
    const int ksw = (l31 >> 1) & 7;
    const int k_rd = l31 * ROWB;
    // V transposed reads: 16-lane group g = lane>>4 takes d columns 16 (g&1) .. +15 of a 32-d block; lane 4q+p of the group
    // supplies the address of key row q, columns 4p .. 4p+3, and receives column (lane & 15) of the four rows
    const int i16 = lane & 15, cb = (lane >> 4) & 1;
    const int v_q = i16 >> 2, v_p = i16 & 3;

    const bool active = qb * (NW * 32) + wave * 32 < n_query;   // wave-uniform: a wave of padding rows only loads
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (!active) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (more) dma(Kb, Ks, kt + 1);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (more) dma(Vb, Vs, kt + 1);
            continue;
        }
        // ---- S^T = K Q^T for keys [0,32) and [32,64) of the tile (reads Ks only) ----
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
        // (K plane, Q plane) resp. (V plane, P plane), small terms first
        using TT = A3Terms<TERMS>;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8 ka[3], kb2[3];
            const int ch = ((2 * s + hi) ^ ksw) << 4;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                ka[p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + p * 128 + ch);
                kb2[p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + 32 * ROWB + p * 128 + ch);
            }
#pragma unroll
            for (int t = 0; t < TT::N; ++t) {
                s0 = mma16<F16>(ka[TT::PA[t]], qf[s][TT::PB[t]], s0);
                s1 = mma16<F16>(kb2[TT::PA[t]], qf[s][TT::PB[t]], s1);
            }
        }
        // K is free once every wave is here; V(kt) (issued a phase ago) has landed after the wait
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma(Kb, Ks, kt + 1);

        if constexpr (F16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { s0[r] *= s_inv2; s1[r] *= s_inv2; }
        }
        if (!more && (N & (A3_KT - 1))) {   // ragged last tile: keys >= N contribute nothing
            const int kbase = kt * A3_KT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + mfma32_row(r, hi);
                if (key >= N) s0[r] = A3_NEG;
                if (key + 32 >= N) s1[r] = A3_NEG;
            }
        }

        // ---- online softmax for this lane's query column (scores are in the exp2 domain) ----
        float mt = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s1[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float m_sub = F16 ? m_new - 15.0f : m_new;    // f16x2: probabilities are kept at scale 2^15 (sum and planes alike)
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_sub);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_sub);
            ps += s0[r] + s1[r];
        }
        if (__any(m_new > m_run)) {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        l_run += ps;
        m_run = m_new;

        // ---- O^T += V^T P^T (reads Vs only): k-step (kb, t) covers keys 32 kb + 16 t + 4 hi + (j&3) + 8 (j>>2), j = fragment
        //      element — the key order in which accumulator registers 8t .. 8t+7 of S^T hold P ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float pv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = kb ? s1[8 * t + j] : s0[8 * t + j];
                u32x4 P[3];
                if constexpr (F16) split8_h2(pv, 1.0f, P[0], P[1]);
                else split8<false>(pv, P[0], P[1], P[2]);
                bf16x8 pf[3];
#pragma unroll
                for (int p = 0; p < NPL; ++p) pf[p] = __builtin_bit_cast(bf16x8, P[p]);
                const int key0 = 32 * kb + 16 * t + 4 * hi + v_q;     // this lane's ADDRESS row of the first 4-key block
                const int sw = qkv3_swizzle(2, key0);                 // same for key0 + 8
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const int chunk = 4 * db + 2 * cb + (v_p >> 1);
                    bf16x8 vf[3];
#pragma unroll
                    for (int p = 0; p < NPL; ++p) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)AVD_LDS_PTR(
                            Vs + key0 * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)AVD_LDS_PTR(
                            Vs + (key0 + 8) * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                        const u32x2 a = __builtin_bit_cast(u32x2, lo), c2 = __builtin_bit_cast(u32x2, up);
                        const u32x4 w = {a[0], a[1], c2[0], c2[1]};
                        vf[p] = __builtin_bit_cast(bf16x8, w);
                    }
#pragma unroll
                    for (int tt = 0; tt < TT::N; ++tt) {
                        if (db == 0) o0 = mma16<F16>(vf[TT::PA[tt]], pf[TT::PB[tt]], o0);
                        else o1 = mma16<F16>(vf[TT::PA[tt]], pf[TT::PB[tt]], o1);
                    }
                }
            }
        // V is free once every wave is here; K(kt+1) has landed after the wait
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma(Vb, Vs, kt + 1);
    }
    if (!active) return;      // no barriers below

    // ---- normalise and store: lane (q, hi) holds O[q][8 g + 4 hi + (0..3)] in regs 4g..4g+3 of o0 (d < 32) / o1 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    // f16x2: o carries 2^15 s_v, l_tot carries 2^15
    const float inv = F16 ? v_inv / l_tot : 1.0f / l_tot;
    const int d = H * A3_DH;
    if constexpr (SPLIT_OUT) {
        float ch[8][4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ch[g4][e] = o0[4 * g4 + e] * inv;
                ch[4 + g4][e] = o1[4 * g4 + e] * inv;
            }
        unsigned char* o3 = reinterpret_cast<unsigned char*>(out);
#pragma unroll
        for (int c = 0; c < 8; c += 2) {     // the low half-lane ends up with even 8-d chunks, the high half-lane with odd ones
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float send = hi ? ch[c][e] : ch[c + 1][e];
                const float recv = __shfl_xor(send, 32, 64);
                v[e] = hi ? recv : ch[c][e];
                v[4 + e] = hi ? ch[c + 1][e] : recv;
            }
            if (q_row < n_query) {
                if constexpr (F16) store_split8_h2(o3, (int64_t)b * out_tok + q_row, h * A3_DH + 8 * (c + hi), d, v, o_scale);
                else store_split8(o3, (int64_t)b * out_tok + q_row, h * A3_DH + 8 * (c + hi), d, v);
            }
        }
    } else if (q_row < n_query) {
        float* dst = out + ((int64_t)b * out_tok + q_row) * d + h * A3_DH + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<f32x4*>(dst + 8 * g4) = a;
            *reinterpret_cast<f32x4*>(dst + 32 + 8 * g4) = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 4: the same algorithm as ONE software pipeline per wave (attn_bf16x3_pipe_kernel; default for the three-plane modes, avd_tune_set
// "attn_pipe" 0 takes the kernel above everywhere, 2 this one everywhere).  Why: in the kernel above a wave's 96 MFMAs per 64-key tile (3,072 matrix-pipe cycles, six terms) and its ~1,900 cycles
// of softmax / split VALU work run one after the other, and on a SIMD the two ADD (DESIGN.md 4.8 g): 0.32 of the MFMA peak with the pipe
// 40 % busy.  A v_mfma_f32_32x32x16 occupies the SIMD's vector issue for 8 of its 32 cycles; the other 24 take up to five single-issue
// VALU instructions of the SAME wave for free (MI355X_MICROARCH.md, cycle constants) — but only instructions that do not depend on the
// MFMAs around them.  So the tile loop is skewed by half a tile:
//   phase A(kt):  S^T(kt+1) = K(kt+1) Q^T   [48 MFMAs, six terms]   beside   p(kt) = exp2(s(kt) - m), row sums, O *= alpha, split of
//                                                                           the first 16-key group of p(kt)
//   phase B(kt):  O^T += V(kt)^T P(kt)^T     [48 MFMAs]              beside   split of key groups 1..3 of p(kt), max of s(kt+1)
// with the VALU work cut into items that are dealt out behind every MFMA pair in program order (a sched_barrier pins each slot).
// K(kt+2) is fetched while phase B runs, V(kt+1) while phase A runs: the same two LDS tiles and two barriers per tile as before.
// The O rescale is unconditional here (alpha = 1 when no maximum moved): a wave-uniform branch would cut the schedule in two.
int g_attn_pipe = getenv("AVD_ATTN_PIPE") ? atoi(getenv("AVD_ATTN_PIPE")) : 1;

// Round 5: the tile loop is PEELED.  The instruction mix of round 4's loop body (867 instructions per 64-key tile, 96 of them MFMAs) held ~250
// vector instructions that do no arithmetic of the algorithm: the ragged-tail mask (32 compares + 32 selects, executed by EVERY tile because a
// branch would have cut the schedule), the DMA's per-piece row / clamp / 64-bit address arithmetic (24 v_mad_u64 + ... for 12 pieces), and
// 32 register moves that hand the next tile's scores to the current tile's names.  Now
//   * only the LAST tile can be ragged, so only the iteration that issues its K fetch (KCL), the one that issues its V fetch and computes its
//     scores (VCL) and the prologue carry the clamp / the mask; every other iteration moves whole 24 KiB tiles with one wave-uniform base + lane
//     offset per piece and masks nothing (template flags: the bodies are separate instantiations, as the last tile already was);
//   * the main loop is unrolled twice with the two score register sets trading roles — no copies (the tail iterations copy).
// The VALU work that is left per tile is the algorithm's: 32 exp2 + 32 subtractions, 32 row-sum adds, 32 O multiplies, 16 max3, the 16 pair
// splits of P (11 instructions per pair for three planes).
template <bool MORE_, bool KCL_, bool VCL_, bool COPY_> struct A3Flags {
    static constexpr bool MORE = MORE_, KCL = KCL_, VCL = VCL_, COPY = COPY_;
};

template <bool SPLIT_OUT, int TERMS>
__global__ __launch_bounds__(A3_NW * 64, 2) void attn_bf16x3_pipe_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                                                         int Bt, int N, int Npad, int H, int n_query, int nqb,
                                                                         float s_inv2, float v_inv, float o_scale, int out_tok) {
    constexpr int NW = A3_NW, ROWB = QKV3_ROWB;
    constexpr bool F16 = TERMS == 3;
    constexpr int NPL = s3_planes(TERMS);
    constexpr int PPW = 24 / NW;
    using TT = A3Terms<TERMS>;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[A3_KT * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[A3_KT * ROWB];

    int qb, h, b;
    {
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        const int w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
        qb = w % nqb;
        h = (w / nqb) % H;
        b = w / (nqb * H);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: what derives from it stays scalar
    const int l31 = lane & 31, hi = lane >> 5;
    const int64_t hstride = (int64_t)Npad * ROWB;
    const unsigned char* Qb = img + (((int64_t)0 * Bt + b) * H + h) * hstride;
    const unsigned char* Kb = img + (((int64_t)1 * Bt + b) * H + h) * hstride;
    const unsigned char* Vb = img + (((int64_t)2 * Bt + b) * H + h) * hstride;

    const int q_row = qb * (NW * 32) + wave * 32 + l31;
    bf16x8 qf[4][3];
    {
        const unsigned char* src = Qb + (int64_t)(q_row < N ? q_row : N - 1) * ROWB + hi * 16;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < NPL; ++p) qf[s][p] = *reinterpret_cast<const bf16x8*>(src + p * 128 + s * 32);
    }
    // DMA of tile kt.  dma_clamp: rows past the last token are fetched from token N-1 (finite filler: the scores of those keys are masked,
    // and a zero probability times a finite V is zero) — needed for the last tile only.  dma_full: a whole tile, linear: wave-uniform base,
    // the lane's 16 bytes as the only vector part of the address.
    auto dma_clamp = [&](const unsigned char* gsrc, unsigned char* ldst, int kt) {
        const int last_row = N - 1 - kt * A3_KT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            const int off = p * 1024 + lane * 16;
            int row = off / ROWB;
            const int within = off - row * ROWB;
            row = row < last_row ? row : last_row;
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(gsrc + ((int64_t)kt * A3_KT + row) * ROWB + within), AVD_LDS_PTR(ldst + p * 1024), 16,
                                             0, 0);
        }
    };
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma_full = [&](const unsigned char* gsrc, unsigned char* ldst, int kt) {
        const unsigned char* base = gsrc + ((int64_t)kt * (A3_KT * ROWB) + wave * 1024);        // wave-uniform
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(base + i * (NW * 1024) + lane16), AVD_LDS_PTR(ldst + (wave + NW * i) * 1024), 16, 0, 0);
    };
    auto sync = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's DMA pieces have landed
        __syncthreads();
    };

    const int nkt = (N + A3_KT - 1) / A3_KT;
    const bool ragged = (N & (A3_KT - 1)) != 0;
    dma_clamp(Kb, Ks, 0);
    dma_clamp(Vb, Vs, 0);
    sync();

    const bool active = qb * (NW * 32) + wave * 32 < n_query;   // wave-uniform: a wave of padding rows only loads
    if (!active) {
        // same barriers and the same DMA duty as the computing waves
        sync();
        if (nkt > 1) dma_clamp(Kb, Ks, 1);
        sync();
        for (int kt = 0; kt < nkt; ++kt) {
            sync();
            if (kt + 2 < nkt) dma_clamp(Kb, Ks, kt + 2);
            if (kt + 1 < nkt) {
                sync();
                dma_clamp(Vb, Vs, kt + 1);
            }
        }
        return;
    }

    const int ksw = (l31 >> 1) & 7;
    const int k_rd = l31 * ROWB;
    const int i16 = lane & 15, cb = (lane >> 4) & 1;
    const int v_q = i16 >> 2, v_p = i16 & 3;
#define A3_SB() __builtin_amdgcn_sched_barrier(0)

    f32x16 o0, o1, sa0, sa1, sb0, sb1;        // O^T halves; two score / probability register sets that trade roles (current tile / next tile)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; sa0[r] = 0.f; sa1[r] = 0.f; sb0[r] = 0.f; sb1[r] = 0.f; }
    float m_run = A3_NEG, l_run = 0.f;

    // the two MFMAs (keys 0..31 and 32..63) of term t of d-step s of S^T = K Q^T, with that d-step's K fragments read in front of term 0
    bf16x8 ka[3], kb2[3];
    auto s_slot = [&](f32x16& d0, f32x16& d1, int s, int t) {
        if (t == 0) {
            const int ch = ((2 * s + hi) ^ ksw) << 4;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                ka[p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + p * 128 + ch);
                kb2[p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + 32 * ROWB + p * 128 + ch);
            }
        }
        d0 = mma16<F16>(ka[TT::PA[t]], qf[s][TT::PB[t]], d0);
        d1 = mma16<F16>(kb2[TT::PA[t]], qf[s][TT::PB[t]], d1);
    };
    auto mask_tail = [&](f32x16& d0, f32x16& d1, int kt) {      // ragged last tile: keys >= N contribute nothing
        const int kbase = kt * A3_KT;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kbase + mfma32_row(r, hi);
            if (key >= N) d0[r] = A3_NEG;
            if (key + 32 >= N) d1[r] = A3_NEG;
        }
    };
    // running maximum over this lane's query column (both lane halves), in the exp2 domain (f16x2: scores still carry the image scale^2)
    auto tile_max = [&](const f32x16& d0, const f32x16& d1) {
        float mt = fmaxf(d0[0], d1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, fmaxf(d0[r], d1[r]));
        if constexpr (F16) mt *= s_inv2;
        return fmaxf(mt, __shfl_xor(mt, 32, 64));
    };

    // ---- prologue: S(0) with nothing beside it, K(1) on its way while its maximum is taken ----
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < TT::N; ++t) s_slot(sa0, sa1, s, t);
    if (nkt == 1 && ragged) mask_tail(sa0, sa1, 0);
    sync();
    if (nkt > 1) dma_clamp(Kb, Ks, 1);
    float m_new = fmaxf(m_run, tile_max(sa0, sa1));
    sync();

    u32x4 pf[2][3];               // P fragments (planes) of two consecutive 16-key groups
    // pair e (two of the eight probabilities) of key group g = (kb, t) of the current tile (c0 / c1), split into planes: dword e of every plane
    auto split_pair = [&](const f32x16& c0, const f32x16& c1, int g, int e, u32x4 (&dst)[3]) {
        const float a = (g >> 1) ? c1[8 * (g & 1) + 2 * e] : c0[8 * (g & 1) + 2 * e];
        const float c = (g >> 1) ? c1[8 * (g & 1) + 2 * e + 1] : c0[8 * (g & 1) + 2 * e + 1];
        if constexpr (F16) {
            const unsigned int hh = pk_f16(a, c);
            const f32x2 u = unpk_f16(hh);
            dst[0][e] = hh;
            dst[1][e] = pk_f16(a - u[0], c - u[1]);
        } else {
            const unsigned int hh = pk_bf16(a, c);
            const float ra = a - bf16_lo(hh), rc = c - bf16_hi(hh);
            dst[0][e] = hh;
            if constexpr (NPL > 1) {
                const unsigned int mm = pk_bf16(ra, rc);
                dst[1][e] = mm;
                dst[2][e] = pk_bf16(ra - bf16_lo(mm), rc - bf16_hi(mm));
            }
        }
    };
    // the MFMAs of (key group g, d block db, term tt) of O^T += V^T P^T, that block's V fragments read in front of term 0
    bf16x8 vf[3];
    auto pv_slot = [&](int g, int db, int tt, const u32x4 (&P)[3]) {
        if (tt == 0) {
            const int key0 = 16 * g + 4 * hi + v_q;
            const int sw = qkv3_swizzle(2, key0);
            const int chunk = 4 * db + 2 * cb + (v_p >> 1);
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)AVD_LDS_PTR(
                    Vs + key0 * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)AVD_LDS_PTR(
                    Vs + (key0 + 8) * ROWB + p * 128 + ((chunk ^ sw) << 4) + 8 * (v_p & 1)));
                const u32x2 a = __builtin_bit_cast(u32x2, lo), c2 = __builtin_bit_cast(u32x2, up);
                const u32x4 w = {a[0], a[1], c2[0], c2[1]};
                vf[p] = __builtin_bit_cast(bf16x8, w);
            }
        }
        if (db == 0) o0 = mma16<F16>(vf[TT::PA[tt]], __builtin_bit_cast(bf16x8, P[TT::PB[tt]]), o0);
        else o1 = mma16<F16>(vf[TT::PA[tt]], __builtin_bit_cast(bf16x8, P[TT::PB[tt]]), o1);
    };

    // One tile.  c0 / c1: scores (then probabilities) of tile kt; n0 / n1: receive the scores of tile kt + 1.  Flags (compile time; each
    // combination is its own instantiation — as the two arms of a runtime `if` the compiler hoists the VALU items common to both arms in
    // front of the branch and the schedule is gone): MORE a tile kt + 1 exists; KCL tile kt + 2 may be the ragged last one (its K fetch
    // clamps rows); VCL tile kt + 1 may be the ragged last one (its V fetch clamps, its scores are masked); COPY hand n back to c at the end.
    auto tile = [&](auto flags, int kt, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
        using FL = decltype(flags);
        constexpr bool MORE = FL::MORE;
        // ================= phase A: S(kt+1) beside the exponentials of tile kt =================
        const float m_sub = F16 ? m_new - 15.0f : m_new;        // f16x2: probabilities are kept at scale 2^15 (sum and planes alike)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float ps = 0.f;
        // VALU items of this phase: 0..15 two exponentials each (register r of both halves), 16..23 four O multiplies each, 24..27 one
        // pair each of the split of key group 0 — dealt out over the 4 x TT::N MFMA slots
        constexpr int NSLOT_A = 4 * TT::N, NITEM_A = 28;
        // Every item ends in an empty volatile statement that names its results: nothing in this phase consumes them, so without it the
        // compiler sinks the whole item into the block of its first use (behind the barrier) and the MFMAs run alone again.
        auto item_a = [&](int w) {
            if (w < 16) {
                const float e0 = F16 ? c0[w] * s_inv2 : c0[w], e1 = F16 ? c1[w] * s_inv2 : c1[w];
                float p0 = __builtin_amdgcn_exp2f(e0 - m_sub), p1 = __builtin_amdgcn_exp2f(e1 - m_sub);
                ps += p0 + p1;
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(ps));
                c0[w] = p0;
                c1[w] = p1;
            } else if (w < 24) {
                const int r0 = (w - 16) * 2;
                float a0 = o0[r0] * alpha, a1 = o0[r0 + 1] * alpha, q0 = o1[r0] * alpha, q1 = o1[r0 + 1] * alpha;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(q0), "+v"(q1));
                o0[r0] = a0; o0[r0 + 1] = a1;
                o1[r0] = q0; o1[r0 + 1] = q1;
            } else {
                split_pair(c0, c1, 0, w - 24, pf[0]);
                // (name only the planes this mode writes: an unwritten plane as a read-write operand would read an uninitialised register
                // and pin it for nothing — ADVICE r4)
                if constexpr (NPL == 3) asm volatile("" : "+v"(pf[0][0][w - 24]), "+v"(pf[0][1][w - 24]), "+v"(pf[0][2][w - 24]));
                else if constexpr (NPL == 2) asm volatile("" : "+v"(pf[0][0][w - 24]), "+v"(pf[0][1][w - 24]));
                else asm volatile("" : "+v"(pf[0][0][w - 24]));
            }
        };
        if constexpr (MORE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { n0[r] = 0.f; n1[r] = 0.f; }
#pragma unroll
            for (int sl = 0; sl < NSLOT_A; ++sl) {
                A3_SB();
                s_slot(n0, n1, sl / TT::N, sl % TT::N);
#pragma unroll
                for (int w = sl * NITEM_A / NSLOT_A; w < (sl + 1) * NITEM_A / NSLOT_A; ++w) item_a(w);
            }
            A3_SB();
            if constexpr (FL::VCL) {
                if (ragged) mask_tail(n0, n1, kt + 1);          // wave-uniform; this instantiation runs once per block
            }
        } else {
#pragma unroll
            for (int w = 0; w < NITEM_A; ++w) item_a(w);
        }
        l_run = l_run * alpha + ps;
        // V(kt) has landed (issued a phase ago); every wave is past its reads of K(kt+1)
        sync();
        if constexpr (MORE) {
            if constexpr (FL::KCL) { if (kt + 2 < nkt) dma_clamp(Kb, Ks, kt + 2); }
            else if constexpr (!FL::VCL) dma_full(Kb, Ks, kt + 2);      // (VCL: tile kt + 1 is the last one, nothing left to fetch)
        }

        // ================= phase B: O += V(kt) P(kt) beside the remaining splits and the maximum of tile kt+1 =================
        // slots: (g, db, tt) in that order, 2 x TT::N per key group.  Items: the split of group g+1, one pair behind each of the first
        // slots of group g, and the 16 max pairs of s(kt+1) spread over the last group's slots
        float mt = A3_NEG;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int sl = 0; sl < 2 * TT::N; ++sl) {
                A3_SB();
                pv_slot(g, sl / TT::N, sl % TT::N, pf[g & 1]);
                if (g < 3) {        // the split of the next group: its four pairs behind the first slots of this group
                    constexpr int SPS = 2 * TT::N >= 4 ? 1 : 2;      // pairs per slot (one-term mode: two slots per group)
#pragma unroll
                    for (int e = sl * SPS; e < (sl + 1) * SPS && e < 4; ++e) split_pair(c0, c1, g + 1, e, pf[(g + 1) & 1]);
                }
                if (MORE && g == 3) {
#pragma unroll
                    for (int w = sl * 16 / (2 * TT::N); w < (sl + 1) * 16 / (2 * TT::N); ++w) mt = fmaxf(mt, fmaxf(n0[w], n1[w]));
                    asm volatile("" : "+v"(mt));      // stays in this slot
                }
            }
        }
        A3_SB();
        if constexpr (MORE) {
            if constexpr (F16) mt *= s_inv2;
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            m_new = fmaxf(m_run, mt);
            if constexpr (FL::COPY) {
                c0 = n0;
                c1 = n1;
            }
            // K(kt+2) has landed; every wave is past its reads of V(kt)
            sync();
            if constexpr (FL::VCL) dma_clamp(Vb, Vs, kt + 1);
            else dma_full(Vb, Vs, kt + 1);
        }
    };
    {
        using Main = A3Flags<true, false, false, false>;        // tiles kt + 1 and kt + 2 exist and are full; register sets trade roles
        using MainC = A3Flags<true, false, false, true>;
        using KClamp = A3Flags<true, true, false, true>;        // tile kt + 2 is the last one (if it exists)
        using VClamp = A3Flags<true, false, true, true>;        // tile kt + 1 is the last one
        using Last = A3Flags<false, false, false, false>;
        int kt = 0;
        for (; kt + 4 < nkt; kt += 2) {
            tile(Main{}, kt, sa0, sa1, sb0, sb1);
            tile(Main{}, kt + 1, sb0, sb1, sa0, sa1);
        }
        // 1 .. 4 tiles left, the current scores in sa: each of these hands the next scores back to sa
        if (nkt - kt == 4) { tile(MainC{}, kt, sa0, sa1, sb0, sb1); ++kt; }
        if (nkt - kt == 3) { tile(KClamp{}, kt, sa0, sa1, sb0, sb1); ++kt; }
        if (nkt - kt == 2) { tile(VClamp{}, kt, sa0, sa1, sb0, sb1); ++kt; }
        tile(Last{}, kt, sa0, sa1, sb0, sb1);
    }
#undef A3_SB

    // ---- normalise and store: lane (q, hi) holds O[q][8 g + 4 hi + (0..3)] in regs 4g..4g+3 of o0 (d < 32) / o1 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = F16 ? v_inv / l_tot : 1.0f / l_tot;
    const int d = H * A3_DH;
    if constexpr (SPLIT_OUT) {
        float ch[8][4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ch[g4][e] = o0[4 * g4 + e] * inv;
                ch[4 + g4][e] = o1[4 * g4 + e] * inv;
            }
        unsigned char* o3 = reinterpret_cast<unsigned char*>(out);
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float send = hi ? ch[c][e] : ch[c + 1][e];
                const float recv = __shfl_xor(send, 32, 64);
                v[e] = hi ? recv : ch[c][e];
                v[4 + e] = hi ? ch[c + 1][e] : recv;
            }
            if (q_row < n_query) {
                if constexpr (F16) store_split8_h2(o3, (int64_t)b * out_tok + q_row, h * A3_DH + 8 * (c + hi), d, v, o_scale);
                else store_split8(o3, (int64_t)b * out_tok + q_row, h * A3_DH + 8 * (c + hi), d, v);
            }
        }
    } else if (q_row < n_query) {
        float* dst = out + ((int64_t)b * out_tok + q_row) * d + h * A3_DH + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<f32x4*>(dst + 8 * g4) = a;
            *reinterpret_cast<f32x4*>(dst + 32 + 8 * g4) = c;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Round 5: the same pipeline on v_mfma_f32_16x16x32 (VERDICT r4 next-round 3 a; avd_tune_set "attn_m16").  Why build it: the step runs on the
// socket power cap, cycles per FLOP of the two MFMA shapes are equal, and the 16x16x32 shape draws less power per FLOP (MI355X_MICROARCH.md,
// DVFS give-back 7: 1.12-1.15x the FLOP/s of the 32x32x16 loop at equal cycles) — the change that bought clock on the GEMMs (DESIGN 4.8 d).
// Same wave tile (64 keys x 32 queries per step, 32 queries x 64 d of O), same LDS tiles, same DMA, same number of LDS reads:
//   S^T tiles [4 key tiles of 16][2 query tiles of 16]: A = K fragment (key = lane & 15, d = 32 s + 8 (lane >> 4) + j), B = Q fragment
//       (query = lane & 15, same d); a lane holds keys 4 (lane >> 4) + r of ONE query column per tile: 2 queries x 16 keys (32x32: 1 x 32),
//       so the row maxima / sums cross the four 16-lane groups (v_permlane16_swap + v_permlane32_swap) instead of one lane half;
//   O^T tiles [4 d tiles][2 query tiles]: K = 32 keys per MFMA = the registers of S^T tiles 2g and 2g + 1 of the lane (the MFMA k order is
//       free), V read transposed by ds_read_b64_tr_b16 in exactly that key order (keys 32 g + 4 (lane >> 4) + q, then + 16).
// An MFMA of this shape holds the vector issue port for 8 of its 16 cycles (32x32x16: 8 of 32), so there is less room for the softmax VALU
// work beside the matrix pipe; which of the two effects wins is a measurement (profiles/r05_attn_m16.txt).
template <bool F16>
__device__ __forceinline__ f32x4t mma16s(bf16x8 a, bf16x8 b, f32x4t c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// Measured (profiles/r05_attn_m16.txt): C3 151.8 -> 146.2 us per launch alone (shader clock 2,140 -> 2,254 MHz at the same ~1,355 W), 157.4 -> 152.3 us inside
// the step, C5 geometry 415.6 -> 400.8 us: default 1 for the three-plane modes (2: every split mode, 0: the 32x32x16 kernels).
int g_attn_m16 = getenv("AVD_ATTN_M16") ? atoi(getenv("AVD_ATTN_M16")) : 1;

struct S16 { f32x4t t[4][2]; };       // scores / probabilities of one 64-key tile: [key tile][query tile]

template <bool SPLIT_OUT, int TERMS>
__global__ __launch_bounds__(A3_NW * 64, 2) void attn_bf16x3_p16_kernel(const unsigned char* __restrict__ img, float* __restrict__ out,
                                                                        int Bt, int N, int Npad, int H, int n_query, int nqb,
                                                                        float s_inv2, float v_inv, float o_scale, int out_tok) {
    constexpr int NW = A3_NW, ROWB = QKV3_ROWB;
    constexpr bool F16 = TERMS == 3;
    constexpr int NPL = s3_planes(TERMS);
    constexpr int PPW = 24 / NW;
    using TT = A3Terms<TERMS>;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[A3_KT * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[A3_KT * ROWB];

    int qb, h, b;
    {
        const int nwg = gridDim.x, id = blockIdx.x, q = nwg >> 3, r = nwg & 7, x = id & 7;
        const int w = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
        qb = w % nqb;
        h = (w / nqb) % H;
        b = w / (nqb * H);
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int64_t hstride = (int64_t)Npad * ROWB;
    const unsigned char* Qb = img + (((int64_t)0 * Bt + b) * H + h) * hstride;
    const unsigned char* Kb = img + (((int64_t)1 * Bt + b) * H + h) * hstride;
    const unsigned char* Vb = img + (((int64_t)2 * Bt + b) * H + h) * hstride;

    // Q fragments: query tile qt, d-step s (32 d), plane p: Q[query 16 qt + l15][32 s + 8 kq .. + 7]
    const int q_row0 = qb * (NW * 32) + wave * 32 + l15;
    bf16x8 qf[2][2][3];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qr = q_row0 + 16 * qt;
        const unsigned char* src = Qb + (int64_t)(qr < N ? qr : N - 1) * ROWB + kq * 16;
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int p = 0; p < NPL; ++p) qf[qt][sd][p] = *reinterpret_cast<const bf16x8*>(src + p * 128 + sd * 64);
    }
    auto dma_clamp = [&](const unsigned char* gsrc, unsigned char* ldst, int kt) {
        const int last_row = N - 1 - kt * A3_KT;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            const int off = p * 1024 + lane * 16;
            int row = off / ROWB;
            const int within = off - row * ROWB;
            row = row < last_row ? row : last_row;
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(gsrc + ((int64_t)kt * A3_KT + row) * ROWB + within), AVD_LDS_PTR(ldst + p * 1024), 16,
                                             0, 0);
        }
    };
    const unsigned lane16 = (unsigned)lane * 16u;
    auto dma_full = [&](const unsigned char* gsrc, unsigned char* ldst, int kt) {
        const unsigned char* base = gsrc + ((int64_t)kt * (A3_KT * ROWB) + wave * 1024);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(base + i * (NW * 1024) + lane16), AVD_LDS_PTR(ldst + (wave + NW * i) * 1024), 16, 0, 0);
    };
    auto sync = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    };

    const int nkt = (N + A3_KT - 1) / A3_KT;
    const bool ragged = (N & (A3_KT - 1)) != 0;
    dma_clamp(Kb, Ks, 0);
    dma_clamp(Vb, Vs, 0);
    sync();

    const bool active = qb * (NW * 32) + wave * 32 < n_query;
    if (!active) {
        sync();
        if (nkt > 1) dma_clamp(Kb, Ks, 1);
        sync();
        for (int kt = 0; kt < nkt; ++kt) {
            sync();
            if (kt + 2 < nkt) dma_clamp(Kb, Ks, kt + 2);
            if (kt + 1 < nkt) {
                sync();
                dma_clamp(Vb, Vs, kt + 1);
            }
        }
        return;
    }

    // K row reads: key 16 kt4 + l15, chunk (4 s + kq) ^ ((key >> 1) & 7) — 16 kt4 does not reach the swizzle bits
    const int ksw = (l15 >> 1) & 7;
    const int k_rd = l15 * ROWB;
    // V transposed reads: the 16-lane group kq takes the 4-key block 32 g + 4 kq (+ 16) of a d tile; lane 4 q + p of the group supplies the
    // address of key row q, columns 4 p .. 4 p + 3, and receives column l15 of the four rows
    const int v_q = l15 >> 2, v_p = l15 & 3;
    const int v_key = 4 * kq + v_q;
    const int v_sw = qkv3_swizzle(2, v_key);            // (+ 32 g, + 16: the swizzle bits of the key do not change)
    const int v_rd = v_key * ROWB + 8 * (v_p & 1);
#define A3_SB() __builtin_amdgcn_sched_barrier(0)

    f32x4t o[4][2];                           // O^T: [d tile][query tile]
    S16 sa, sb;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { o[i][j] = f32x4t{0.f, 0.f, 0.f, 0.f}; sa.t[i][j] = o[i][j]; sb.t[i][j] = o[i][j]; }
    float m_run[2] = {A3_NEG, A3_NEG}, l_run[2] = {0.f, 0.f}, m_new[2];

    // the 2 MFMAs (query tiles 0, 1) of term t of (key tile kt4, d-step sd) of S^T = K Q^T, that pair's K fragments read in front of term 0
    bf16x8 ka[3];
    auto s_slot = [&](S16& d, int kt4, int sd, int t) {
        if (t == 0) {
            const int ch = ((4 * sd + kq) ^ ksw) << 4;
#pragma unroll
            for (int p = 0; p < NPL; ++p) ka[p] = *reinterpret_cast<const bf16x8*>(Ks + k_rd + kt4 * (16 * ROWB) + p * 128 + ch);
        }
        d.t[kt4][0] = mma16s<F16>(ka[TT::PA[t]], qf[0][sd][TT::PB[t]], d.t[kt4][0]);
        d.t[kt4][1] = mma16s<F16>(ka[TT::PA[t]], qf[1][sd][TT::PB[t]], d.t[kt4][1]);
    };
    auto mask_tail = [&](S16& d, int kt) {
        const int kbase = kt * A3_KT + 4 * kq;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kbase + 16 * i + r >= N) { d.t[i][0][r] = A3_NEG; d.t[i][1][r] = A3_NEG; }
    };
    // maximum over the four 16-lane groups that share a query column
    auto xmax = [&](float v) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
        const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
    };
    auto xsum = [&](float v) {
        const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
        const auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(c[0]) + __uint_as_float(c[1]);
    };
    auto tile_max = [&](const S16& d, int qt) {
        float mt = A3_NEG;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) mt = fmaxf(mt, d.t[i][qt][r]);
        if constexpr (F16) mt *= s_inv2;
        return xmax(mt);
    };

    // ---- prologue: S(0) with nothing beside it, K(1) on its way while its maxima are taken ----
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int sd = 0; sd < 2; ++sd)
#pragma unroll
            for (int t = 0; t < TT::N; ++t) s_slot(sa, i, sd, t);
    if (nkt == 1 && ragged) mask_tail(sa, 0);
    sync();
    if (nkt > 1) dma_clamp(Kb, Ks, 1);
    m_new[0] = fmaxf(m_run[0], tile_max(sa, 0));
    m_new[1] = fmaxf(m_run[1], tile_max(sa, 1));
    sync();

    // P fragments: [32-key group parity][query tile][plane], dword e = pair e of the eight probabilities of (group g, query tile qt):
    // e < 2: registers 2 e, 2 e + 1 of S^T tile 2 g; e >= 2: registers 2 (e - 2), + 1 of tile 2 g + 1
    u32x4 pf[2][2][3];
    auto split_pair = [&](const S16& c, int g, int qt, int e, u32x4 (&dst)[3]) {
        const float a = c.t[2 * g + (e >> 1)][qt][2 * (e & 1)], cc = c.t[2 * g + (e >> 1)][qt][2 * (e & 1) + 1];
        if constexpr (F16) {
            const unsigned int hh = pk_f16(a, cc);
            const f32x2 u = unpk_f16(hh);
            dst[0][e] = hh;
            dst[1][e] = pk_f16(a - u[0], cc - u[1]);
        } else {
            const unsigned int hh = pk_bf16(a, cc);
            const float ra = a - bf16_lo(hh), rc = cc - bf16_hi(hh);
            dst[0][e] = hh;
            if constexpr (NPL > 1) {
                const unsigned int mm = pk_bf16(ra, rc);
                dst[1][e] = mm;
                dst[2][e] = pk_bf16(ra - bf16_lo(mm), rc - bf16_hi(mm));
            }
        }
    };
    // the 2 MFMAs (query tiles 0, 1) of (key group g, d tile dt, term tt) of O^T += V^T P^T, that (g, dt)'s V fragments read in front of term 0
    bf16x8 vf[3];
    auto pv_slot = [&](int g, int dt, int tt, const u32x4 (&P)[2][3]) {
        if (tt == 0) {
            const int chunk = (2 * dt + (v_p >> 1)) ^ v_sw;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                const unsigned char* a0 = Vs + v_rd + g * (32 * ROWB) + p * 128 + (chunk << 4);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)AVD_LDS_PTR(a0));
                const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)AVD_LDS_PTR(a0 + 16 * ROWB));
                const u32x2 a = __builtin_bit_cast(u32x2, lo), c2 = __builtin_bit_cast(u32x2, up);
                const u32x4 w = {a[0], a[1], c2[0], c2[1]};
                vf[p] = __builtin_bit_cast(bf16x8, w);
            }
        }
        o[dt][0] = mma16s<F16>(vf[TT::PA[tt]], __builtin_bit_cast(bf16x8, P[0][TT::PB[tt]]), o[dt][0]);
        o[dt][1] = mma16s<F16>(vf[TT::PA[tt]], __builtin_bit_cast(bf16x8, P[1][TT::PB[tt]]), o[dt][1]);
    };

    auto tile = [&](auto flags, int kt, S16& c, S16& n) {
        using FL = decltype(flags);
        constexpr bool MORE = FL::MORE;
        // ================= phase A: S(kt+1) beside the exponentials of tile kt =================
        float m_sub[2], alpha[2], ps[2] = {0.f, 0.f};
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            m_sub[qt] = F16 ? m_new[qt] - 15.0f : m_new[qt];
            alpha[qt] = __builtin_amdgcn_exp2f(m_run[qt] - m_new[qt]);
            m_run[qt] = m_new[qt];
        }
        // VALU items: 0..15 two exponentials each (registers 2 (w & 1), + 1 of S^T tile (w >> 2, (w >> 1) & 1)), 16..23 four O multiplies each
        // (tile (w - 16) >> 1, query tile (w - 16) & 1), 24..31 one pair each of the split of key group 0 (query tile (w - 24) >> 2)
        constexpr int NSLOT_A = 8 * TT::N, NITEM_A = 32;
        auto item_a = [&](int w) {
            if (w < 16) {
                const int i = w >> 2, qt = (w >> 1) & 1, r0 = 2 * (w & 1);
                const float e0 = F16 ? c.t[i][qt][r0] * s_inv2 : c.t[i][qt][r0], e1 = F16 ? c.t[i][qt][r0 + 1] * s_inv2 : c.t[i][qt][r0 + 1];
                float p0 = __builtin_amdgcn_exp2f(e0 - m_sub[qt]), p1 = __builtin_amdgcn_exp2f(e1 - m_sub[qt]);
                ps[qt] += p0 + p1;
                asm volatile("" : "+v"(p0), "+v"(p1), "+v"(ps[qt]));
                c.t[i][qt][r0] = p0;
                c.t[i][qt][r0 + 1] = p1;
            } else if (w < 24) {
                const int dt = (w - 16) >> 1, qt = (w - 16) & 1;
                float a0 = o[dt][qt][0] * alpha[qt], a1 = o[dt][qt][1] * alpha[qt], a2 = o[dt][qt][2] * alpha[qt], a3 = o[dt][qt][3] * alpha[qt];
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                o[dt][qt] = f32x4t{a0, a1, a2, a3};
            } else {
                const int qt = (w - 24) >> 2, e = (w - 24) & 3;
                split_pair(c, 0, qt, e, pf[0][qt]);
                if constexpr (NPL == 3) asm volatile("" : "+v"(pf[0][qt][0][e]), "+v"(pf[0][qt][1][e]), "+v"(pf[0][qt][2][e]));
                else if constexpr (NPL == 2) asm volatile("" : "+v"(pf[0][qt][0][e]), "+v"(pf[0][qt][1][e]));
                else asm volatile("" : "+v"(pf[0][qt][0][e]));
            }
        };
        // the exponentials of key group 0 (S^T tiles 0, 1 = items 0..7) come before its split (items 24..31): items are dealt in index order
        if constexpr (MORE) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) n.t[i][j] = f32x4t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sl = 0; sl < NSLOT_A; ++sl) {
                A3_SB();
                s_slot(n, sl / (2 * TT::N), (sl / TT::N) & 1, sl % TT::N);
#pragma unroll
                for (int w = sl * NITEM_A / NSLOT_A; w < (sl + 1) * NITEM_A / NSLOT_A; ++w) item_a(w);
            }
            A3_SB();
            if constexpr (FL::VCL) {
                if (ragged) mask_tail(n, kt + 1);
            }
        } else {
#pragma unroll
            for (int w = 0; w < NITEM_A; ++w) item_a(w);
        }
        l_run[0] = l_run[0] * alpha[0] + ps[0];
        l_run[1] = l_run[1] * alpha[1] + ps[1];
        sync();
        if constexpr (MORE) {
            if constexpr (FL::KCL) { if (kt + 2 < nkt) dma_clamp(Kb, Ks, kt + 2); }
            else if constexpr (!FL::VCL) dma_full(Kb, Ks, kt + 2);
        }

        // ================= phase B: O += V(kt) P(kt) beside the split of key group 1 and the maxima of tile kt+1 =================
        // slots (g, dt, tt): 4 TT::N per key group; items: group 0's slots carry the 8 pairs of group 1, group 1's slots the 32 maxima
        float mt[2] = {A3_NEG, A3_NEG};
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int sl = 0; sl < 4 * TT::N; ++sl) {
                A3_SB();
                pv_slot(g, sl / TT::N, sl % TT::N, pf[g]);
                if (g == 0) {
                    constexpr int NS = 4 * TT::N;
#pragma unroll
                    for (int w = sl * 8 / NS; w < (sl + 1) * 8 / NS; ++w) split_pair(c, 1, w >> 2, w & 3, pf[1][w >> 2]);
                }
                if (MORE && g == 1) {
                    constexpr int NS = 4 * TT::N;
#pragma unroll
                    for (int w = sl * 16 / NS; w < (sl + 1) * 16 / NS; ++w) {         // item w: two registers of S^T tile (w >> 2, (w >> 1) & 1)
                        const int i = w >> 2, qt = (w >> 1) & 1, r0 = 2 * (w & 1);
                        mt[qt] = fmaxf(mt[qt], fmaxf(n.t[i][qt][r0], n.t[i][qt][r0 + 1]));
                    }
                    asm volatile("" : "+v"(mt[0]), "+v"(mt[1]));
                }
            }
        }
        A3_SB();
        if constexpr (MORE) {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                if constexpr (F16) mt[qt] *= s_inv2;
                m_new[qt] = fmaxf(m_run[qt], xmax(mt[qt]));
            }
            if constexpr (FL::COPY) c = n;
            sync();
            if constexpr (FL::VCL) dma_clamp(Vb, Vs, kt + 1);
            else dma_full(Vb, Vs, kt + 1);
        }
    };
    {
        using Main = A3Flags<true, false, false, false>;
        using MainC = A3Flags<true, false, false, true>;
        using KClamp = A3Flags<true, true, false, true>;
        using VClamp = A3Flags<true, false, true, true>;
        using Last = A3Flags<false, false, false, false>;
        int kt = 0;
        for (; kt + 4 < nkt; kt += 2) {
            tile(Main{}, kt, sa, sb);
            tile(Main{}, kt + 1, sb, sa);
        }
        if (nkt - kt == 4) { tile(MainC{}, kt, sa, sb); ++kt; }
        if (nkt - kt == 3) { tile(KClamp{}, kt, sa, sb); ++kt; }
        if (nkt - kt == 2) { tile(VClamp{}, kt, sa, sb); ++kt; }
        tile(Last{}, kt, sa, sb);
    }
#undef A3_SB

    // ---- normalise and store.  Lane (l15, kq) holds O[query 16 qt + l15][16 dt + 4 kq + r] in o[dt][qt][r]; one v_permlane16_swap per register
    // pair (qt = 0, 1) leaves it with 8 consecutive d of ONE query: query tile kq & 1, d = 16 dt + 8 (kq >> 1) + (0..7) ----
    float inv[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float l_tot = xsum(l_run[qt]);
        inv[qt] = F16 ? v_inv / l_tot : 1.0f / l_tot;
    }
    const int d = H * A3_DH;
    const int q_row = q_row0 + 16 * (kq & 1);
    unsigned char* o3 = reinterpret_cast<unsigned char*>(out);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[dt][0][r] * inv[0]), __float_as_uint(o[dt][1][r] * inv[1]), false, false);
            v[r] = __uint_as_float(sw[0]);
            v[4 + r] = __uint_as_float(sw[1]);
        }
        const int col = h * A3_DH + 16 * dt + 8 * (kq >> 1);
        if (q_row < n_query) {
            if constexpr (SPLIT_OUT) {
                if constexpr (F16) store_split8_h2(o3, (int64_t)b * out_tok + q_row, col, d, v, o_scale);
                else store_split8(o3, (int64_t)b * out_tok + q_row, col, d, v);
            } else {
                float* dst = out + ((int64_t)b * out_tok + q_row) * d + col;
                *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
        }
    }
}

int64_t qkv3_bytes(int B, int N, int H) { return (int64_t)3 * B * H * qkv3_npad(N) * QKV3_ROWB; }

// out3 != null: split3 image of the [B*N, H*64] result; otherwise fp32 out [B, N, H*64]
template <int TERMS>
static void attn3_launch(const unsigned char* img, float* out, void* out3, int B, int N, int Npad, int H, int n_query, int nqb, hipStream_t st,
                         float img_scale, float out_scale, int out_tok) {
    const float s_inv2 = 1.0f / (img_scale * img_scale), v_inv = 1.0f / img_scale;
    // the pipelined kernel pays for the exact three-plane modes (C3: 175 -> 170 us per launch, profiles/r04_attn_pipe.txt); with two fp16
    // planes or one bf16 plane a tile has half / a sixth of the MFMAs to hide the VALU work behind and the plain kernel is as fast or faster
    if (g_attn_m16 && (TERMS == 6 || TERMS == 9 || g_attn_m16 == 2)) {
        if (out3)
            hipLaunchKernelGGL((attn_bf16x3_p16_kernel<true, TERMS>), dim3(nqb * H * B), dim3(A3_NW * 64), 0, st, img, static_cast<float*>(out3),
                               B, N, Npad, H, n_query, nqb, s_inv2, v_inv, out_scale, out_tok);
        else
            hipLaunchKernelGGL((attn_bf16x3_p16_kernel<false, TERMS>), dim3(nqb * H * B), dim3(A3_NW * 64), 0, st, img, out, B, N, Npad, H,
                               n_query, nqb, s_inv2, v_inv, out_scale, out_tok);
        return;
    }
    if (g_attn_pipe == 2 || (g_attn_pipe == 1 && (TERMS == 6 || TERMS == 9))) {
        if (out3)
            hipLaunchKernelGGL((attn_bf16x3_pipe_kernel<true, TERMS>), dim3(nqb * H * B), dim3(A3_NW * 64), 0, st, img, static_cast<float*>(out3),
                               B, N, Npad, H, n_query, nqb, s_inv2, v_inv, out_scale, out_tok);
        else
            hipLaunchKernelGGL((attn_bf16x3_pipe_kernel<false, TERMS>), dim3(nqb * H * B), dim3(A3_NW * 64), 0, st, img, out, B, N, Npad, H,
                               n_query, nqb, s_inv2, v_inv, out_scale, out_tok);
        return;
    }
    if (out3)
        hipLaunchKernelGGL((attn_bf16x3_kernel<true, TERMS>), dim3(nqb * H * B), dim3(A3_NW * 64), 0, st, img, static_cast<float*>(out3), B, N,
                           Npad, H, n_query, nqb, s_inv2, v_inv, out_scale, out_tok);
    else
        hipLaunchKernelGGL((attn_bf16x3_kernel<false, TERMS>), dim3(nqb * H * B), dim3(A3_NW * 64), 0, st, img, out, B, N, Npad, H, n_query, nqb,
                           s_inv2, v_inv, out_scale, out_tok);
}

int attn_bf16x3(const void* qkv3, float* out, void* out3, int B, int N, int H, int n_query, int terms, hipStream_t st, float img_scale,
                float out_scale, int out_tokens) {
    const int out_tok = out_tokens > 0 ? out_tokens : N;
    AVD_REQUIRE(out_tok >= n_query, AVD_EINVAL, "attn_bf16x3: out_tokens=%d < n_query=%d", out_tok, n_query);
    AVD_REQUIRE(terms == 0 || terms == 6 || terms == 9 || terms == 1 || terms == 3, AVD_EINVAL, "attn_bf16x3: terms must be 6, 9, 1 or 3, got %d", terms);
    AVD_REQUIRE(img_scale > 0.f && img_scale < __builtin_inff() && out_scale > 0.f && out_scale < __builtin_inff() &&
                    img_scale * img_scale < __builtin_inff(), AVD_EINVAL, "attn_bf16x3: image scales must be positive and finite");
    AVD_REQUIRE(qkv3 && (out || out3), AVD_EINVAL, "attn_bf16x3: null pointer");
    AVD_REQUIRE(B > 0 && N > 0 && H > 0, AVD_EINVAL, "attn_bf16x3: bad dims B=%d N=%d H=%d", B, N, H);
    AVD_REQUIRE(n_query >= 0 && n_query <= N, AVD_EINVAL, "attn_bf16x3: n_query=%d outside [0,%d]", n_query, N);
    AVD_REQUIRE(aligned16(qkv3) && aligned16(out) && aligned16(out3), AVD_EUNSUPPORTED, "attn_bf16x3: pointers must be 16-byte aligned");
    AVD_REQUIRE(!out3 || (H * A3_DH) % 16 == 0, AVD_EUNSUPPORTED, "attn_bf16x3: split3 output needs d %% 16 == 0");
    if (n_query == 0) return AVD_OK;
    const int nqb = (n_query + 32 * A3_NW - 1) / (32 * A3_NW);
    AVD_REQUIRE((int64_t)B * H * nqb < (1ll << 31), AVD_EUNSUPPORTED, "attn_bf16x3: grid too large");
    const int tag = prof_tag_id("attn_bf16x3_kernel<%d>", terms ? terms : 6);
    ProfScope prof(tag, 4.0 * (double)B * H * (double)n_query * N * A3_DH, st);
    const int Npad = qkv3_npad(N);
    const auto* img = static_cast<const unsigned char*>(qkv3);
    if (terms == 9) attn3_launch<9>(img, out, out3, B, N, Npad, H, n_query, nqb, st, 1.f, 1.f, out_tok);
    else if (terms == 1) attn3_launch<1>(img, out, out3, B, N, Npad, H, n_query, nqb, st, 1.f, 1.f, out_tok);
    else if (terms == 3) attn3_launch<3>(img, out, out3, B, N, Npad, H, n_query, nqb, st, img_scale, out_scale, out_tok);
    else attn3_launch<6>(img, out, out3, B, N, Npad, H, n_query, nqb, st, 1.f, 1.f, out_tok);
    AVD_CHECK_LAUNCH("attn_bf16x3");
    return AVD_OK;
}

}  // namespace avd

extern "C" int64_t avd_qkv3_bytes(int B, int N, int H) {
    if (B <= 0 || N <= 0 || H <= 0) return -1;
    return avd::qkv3_bytes(B, N, H);
}
extern "C" int avd_attn_fwd_qkv_f16x2_f32(const void* qkv, float* out, void* out2, int B, int N, int H, int n_query, float qkv_scale,
                                          float out_scale, avd_stream_t stream) {
    return avd::attn_bf16x3(qkv, out, out2, B, N, H, n_query, 3, static_cast<hipStream_t>(stream), qkv_scale, out_scale);
}
extern "C" int avd_attn_fwd_qkv3_f32(const void* qkv3, float* out, void* out3, int B, int N, int H, int n_query, int terms,
                                     avd_stream_t stream) {
    return avd::attn_bf16x3(qkv3, out, out3, B, N, H, n_query, terms, static_cast<hipStream_t>(stream));
}
