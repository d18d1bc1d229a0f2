// fp32 multi-head self-attention core for the MMDiT blocks (head_dim 64, no mask, eval):
//   out[b,n,h,:] = softmax_j(q[b,n,h]·k[b,j,h] * scale) · v[b,j,h]
// Stands under nn.MultiheadAttention's scaled-dot-product step (avdiff/models/mmdt.py:51-61) between the packed
// in_proj and out_proj GEMMs.  The "cross-attention over conditioning tokens" of the task description is the
// off-diagonal block of this one joint softmax over [target ; prompt] tokens (sample_clip.py:371-375).
//
// Design (MI355X, flash-style, both contractions on v_mfma_f32_32x32x2_f32 = exact fp32):
//  * one wave owns 32 query rows; a block is NW waves sharing 64-key K/V tiles through LDS.
//  * the score tile is computed TRANSPOSED, S^T = K·Q^T, so a lane holds 16 keys of ONE query column:
//    the row max / row sum are in-register reductions plus a single cross-half shuffle.
//  * the MFMA k index may be summed in any order, so the accumulator register s of S^T (key row
//    kappa(s,half)) is used directly as the B operand of step s of O^T = V^T·P^T — P never moves between
//    lanes or through LDS; V is read from LDS row kappa(s,half), 32 consecutive floats per half-wave
//    (conflict-free ds_read_b32).
//  * K/V tiles go global -> LDS directly (global_load_lds_dwordx4: 4 rows x 256 B per wave instruction, no VGPR
//    staging).  K is stored unpadded with its 16-byte chunks XOR-swizzled by (key & 15) — applied on the per-lane
//    source address and again on the ds_read_b128 fragment read — so 16 distinct key rows hit 16 distinct slots.
//  * one LDS buffer per operand, but the loads overlap compute by phase: the S phase reads only K, so V's next
//    tile flies under it; the softmax + PV phase reads only V, so K's next tile flies under that.
//    Two barriers per tile, each preceded by a vmcnt(0) for DMA issued a whole phase earlier.
#include "avd_common.h"

namespace avd {

constexpr int ATT_DH = 64;
constexpr int ATT_KT = 64;            // keys per tile
constexpr float ATT_NEG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;

#define AVD_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AVD_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int NW, bool SPLIT = false>   // SPLIT: `out` is the split3 image of the [B*N, d] result (bf16x3 path)
__global__ __launch_bounds__(NW * 64, 2) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                              int N, int H, float scale, int n_query, int nqb,
                                                              const unsigned char* __restrict__ kpm) {
    constexpr int PPW = 16 / NW;       // 1-KiB DMA pieces (4 key rows) per wave per operand per tile
    __shared__ __attribute__((aligned(16))) float Ks[ATT_KT * ATT_DH];
    __shared__ __attribute__((aligned(16))) float Vs[ATT_KT * ATT_DH];

    // 1-D grid, XCD-aware: hardware deals consecutive block ids round-robin over the 8 XCDs; remap so each XCD gets a
    // contiguous range of work ids, i.e. the q-blocks of one (batch, head) — which re-read the same K/V — share an L2.
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
    const int d = H * ATT_DH;
    const int rs = 3 * d;
    const float* base = qkv + (int64_t)b * N * rs + h * ATT_DH;
    const float* kp = base + d;
    const float* vp = base + 2 * d;

    // ---- Q fragment: lane (q = l31, half hi) holds Q[q][32*hi + s], s = 0..31, pre-scaled by scale*log2(e) so the
    //      scores come out of the MFMA already in the exp2 domain (one multiply per score saved in the softmax) ----
    scale *= LOG2E;
    const int q_row = qb * (NW * 32) + wave * 32 + l31;
    float qf[32];
    {
        const int qr = q_row < N ? q_row : N - 1;
        const float* src = base + (int64_t)qr * rs + 32 * hi;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * c);
            qf[4 * c + 0] = t[0] * scale;
            qf[4 * c + 1] = t[1] * scale;
            qf[4 * c + 2] = t[2] * scale;
            qf[4 * c + 3] = t[3] * scale;
        }
    }

    // ---- DMA: piece p = key rows 4p..4p+3 of the tile; lane -> row 4p + lane/16, PHYSICAL 16-byte chunk lane%16 ----
    const int r4 = lane >> 4, pc = lane & 15;
    // (keep these unrolled: a rolled loop serialises the DMA issue and measured 25 % slower)
    auto dma_k = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            const int kl = p * 4 + r4;
            int key = kt * ATT_KT + kl;
            key = key < N ? key : N - 1;                       // ragged tail: finite filler, masked below
            const float* src = kp + key * rs + ((pc ^ (kl & 15)) << 2);
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(src), AVD_LDS_PTR(Ks + p * 4 * ATT_DH), 16, 0, 0);
        }
    };
    auto dma_v = [&](int kt) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + NW * i;
            int key = kt * ATT_KT + p * 4 + r4;
            key = key < N ? key : N - 1;
            const float* src = vp + key * rs + (pc << 2);
            __builtin_amdgcn_global_load_lds(AVD_GLB_PTR(src), AVD_LDS_PTR(Vs + p * 4 * ATT_DH), 16, 0, 0);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = ATT_NEG, l_run = 0.f;

    const int nkt = (N + ATT_KT - 1) / ATT_KT;
    dma_k(0);
    dma_v(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();

    const int ksw = l31 & 15;
    const int k_rd0 = l31 * ATT_DH, k_rd1 = (32 + l31) * ATT_DH;
    const bool active = qb * (NW * 32) + wave * 32 < n_query;   // wave-uniform: a wave of padding rows only loads
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = (kt + 1) < nkt;
        if (!active) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (more) dma_k(kt + 1);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            if (more) dma_v(kt + 1);
            continue;
        }

        // ---- S^T = K·Q^T for keys [0,32) and [32,64) of the tile (reads Ks only) ----
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ph = ((8 * hi + c) ^ ksw) << 2;
            const f32x4 ka = *reinterpret_cast<const f32x4*>(&Ks[k_rd0 + ph]);
            const f32x4 kb = *reinterpret_cast<const f32x4*>(&Ks[k_rd1 + ph]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[j], qf[4 * c + j], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[j], qf[4 * c + j], s1, 0, 0, 0);
            }
        }
        // K is free once every wave is here; V(kt) (issued a phase ago) has landed after the wait
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma_k(kt + 1);

        if (!more && (N & (ATT_KT - 1))) {   // ragged last tile: keys >= N contribute nothing
            const int kbase = kt * ATT_KT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kbase + mfma32_row(r, hi);
                if (key >= N) s0[r] = ATT_NEG;
                if (key + 32 >= N) s1[r] = ATT_NEG;
            }
        }
        if (kpm) {                           // key_padding_mask (mmdt.py:57-60): padded keys get no weight (wave-uniform branch)
            const unsigned char* mrow = kpm + (int64_t)b * N + kt * ATT_KT;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = mfma32_row(r, hi), key = kt * ATT_KT + kl;
                if (key < N && mrow[kl]) s0[r] = ATT_NEG;
                if (key + 32 < N && mrow[kl + 32]) s1[r] = ATT_NEG;
            }
        }

        // ---- online softmax for this lane's query column ----
        float mt = s0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, s1[r]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
            ps += s0[r] + s1[r];
        }
        if (__any(m_new > m_run)) {          // wave-uniform: skip the O rescale when no row's running max moved
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
        l_run += ps;
        m_run = m_new;

        // ---- O^T += V^T · P^T ; step s consumes key row kappa(s,hi) of each 32-key half (reads Vs only) ----
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int kr = mfma32_row(s, hi);
            const float va0 = Vs[kr * ATT_DH + l31];
            const float va1 = Vs[kr * ATT_DH + 32 + l31];
            const float vb0 = Vs[(32 + kr) * ATT_DH + l31];
            const float vb1 = Vs[(32 + kr) * ATT_DH + 32 + l31];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(va0, s0[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(va1, s0[s], o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb0, s1[s], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vb1, s1[s], o1, 0, 0, 0);
        }
        // V is free once every wave is here; K(kt+1) has landed after the wait
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        if (more) dma_v(kt + 1);
    }

    if (!active) return;      // no barriers below
    // ---- normalise and store: lane (q, hi) holds O[q][8*g + 4*hi + (0..3)] in regs 4g..4g+3 ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if constexpr (SPLIT) {
        // a split3 chunk is 8 consecutive d: the two half-lanes of a query hold 4 + 4 of every chunk -> swap so that the
        // low half-lane owns even chunks and the high half-lane odd ones, then store four 8-value chunks each
        float ch[8][4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ch[g4][e] = o0[4 * g4 + e] * inv;
                ch[4 + g4][e] = o1[4 * g4 + e] * inv;
            }
        unsigned char* img = reinterpret_cast<unsigned char*>(out);
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
            if (q_row < n_query) store_split8(img, (int64_t)b * N + q_row, h * ATT_DH + 8 * (c + hi), d, v);
        }
    } else if (q_row < n_query) {
        float* dst = out + ((int64_t)b * N + q_row) * d + h * ATT_DH + 4 * hi;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 a = {o0[4 * g4] * inv, o0[4 * g4 + 1] * inv, o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv};
            f32x4 c = {o1[4 * g4] * inv, o1[4 * g4 + 1] * inv, o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv};
            *reinterpret_cast<f32x4*>(dst + 8 * g4) = a;
            *reinterpret_cast<f32x4*>(dst + 32 + 8 * g4) = c;
        }
    }
}

template <bool SPLIT>
static int attn_launch(const float* qkv, float* out, int B, int N, int H, int Dh, float scale, int n_query, const unsigned char* kpm,
                       hipStream_t st) {
    AVD_REQUIRE(qkv && out, AVD_EINVAL, "attn: null pointer");
    AVD_REQUIRE(B > 0 && N > 0 && H > 0, AVD_EINVAL, "attn: bad dims B=%d N=%d H=%d", B, N, H);
    AVD_REQUIRE(Dh == ATT_DH, AVD_EUNSUPPORTED, "attn: head_dim %d unsupported (kernel is built for 64)", Dh);
    AVD_REQUIRE(n_query >= 0 && n_query <= N, AVD_EINVAL, "attn: n_query=%d outside [0,%d]", n_query, N);
    AVD_REQUIRE(aligned16(qkv) && aligned16(out), AVD_EUNSUPPORTED, "attn: pointers must be 16-byte aligned");
    AVD_REQUIRE((int64_t)B * H * ((N + 63) / 64) < (1ll << 31), AVD_EUNSUPPORTED, "attn: grid too large");
    AVD_REQUIRE((int64_t)N * 3 * H * Dh < (1ll << 31), AVD_EUNSUPPORTED, "attn: one sample's qkv exceeds 2^31 elements");
    if (n_query == 0) return AVD_OK;
    // 2-wave blocks (64 query rows) waste the fewest padded rows on the ragged N of this model
    // (421 -> 448); 4-wave blocks halve K/V re-reads when N is a comfortable multiple of 128.
    const int pad2 = ((n_query + 63) / 64) * 64, pad4 = ((n_query + 127) / 128) * 128;
    static const int tag4 = prof_tag_id(SPLIT ? "attn_f32_kernel<4, true>" : "attn_f32_kernel<4>"),
                     tag2 = prof_tag_id(SPLIT ? "attn_f32_kernel<2, true>" : "attn_f32_kernel<2>");
    // (measured at N=421: 4-wave blocks for every layer, padding waves skipping the arithmetic, 1.93 ms per step against
    // 1.81 ms for 2-wave blocks — the fp32 kernel prefers more, smaller blocks)
    const bool use4 = pad4 == pad2;
    ProfScope prof(use4 ? tag4 : tag2, 4.0 * (double)B * H * (double)n_query * N * ATT_DH, st);
    if (use4) {
        hipLaunchKernelGGL((attn_f32_kernel<4, SPLIT>), dim3((pad4 / 128) * H * B), dim3(256), 0, st, qkv, out, N, H, scale, n_query, pad4 / 128, kpm);
    } else {
        hipLaunchKernelGGL((attn_f32_kernel<2, SPLIT>), dim3((pad2 / 64) * H * B), dim3(128), 0, st, qkv, out, N, H, scale, n_query, pad2 / 64, kpm);
    }
    AVD_CHECK_LAUNCH("attn_f32");
    return AVD_OK;
}

int attn_f32(const float* qkv, float* out, int B, int N, int H, int Dh, float scale, int n_query, const unsigned char* kpm,
             hipStream_t st) {
    return attn_launch<false>(qkv, out, B, N, H, Dh, scale, n_query, kpm, st);
}
int attn_f32_split3(const float* qkv, void* out3, int B, int N, int H, int Dh, float scale, int n_query, hipStream_t st) {
    AVD_REQUIRE((H * Dh) % 16 == 0, AVD_EUNSUPPORTED, "attn: split3 output needs d %% 16 == 0");
    return attn_launch<true>(qkv, static_cast<float*>(out3), B, N, H, Dh, scale, n_query, nullptr, st);
}

}  // namespace avd

extern "C" int avd_attn_fwd_f32(const float* qkv, float* out, int B, int N, int H, int Dh, float scale,
                                int n_query, const uint8_t* key_padding_mask, avd_stream_t stream) {
    return avd::attn_f32(qkv, out, B, N, H, Dh, scale, n_query, key_padding_mask, static_cast<hipStream_t>(stream));
}

extern "C" int avd_attn_fwd_split3_f32(const float* qkv, void* out3, int B, int N, int H, int Dh, float scale, int n_query,
                                       avd_stream_t stream) {
    return avd::attn_f32_split3(qkv, out3, B, N, H, Dh, scale, n_query, static_cast<hipStream_t>(stream));
}
