/* avdiff_hip.h — C ABI of libavdiff_hip.so: the MI355X (gfx950) denoising hot path of
 * mauruszach/multimodal_diffusion behind the reference's Python nn.Module / sampler API.
 *
 * The reference has no FFI/plugin layer: its boundary for this path is a set of
 * Python call signatures.  Each entry point below names the reference call it stands under
 * (paths relative to the reference repo).  The host side (the multimodal_diffusion_amd package) binds these
 * with ctypes and keeps the reference's class names, constructor kwargs, forward signatures and
 * state_dict keys.
 *
 * Conventions
 *   - every function returns 0 on success, a negative AVD_E* code on failure;
 *     avd_last_error() returns a thread-local human-readable message for the last failure.
 *   - all tensor pointers are DEVICE pointers owned by the caller, fp32, contiguous unless a leading
 *     dimension is given, 16-byte aligned; int64 for timesteps (as in the reference).
 *   - nothing allocates, synchronises or copies to the host: every call only enqueues kernels on
 *     `stream` (a hipStream_t passed as void*), so calls are stream-ordered and hipGraph-capturable.
 *   - inputs are never written; outputs never alias inputs unless stated.
 */
#ifndef AVDIFF_HIP_H
#define AVDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVD_ABI_VERSION 7

#define AVD_OK            0
#define AVD_EINVAL       -1   /* bad shape / argument (reference: AssertionError / ValueError) */
#define AVD_EUNSUPPORTED -2   /* shape outside what the gfx950 kernels are built for */
#define AVD_ELAUNCH      -3   /* HIP launch error */
#define AVD_EWORKSPACE   -4   /* workspace too small */

#define AVD_ACT_NONE 0
#define AVD_ACT_GELU 1        /* exact erf GELU (torch.nn.functional.gelu default) */
#define AVD_ACT_SILU 2
#define AVD_ACT_TANH 3        /* conv1d only (AudioCodec.decode output) */
#define AVD_ACT_RELU 4        /* avd_layernorm_act_f32 / noise head only (heads/noise_heads.py:28-36) */
#define AVD_ACT_LEAKY_RELU 5  /* slope 0.1, same scope */

typedef void* avd_stream_t;   /* hipStream_t */

int         avd_abi_version(void);
const char* avd_last_error(void);
/* gfx arch name of device 0 as seen by the library ("gfx950"); diagnostic only. */
int         avd_device_arch(char* buf, int buflen);
/* Measurement / test hooks, process-wide, never needed for correct results: "gemm_tile" (-1 auto, 0 = 128x128, 1 = 128x64,
 * 2 = 64x64 LDS-DMA tile of avd_gemm_bias_act_f32), "gemm_stages" (0 by size, 2 / 3 LDS stages), "gemm_splitk" (largest number of K slices of the fp32 fc2 launch of
 * avd_core_forward_f32 when its 64x64 blocks cover less than half of the CUs — batches of a few hundred rows; 0 = never, default and maximum 4;
 * partial sums added in slice order, no atomics), "s3_tile" (-1 per epilogue, 0 = 8-wave
 * 256x256, 1 = 4-wave 256x128 blocks of the split-operand GEMMs), "s3_stagger" (first-generation stagger of co-resident 4-wave blocks,
 * x 1024 cycles; -1 automatic), "cfg_rows" (1 default: the fused CFG + un-patch + DDIM kernel reads token rows whole and writes whole 128-byte lines of latent through an LDS
 * transpose; 0: one 16-byte gather per lane; bit-identical), "vae_lat" (1 default: avd_vae_decode_f32 takes the latent-composed first convolution when the descriptor
 * carries conv0_lat_w3; 0: from_lat -> upsample -> 64-channel convolution), "vae_fold" (1 default: the three-plane two-block decoder hands conv 0's output to conv 1 as its operand image, folds the GroupNorm
 * between them into conv 1's per-sample weights and finishes to_img from per-group partial sums; 0: fp32 activations between the kernels), "codec_mfma" (1 default: avd_conv1d_act_f32 runs 64 -> 64 layers with k = 7 / 9 on the fp32 matrix pipe; 0: the vector kernel), "s3_sn" / "s3_super4" / "s3_super8" (super-tile of the split GEMMs' block order — the blocks an XCD runs together: width in column blocks /
 * blocks per super-tile of the two-per-CU / one-per-CU kernels; 0 = default; bit-identical, measurement aid of profiles/r05_fetch_ab.txt), "s3_min_rows" (smallest 2B*N that takes the split-operand kernels: -1 default = 2,048 rows for the core in the six-term bf16x3 mode, 6,144 otherwise; >= 0 = that many in every mode), "no_fold" (1 = keep RMSNorm as
 * separate kernels in avd_core_forward_f32, in every mode), "s3_m16" (1 default: bf16x3 GEMMs on v_mfma_f32_16x16x32_bf16 with two product
 * terms per instruction; 0: the 32x32x16 kernel), "s3_rt" (rows per block of the bf16x3 residual + image epilogue: 0 automatic,
 * 7 = 224 rows, 8 = 256 rows, 6 = 192 rows (four-wave kernel only); results are bit-identical), "s3_rt4" (rows per 256x128-class four-wave
 * block of the bf16x3 in_proj / fc1 / out_proj / fc2 launches, in units of 32: 0 = the host picks per launch what leaves the fewest CUs idle —
 * mid-size and small batches — 2 .. 8 forced (in_proj / fc1: at least 5); bit-identical), "s3_deep4" (1 default: an out_proj / fc2 launch whose four-wave blocks fit the CUs once runs
 * one block per CU on a four-stage LDS ring instead of a two-stage one; 0: never; bit-identical), "s3_w128" (1 default: the bf16x3 residual + image GEMMs run as four waves
 * with a 128 x 128 wave tile and accumulators in AGPRs; 0: eight waves with 128 x 64 tiles; bit-identical), "s3_splitk" (largest number of K slices of the fc2 launch when its blocks
 * fill at most half of the chip's block slots — small and mid-size batches; 0 = never, default and maximum 4; partial sums are added in
 * slice order by a reduction kernel, no atomics), "attn_pipe" (1 default: the three-plane split-operand attention runs as one software
 * pipeline per wave — the next tile's score MFMAs beside this tile's softmax; 0: the plain kernel everywhere, 2: the pipeline in every
 * split mode), "attn_m16" (1 default: that pipeline on v_mfma_f32_16x16x32 in the three-plane modes — same cycles per FLOP, less power, so a higher clock under the cap; 0: the 32x32x16 kernels; 2: the 16x16x32 kernel in every split mode), "core_trim" (1 default: the last block of avd_core_forward_f32 runs out_proj / fc1 / fc2 / the final norm on the caller's
 * row window only; 0: on every row; the window's results are bit-identical), "mlp_fused" (0 default; 1: fc1 -> GELU -> fc2 of the six-term
 * bf16x3 path as ONE launch whose hidden activations stay on the CU — d = 512 only; bit-identical to the two launches, and measures
 * 1.5x slower: DESIGN.md 4.9). */
int         avd_tune_set(const char* key, int64_t value);

/* ---- a6: RMSNorm — avdiff/models/mmdt.py:33-42 (RMSNorm.forward)
 * y = scale * x / (||x||_2 / sqrt(d) + eps), eps OUTSIDE the sqrt. x,y: [rows,d]. */
int avd_rmsnorm_f32(const float* x, const float* scale, float* y, int64_t rows, int d, float eps,
                    avd_stream_t stream);

/* ---- a3/a6/a7: Linear (+bias)(+act)(+residual) — torch.nn.Linear as used at
 * avdiff/models/mmdt.py:60 (packed in_proj / out_proj inside nn.MultiheadAttention), :77-83 (MLP fc1/fc2),
 * avdiff/models/heads/noise_heads.py:206-223, avdiff/models/infer/sample_clip.py:54-56 (LinearAdapter).
 * C[M,N] = act(A[M,K] · W[N,K]^T + bias[N]) + residual[M,N].   bias/residual may be NULL.
 * lda/ldr/ldc are row strides in floats (multiples of 4).  K % 4 == 0, N % 4 == 0.
 * f32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products, fp32 accumulate. */
int avd_gemm_bias_act_f32(const float* A, int64_t lda, const float* W, const float* bias,
                          const float* residual, int64_t ldr, float* C, int64_t ldc,
                          int64_t M, int N, int K, int act, avd_stream_t stream);

/* The same Linear with the neighbouring RMSNorms (mmdt.py:39-42) folded in, as avd_core_forward_f32 uses it (A [M,K], C [M,N]
 * contiguous; K % 32 == 0, N % 128 == 0):
 *   ss_in  != NULL: A is the UN-normalised stream and W carries the norm's scale (W * scale[None,:]); every output row is
 *                   multiplied by 1 / (sqrt(sum_j ss_in[row][j]) / sqrt(K) + eps) before the bias.  ss_in: [M, ss_in_cols].
 *   ss_out != NULL: the epilogue also writes the sum of squares of each final output row per 32-column chunk, [M, N/32]
 *                   (fixed summation order: deterministic) — the table the next folded Linear reads as ss_in. */
int avd_gemm_rmsfold_f32(const float* A, const float* W, const float* bias, const float* residual, float* C, int64_t M,
                         int N, int K, int act, const float* ss_in, int ss_in_cols, float eps, float* ss_out,
                         avd_stream_t stream);

/* ---- a6: multi-head self-attention core — nn.MultiheadAttention(batch_first=True) at
 * avdiff/models/mmdt.py:51-61 between in_proj and out_proj: out = softmax(q k^T * scale) v per head,
 * no mask, eval.  qkv: [B,N,3*H*Dh] (q | k | v along the last dim, heads contiguous inside each),
 * out: [B,N,H*Dh].  Dh must be 64.  n_query <= N limits the query rows computed (rows >= n_query of
 * `out` are left untouched); pass N for the reference behaviour.
 * key_padding_mask: NULL, or bytes [B,N] with non-zero = this key is padding and gets no attention weight
 * (MMDiT.forward's key_padding_mask, mmdt.py:57-60,134-149).  A sample whose keys are ALL padded is undefined in the reference
 * (NaN); here it attends uniformly. */
int avd_attn_fwd_f32(const float* qkv, float* out, int B, int N, int H, int Dh, float scale,
                     int n_query, const uint8_t* key_padding_mask, avd_stream_t stream);

/* ---- a7: LayerNorm(d, eps, affine) followed by an activation — the Linear→LayerNorm→GELU block of
 * avdiff/models/heads/noise_heads.py:141-147.  x,y: [rows,d]. */
int avd_layernorm_act_f32(const float* x, const float* gamma, const float* beta, float* y,
                          int64_t rows, int d, float eps, int act, avd_stream_t stream);

/* ---- a4: sinusoidal timestep embedding — avdiff/utils/schedule_utils.py:64-86.
 * out[B,dim] = [cos(t*f) | sin(t*f)], f_i = exp(-ln(max_period)*i/half); odd dim zero-padded.
 * freqs: optional device table [dim/2] of f_i built by the host exactly as the reference builds it
 * (cos/sin at t ~ 1000 turn a 1-ulp difference in exp into 6e-5); NULL computes f_i in the kernel. */
int avd_timestep_embedding_f32(const int64_t* t, const float* freqs, float* out, int B, int dim, float max_period,
                               avd_stream_t stream);

/* ---- a1 / a8: tube patch / unpatch — avdiff/utils/ops.py:100-119, 122-144.
 * z: [B,C,T,H,W]  <->  tok: [B, (T/t)(H/h)(W/w), C*t*h*w].  w % 4 == 0 and W % 4 == 0. */
int avd_tube_patch_f32(const float* z, float* tok, int B, int C, int T, int H, int W, int t, int h, int w,
                       avd_stream_t stream);
int avd_tube_unpatch_f32(const float* tok, float* z, int B, int C, int T, int H, int W, int t, int h, int w,
                         avd_stream_t stream);

/* ---- a2 / a8': audio chunk tokens and their overlap-add inverse —
 * avdiff/models/infer/sample_clip.py:184-188 and :191-215 (-> avdiff/utils/ops.py:17-45, 48-93).
 * z: [B,Ca,F] -> tok: [B,Na,Ca*len], Na = (F-len)/stride + 1.  Inverse: overlap-add weighted by `window` [len] and divided
 * by the summed weights (window == NULL: rectangular, i.e. the overlap count; a Hann table gives ops.py's apply_hann=True),
 * cropped / zero-padded to F frames. */
int avd_audio_tokens_f32(const float* z, float* tok, int B, int Ca, int F, int len, int stride,
                         avd_stream_t stream);
int avd_audio_untokens_f32(const float* tok, const float* window, float* z, int B, int Ca, int F, int len, int stride,
                           avd_stream_t stream);

/* ---- a8: DDIM update — avdiff/utils/schedule_utils.py:146-200 (ddim_step).
 * x_t, eps_hat, x_prev: [B, per_sample]; t_now,t_prev: int64[B] (t_prev may be -1 => abar_prev = 1);
 * alpha_bar: fp32[T_train].  eta > 0 requires `noise` (same shape as x_t); eta == 0 ignores it. */
int avd_ddim_step_f32(const float* x_t, const float* eps_hat, const int64_t* t_now, const int64_t* t_prev,
                      const float* alpha_bar, int T_train, float eta, const float* noise,
                      float* x_prev, int B, int64_t per_sample, avd_stream_t stream);

/* ---- a8 fused: CFG combine + tube un-patch + DDIM — avdiff/models/infer/sample_clip.py:381-389.
 * eps2: [2B,Nv,C*t*h*w] (cond batch then null batch); eps = null + g*(cond-null); un-patched on the fly.
 * z, z_out: [B,C,T,H,W]. */
int avd_cfg_unpatch_ddim_f32(const float* eps2, const float* z, const int64_t* t_now, const int64_t* t_prev,
                             const float* alpha_bar, int T_train, float guidance, float eta,
                             const float* noise, float* z_out,
                             int B, int C, int T, int H, int W, int t, int h, int w, avd_stream_t stream);

/* ---- a8' fused: CFG combine + audio overlap-add + DDIM — sample_clip.py:342-348.
 * eps2: [2B,Na,Ca*len]; z,z_out: [B,Ca,F]. */
int avd_cfg_untoken_ddim_audio_f32(const float* eps2, const float* z, const int64_t* t_now,
                                   const int64_t* t_prev, const float* alpha_bar, int T_train,
                                   float guidance, float eta, const float* noise, float* z_out,
                                   int B, int Ca, int F, int len, int stride, avd_stream_t stream);

/* ---- a1+a3+a4+a5 fused front end — sample_clip.py:363-371,377 (A->V) / :322-333,338 (V->A).
 * Builds the CFG-stacked sequence X2[2B, Nt+Np, d] in one pass:
 *   target rows : [ adapter(tokens(z_target)) | temb(t_now[b]) ]   (same in both halves)
 *   prompt rows : Xp[b] in the cond half, zeros in the null half (whole d-wide rows, as the reference)
 * target_kind 0 = video latent [B,C,T,H,W] with tubes (p0,p1,p2)=(t,h,w); 1 = audio latent [B,Ca,F]
 * with chunk (p0,p1)=(len,stride).  target_first != 0 puts target rows before prompt rows
 * (the reference's order is always [video ; audio]).
 * Xp: [B,Np,d] = adapter(prompt tokens) | temb(0), computed once per run by the caller.
 * tok_ws: scratch [B*Nt, tok_dim].  Wt: [d-tdim, tok_dim], bt: [d-tdim]. */
typedef struct {
    int target_kind;      /* 0 video, 1 audio */
    int target_first;
    int B, d, tdim;
    int C, T, H, W;       /* video latent dims (target_kind 0) — or Ca=C, F=T for audio */
    int p0, p1, p2;
    int Nt, Np;
    const float* temb_freqs;   /* optional device table [tdim/2], see avd_timestep_embedding_f32; may be NULL */
    int temb_add;              /* 0: concat [adapter(d-tdim) | temb(tdim)] as the sampler (sample_clip.py:59-70);
                                  1: adapter(d) + temb(d) as the trainer (train/trainer.py:45-49); needs tdim == d */
} avd_embed_desc;
/* floats of scratch `tok_ws` must hold (tokens + the [B,tdim] timestep embedding); -1 on a bad descriptor */
int64_t avd_embed_workspace_floats(const avd_embed_desc* desc);
int avd_embed_cfg_pair_f32(const avd_embed_desc* desc, const float* z_target, const float* Wt, const float* bt,
                           const int64_t* t_now, const float* Xp, float* tok_ws, float* X2,
                           avd_stream_t stream);

/* ---- composites: whole modules / the whole step as one host call (stateless; weights by pointer table).
 * These enqueue exactly the kernels above in order; they exist to keep the per-step host cost at one FFI
 * call and to make a step one hipGraph-capturable unit. */
typedef struct {                       /* avdiff/models/mmdt.py:88-99 (Block) state_dict, device ptrs */
    const float* norm1_scale;          /* blocks.{i}.norm1.scale (norm="layernorm": .weight) [d] */
    const float* in_proj_weight;       /* blocks.{i}.attn.mha.in_proj_weight  [3d,d]   */
    const float* in_proj_bias;         /* blocks.{i}.attn.mha.in_proj_bias    [3d]     */
    const float* out_proj_weight;      /* blocks.{i}.attn.mha.out_proj.weight [d,d]    */
    const float* out_proj_bias;        /* blocks.{i}.attn.mha.out_proj.bias   [d]      */
    const float* norm2_scale;          /* blocks.{i}.norm2.scale              [d]      */
    const float* fc1_weight;           /* blocks.{i}.mlp.fc1.weight           [hid,d]  */
    const float* fc1_bias;             /* blocks.{i}.mlp.fc1.bias             [hid]    */
    const float* fc2_weight;           /* blocks.{i}.mlp.fc2.weight           [d,hid]  */
    const float* fc2_bias;             /* blocks.{i}.mlp.fc2.bias             [d]      */
    /* optional split3 images of the four weights above (avd_split3_f32); all four non-NULL in every block selects
     * the bf16x3 matmul path of avd_core_forward_f32 for large batches (see "bf16x3" below), NULL keeps fp32 MFMA */
    /* optional: in_proj_weight * norm1.scale[None,:] and fc1_weight * norm2.scale[None,:] (fp32, same shapes).  Both non-NULL
     * lets the fp32 path fold each RMSNorm into its neighbours: the preceding residual epilogue emits the rows' sums of
     * squares, the following Linear runs on the un-normalised stream with these weights and scales its rows by 1/rms. */
    const float* in_proj_weight_n;
    const float* fc1_weight_n;
    const void* in_proj_weight3;
    const void* out_proj_weight3;
    const void* fc1_weight3;
    const void* fc2_weight3;
    const float* norm1_bias;           /* norm="layernorm" only: blocks.{i}.norm1.bias / norm2.bias [d]; NULL for RMSNorm */
    const float* norm2_bias;
    /* split_terms == 3 ("f16x2") only: power-of-two scales of the fp16 operand images (see "f16x2" below).
     * [0..3] the weight images in_proj, out_proj, fc1, fc2 (the *_weight3 pointers then hold avd_split_f16x2_f32 images);
     * [4..7] the activation images: norm1 output, q|k|v (and the attention output), norm2 output, GELU(fc1) output.  Each
     * activation scale must satisfy scale * bound <= 2^15 for a bound on the magnitudes the image can hold. */
    float f16x2_scale[8];
    /* optional (bf16 plane modes, i.e. split_terms 6 / 9 / 1): split3 images of in_proj_weight_n and fc1_weight_n above.  Both non-NULL
     * in every block folds each RMSNorm of the split path into its neighbours: the residual epilogues of out_proj / fc2 also write the
     * stream's operand image and its rows' sums of squares, in_proj / fc1 run on that un-normalised image with these weights and scale
     * their rows by 1 / (rms + eps) before the bias — no RMSNorm kernel between the first block's input and the final norm
     * (mmdt.py:39-42, 95-99; same arithmetic up to rounding). */
    const void* in_proj_weight3n;
    const void* fc1_weight3n;
} avd_block_weights;

typedef struct {                       /* avdiff/models/mmdt.py:116-149 (MMDiT) */
    int d, n_layers, n_heads, mlp_hidden;
    float norm_eps;                    /* 1e-6 */
    const avd_block_weights* blocks;   /* HOST array [n_layers] of device-pointer tables */
    const float* final_norm_scale;     /* final_norm.scale [d] */
    int norm_kind;                     /* 0: RMSNorm (every shipped config); 1: nn.LayerNorm (build_norm, mmdt.py:44-45) — fp32 path only */
    const float* final_norm_bias;      /* final_norm.bias [d] for norm_kind 1, else NULL */
    int split_terms;                   /* bf16x3 path only: product terms kept per k — 0 or 6: default (fp32-level error), 9: strict
                                        * (nothing dropped), 1: plain bf16 operands (reduced precision, BASELINE config C2),
                                        * 3: f16x2 — two scaled fp16 planes per operand, three terms (22-bit operands, fp32
                                        * accumulation; needs avd_block_weights.f16x2_scale) */
    int attn_mode;                     /* bf16x3 path only: 0 = attention follows split_terms; 1 = fp8 (OCP e4m3) QK^T and PV with fp32
                                        * accumulation (csrc/attn_fp8.hip) — reduced precision, BASELINE config C5, reported error */
} avd_core_weights;

typedef struct {                       /* avdiff/models/heads/noise_heads.py:94-229, one modality path */
    int d_in, hidden, d_out, n_shared;
    float ln_eps;                      /* 1e-5 */
    int act;                           /* AVD_ACT_GELU */
    const float* input_proj_weight;    /* input_proj.{m}.weight [hidden,d_in] */
    const float* input_proj_bias;
    const float* const* shared_lin_weight;  /* HOST array [n_shared]: shared.{j}.0.weight [hidden,hidden] */
    const float* const* shared_lin_bias;
    const float* const* shared_ln_weight;   /* shared.{j}.1.weight [hidden] */
    const float* const* shared_ln_bias;
    const float* out_proj_weight;      /* out_proj.{m}.weight [d_out,hidden] */
    const float* out_proj_bias;
    /* optional split-operand mode of the head's Linears (same meaning as avd_core_weights.split_terms; 0 = fp32 MFMA).  Taken when
     * every image pointer below is non-NULL, d_in % 16 == 0, hidden % 256 == 0, d_out % 256 == 0 and there are >= 6144 rows. */
    int split_terms;
    const void* input_proj_weight3;    /* operand images of the weights above (avd_split3_f32, or avd_split_f16x2_f32 for terms 3) */
    const void* const* shared_lin_weight3;
    const void* out_proj_weight3;
    /* split_terms == 3 only, HOST array of 2 * (n_shared + 2) power-of-two scales: the weight images [input_proj, shared 0.., out_proj],
     * then the activation images [head input rows, input_proj output, LayerNorm+act output 0..] (the caller bounds the input rows;
     * the rest follows from the weights, see multimodal_diffusion_amd/noise_heads.py) */
    const float* f16x2_scale;
} avd_head_weights;

/* ---- "bf16x3": fp32-accurate Linear on the bf16 matrix pipe (same reference ops as avd_gemm_bias_act_f32:
 * avdiff/models/mmdt.py:60,77-83).  Every fp32 operand is split exactly into three bf16 planes (x = h + m + l); a
 * product keeps the six terms down to 2^-16 and accumulates them in fp32, so the result carries the error of an fp32
 * FMA chain (measured slightly below it) while the matrix pipe runs 2.67x fewer cycles than with fp32 MFMA.
 * Operands travel as "split3 images" (tiled, 6 bytes per element, rows padded to 256; layout in csrc/gemm_bf16x3.hip).
 * `terms` selects the product terms kept per k: 6 (or 0) the default above; 9 strict — all nine, nothing dropped; 1 — only the
 * high planes, i.e. plain bf16 operands with fp32 accumulation: the reduced-precision variant BASELINE config C2 names, whose
 * error is REPORTED against the fp32 oracle and which is never a parity path.
 * Domain: finite operands with |x| >= ~2^-110 or 0 split exactly (below that the lower planes leave bf16's range and the
 * absolute error per product is < 2^-126); +-inf / NaN in an operand row make that output row non-finite (NaN where an
 * fp32 chain would give +-inf). */
int64_t avd_split3_bytes(int64_t rows, int K);                     /* bytes of the image of a [rows,K] matrix; -1 if K % 16 */
int avd_split3_f32(const float* x, void* out, int64_t rows, int K, avd_stream_t stream);   /* x [rows,K] contiguous */
/* RMSNorm (mmdt.py:39-42) whose output is written as a split3 image (the A operand of the next Linear) */
int avd_rmsnorm_split3_f32(const float* x, const float* scale, void* out, int64_t rows, int d, float eps,
                           avd_stream_t stream);
/* avd_attn_fwd_f32 whose [B*N, H*Dh] result is written as a split3 image (rows >= n_query of a sample are left untouched) */
int avd_attn_fwd_split3_f32(const float* qkv, void* out3, int B, int N, int H, int Dh, float scale, int n_query,
                            avd_stream_t stream);
/* bf16x3 attention (csrc/attn_bf16x3.hip): in_proj writes q, k, v as a "qkv3 image" (three bf16 planes per value, per
 * (part, sample, head) rows of 384 B, q pre-multiplied by qscale = softmax scale * log2 e), the attention kernel reads it.
 * Same reference op as avd_attn_fwd_f32 (mmdt.py:51-61), same fp32-level error. */
int64_t avd_qkv3_bytes(int B, int N, int H);                       /* bytes of the image for [B,N,3*H*64] */
/* qkv = A W^T + bias, A3/W3 split3 images of A [M,K] (M = B*tokens rows) and in_proj_weight [3*heads*64, K] */
int avd_gemm_bf16x3_qkv3_f32(const void* A3, const void* W3, const float* bias, void* qkv3, int64_t M, int tokens,
                             int heads, int K, float qscale, int terms, avd_stream_t stream);
/* softmax(q k^T) v from a qkv3 image; out3 == NULL: fp32 out [B,N,H*64]; else the split3 image of [B*N, H*64].
 * Rows >= n_query of every sample are not computed and left untouched. */
int avd_attn_fwd_qkv3_f32(const void* qkv3, float* out, void* out3, int B, int N, int H, int n_query, int terms,
                          avd_stream_t stream);
/* fp8 (OCP e4m3) attention from the same qkv3 image (csrc/attn_fp8.hip): both contractions on v_mfma_f32_32x32x16_fp8_fp8 with
 * fp32 accumulation and fp32 softmax.  Reduced precision — BASELINE config C5 names it; the reference has no such path
 * (infer/sample_clip.py:399-411), so its error is reported against the fp32 result, never gated as parity.
 * workspace: avd_attn_fp8_workspace_bytes(B, N, H) bytes (the quantised, tile-major Q / K / V^T images). */
int64_t avd_attn_fp8_workspace_bytes(int B, int N, int H);
int avd_attn_fwd_fp8_f32(const void* qkv3, void* workspace, int64_t workspace_bytes, float* out, void* out3, int B, int N, int H,
                         int n_query, avd_stream_t stream);
/* the same with an f16x2 q|k|v image at scale qkv_scale (avd_gemm_f16x2_qkv_f32); out2 != NULL: the result as an f16x2 image at out_scale */
int avd_attn_fwd_fp8_f16x2_f32(const void* qkv, void* workspace, int64_t workspace_bytes, float* out, void* out2, int B, int N,
                               int H, int n_query, float qkv_scale, float out_scale, avd_stream_t stream);
/* C = act(A W^T + bias) (+ residual), A3/W3 split3 images of A [M,K] and W [N,K]; N % 256 == 0, K % 16 == 0.
 * C3 == NULL: fp32 row-major C [M,N], act AVD_ACT_NONE, residual optional (may alias C).
 * C3 != NULL: the result is written as the split3 image of [M,N] instead (bias required, act AVD_ACT_NONE or AVD_ACT_GELU,
 * no residual). */
int avd_gemm_bf16x3_f32(const void* A3, const void* W3, const float* bias, const float* residual, float* C, void* C3,
                        int64_t M, int N, int K, int act, int terms, avd_stream_t stream);

/* ---- "f16x2": the same Linear / attention (mmdt.py:51-61,77-83) with every operand held as TWO fp16 planes, x ~ (h + l) / s
 * with h = rn_f16(s x), l = rn_f16(s x - h), and three product terms hh + hl + lh accumulated in fp32 on
 * v_mfma_f32_32x32x16_f16: half the matrix-pipe work of bf16x3.  An operand carries 22 significant bits (error <= 2^-22
 * relative) and the dropped ll term is <= 2^-22 relative, so a product is accurate to ~7e-7 against fp32's 6e-8 rounding;
 * over a dot product these errors add like the fp32 accumulation rounding both modes share (measured error in DESIGN.md).
 * fp16 has 5 exponent bits, so an image is stored at a power-of-two `scale` with |scale * x| <= 2^15 for every element:
 * the CALLER supplies the scale from a bound on |x| (for the MMDiT core the bounds follow from the weights alone, see
 * multimodal_diffusion_amd/mmdt.py `_f16x2_scales`); a value past the range turns its output rows into NaN, never into a
 * silently saturated number.  Images have the split3 / qkv3 geometry (same byte counts; the third plane is unused).
 * ab_scale = (A image scale) * (W image scale); c_scale / qkv_scale / out_scale = scale of the image being written. */
/* out2[0] = max |w|, out2[1] = max over rows of ||w_row||_2 for w [rows, cols] (a vector: rows = 1) — the quantities the scale
 * bounds above are made of; exact, order-independent (integer atomic maxima); NaN in w makes both NaN. */
int avd_weight_bounds_f32(const float* w, int64_t rows, int cols, float* out2, avd_stream_t stream);
int avd_split_f16x2_f32(const float* x, void* out, int64_t rows, int K, float scale, avd_stream_t stream);
int avd_rmsnorm_split_f16x2_f32(const float* x, const float* gamma, void* out, int64_t rows, int d, float eps, float scale,
                                avd_stream_t stream);
int avd_gemm_f16x2_f32(const void* A2, const void* W2, const float* bias, const float* residual, float* C, void* C2,
                       int64_t M, int N, int K, int act, float ab_scale, float c_scale, avd_stream_t stream);
int avd_gemm_f16x2_qkv_f32(const void* A2, const void* W2, const float* bias, void* qkv, int64_t M, int tokens, int heads,
                           int K, float qscale, float ab_scale, float qkv_scale, avd_stream_t stream);
int avd_attn_fwd_qkv_f16x2_f32(const void* qkv, float* out, void* out2, int B, int N, int H, int n_query, float qkv_scale,
                               float out_scale, avd_stream_t stream);

/* bytes of scratch avd_core_forward_f32 needs for a [B,N,d] input */
int64_t avd_core_workspace_bytes(const avd_core_weights* w, int B, int N);
/* MMDiT.forward(x) -> y, x,y: [B,N,d] (y may alias x).  n_out_rows: number of leading rows per sample whose
 * output is needed (N = reference behaviour; fewer lets the last block skip dead rows when the caller only
 * consumes the first n_out_rows — the engine passes the target-row count). out_row0: first needed row.  Rows of y outside
 * [out_row0, out_row0 + n_out_rows) are unspecified: with the window at row 0 the last block's attention, out_proj, fc1, fc2 and the final
 * norm run on the window's rows only (six-term bf16-plane path).
 * key_padding_mask: NULL or bytes [B,N], see avd_attn_fwd_f32.  A mask, or norm_kind 1 (LayerNorm), keeps the whole forward on the fp32
 * MFMA kernels whatever split_terms says — the split-operand attention takes no mask and the split producers are RMSNorm's; results
 * are the fp32 path's, at its speed.  attn_mode 1 (fp8 attention) with either of them is refused (AVD_EUNSUPPORTED). */
int avd_core_forward_f32(const avd_core_weights* w, const float* x, float* y, int B, int N,
                         int out_row0, int n_out_rows, const uint8_t* key_padding_mask, void* workspace,
                         int64_t workspace_bytes, avd_stream_t stream);

int64_t avd_head_workspace_bytes(const avd_head_weights* w, int64_t rows);
/* MultiModalNoiseHead path for ONE modality over `rows` token rows taken from h with segmented addressing:
 * row r lives at h + (r / seg_rows) * seg_stride + (r % seg_rows) * ldh.  out: [rows, d_out] contiguous. */
int avd_head_forward_f32(const avd_head_weights* w, const float* h, int64_t ldh, int64_t seg_rows,
                         int64_t seg_stride, int64_t rows, float* out, void* workspace,
                         int64_t workspace_bytes, avd_stream_t stream);

typedef struct {                       /* one whole CFG denoising step (sample_clip.py:359-389 / 318-348) */
    avd_embed_desc embed;
    const avd_core_weights* core;
    const avd_head_weights* head;      /* the TARGET modality's path */
    const float* adapt_w; const float* adapt_b;   /* target adapter */
    const float* alpha_bar; int T_train;
    float guidance, eta;
    int split_streams;   /* !=0: run the cond and null CFG halves as two kernel chains on two streams (fork/join by events;
                            graph-capturable); results are bit-identical to the single-stream order */
} avd_step_desc;
int64_t avd_step_workspace_bytes(const avd_step_desc* s);
/* z_out = DDIM(z, eps_cfg(z, Xp, t_now), t_now -> t_prev).  z_out must not alias z. */
int avd_denoise_step_f32(const avd_step_desc* s, const float* z, const float* Xp, const int64_t* t_now,
                         const int64_t* t_prev, const float* noise, float* z_out,
                         void* workspace, int64_t workspace_bytes, avd_stream_t stream);

/* ---- a9 / next-1: VideoVAE.decode — avdiff/models/encoders/vae_video3d.py:195-214 (decode), :79-84
 * (_conv_block_3d: Conv3d 3x3x3 pad 1 -> GELU(erf) -> GroupNorm(min(8,C), eps 1e-5, affine)), :108-119.
 * z [B,Cv,Tp,Hp,Wp] NCDHW -> from_lat (1x1x1) -> trilinear upsample (align_corners=False) to (T,H,W) ->
 * n_blocks x [conv3x3x3 + GELU + GroupNorm] -> to_img (1x1x1) -> sigmoid | tanh -> out [B,out_ch,T,H,W] NCDHW.
 * Inside, activations are NDHWC in a zero-haloed buffer and the convolution is the fp32 MFMA GEMM with a
 * per-tap address shift (no im2col).  Decoder width must be 64 (the reference default). */
typedef struct {
    int B, Cv, Tp, Hp, Wp;             /* latent dims */
    int T, H, W;                       /* output size (reference default: Tp*t_down, Hp*s_down, Wp*s_down) */
    int base, n_blocks, out_ch;        /* dec_base (64), dec_blocks, in_ch of the VAE (3) */
    int out_tanh;                      /* 0 = sigmoid, 1 = tanh (cfg.out_activation) */
    float gn_eps;                      /* 1e-5 */
    const float* from_lat_w;           /* from_lat.weight  [base,Cv]   (1x1x1 kernel squeezed) */
    const float* from_lat_b;           /* from_lat.bias    [base] */
    const float* const* conv_w;        /* HOST array [n_blocks]: dec_net.{i}.0.weight re-laid as [out][kt][kh][kw][in] */
    const float* const* conv_b;        /* dec_net.{i}.0.bias [base] */
    const float* const* gn_w;          /* dec_net.{i}.2.weight [base] */
    const float* const* gn_b;          /* dec_net.{i}.2.bias   [base] */
    const float* to_img_w;             /* to_img.weight [out_ch,base] */
    const float* to_img_b;             /* to_img.bias   [out_ch] */
    const void* const* conv_w3;        /* optional HOST array [n_blocks] of avd_conv3_weight_f32 images: the convolutions then run
                                        * on the bf16 matrix pipe with exactly split operands (fp32-level error, see "bf16x3"); NULL = fp32 MFMA */
    int conv_terms;                    /* with conv_w3: 0 or 6 = bf16x3; 3 = f16x2 (images from avd_conv3_weight_f16x2_f32, see "f16x2") */
    const float* conv_w_scale;         /* conv_terms 3: HOST array [n_blocks], power-of-two scales of the weight images */
    const float* conv_a_scale;         /* conv_terms 3: HOST array [n_blocks], scales of each block's INPUT image: entry i >= 1 from the bound
                                        * |GroupNorm output| <= sqrt(n - 1) max|gamma| + max|beta| (n = elements of one group of one sample);
                                        * entry 0 is ignored — the first image's scale is derived on the device from max |from_lat(z)| */
    /* ABI 6 — the first convolution composed with what precedes it (vae_video3d.py:205-209: from_lat -> trilinear upsample -> dec_net.0.0):
     * upsampling is linear, channel-wise and its weights sum to one, so conv(upsample(from_lat(z))) is a convolution of upsample(z) — Cv <= 16
     * input channels instead of 64, a quarter of the matrix work — plus a bias term.  With conv_w3 and conv0_lat_w3 both given the decoder
     * takes that route for block 0 (same operator up to fp32 rounding); NULL = the three separate passes. */
    const void* conv0_lat_w3;          /* avd_conv3_weight_[f16x2_]f32 image of the composite weight [out][kt][kh][kw][in], in < Cv:
                                        * sum_c conv_w[0][out][c][tap] * from_lat_w[c][in], in >= Cv zero */
    const float* conv0_lat_btab;       /* [64 border classes][base]: sum over the taps INSIDE the volume of sum_c conv_w[0][out][c][tap] * from_lat_b[c];
                                        * class bits: t-1 inside, t+1 inside, h-1, h+1, w-1, w+1 (the conv zero-pads u = from_lat(.), not its bias) */
    float conv0_lat_w_scale;           /* conv_terms 3: scale of the conv0_lat_w3 image */
    /* ABI 7 (round 5): 1 = conv0_lat_w3 is the PACKED image (Cv <= 8): "tap" s of the [out][27][base] tensor handed to avd_conv3_weight_*
     * holds the composite weights of tap 2 s in channels 0 .. 7 and of tap 2 s + 1 in channels 8 .. 15 (s = 0 .. 13; tap 27 = zeros) — the
     * kernel then runs 14 k-steps of two taps instead of 27 of one whose upper 8 channels are zero.  0 = one tap per step. */
    int conv0_lat_packed;
} avd_vae_decode_desc;
/* weight image of one 3x3x3 64->64 convolution for the split-operand decoders: w_tap_major is [out][kt][kh][kw][in] fp32 */
int64_t avd_conv3_weight_bytes(void);
int avd_conv3_weight_f32(const float* w_tap_major, void* img, avd_stream_t stream);
int avd_conv3_weight_f16x2_f32(const float* w_tap_major, void* img, float scale, avd_stream_t stream);
int64_t avd_vae_decode_workspace_bytes(const avd_vae_decode_desc* d);
int avd_vae_decode_f32(const avd_vae_decode_desc* d, const float* z, float* out, void* workspace,
                       int64_t workspace_bytes, avd_stream_t stream);

/* ---- next-1: VideoVAE.encode — avdiff/models/encoders/vae_video3d.py:164-189 (deterministic path):
 * x [B,in_ch,T,H,W] NCDHW (already cropped to multiples of t_down / s_down) -> n_blocks x [conv3x3x3 + GELU + GroupNorm]
 * -> AvgPool3d(t_down,s_down,s_down) -> to_lat (1x1x1) -> z [B,lat_ch,T/t_down,H/s_down,W/s_down] NCDHW.
 * The first conv (in_ch -> 64) runs on the same MFMA kernel with the input padded to 4 channels (K = 32 taps x 4). */
typedef struct {
    int B, in_ch, T, H, W;
    int t_down, s_down;
    int base, n_blocks, lat_ch;        /* enc_base (64), enc_blocks, latent channels */
    float gn_eps;
    const float* const* conv_w;        /* HOST array [n_blocks]: block 0: enc_net.0.0.weight re-laid as [64][32 taps][4]
                                          (27 real taps, channel 3 and taps 27..31 zero); blocks >= 1: [64][27][64] */
    const float* const* conv_b;
    const float* const* gn_w;
    const float* const* gn_b;
    const float* to_lat_w;             /* to_lat.weight (or to_mu.weight) [lat_ch,64] */
    const float* to_lat_b;
    const void* const* conv_w3;        /* optional HOST array [n_blocks] (entry 0 unused): avd_conv3_weight_f32 images of the 64->64
                                        * convolutions, which then run on the bf16 matrix pipe (as in avd_vae_decode_desc); NULL = fp32 */
    int conv_terms;                    /* as in the decode descriptor; entries 0 of the scale arrays are unused (block 0 is the fp32 4 -> 64 conv) */
    const float* conv_w_scale;
    const float* conv_a_scale;
    /* ABI 7 (round 5), optional: the FIRST convolution (in_ch <= 8 -> base) on the matrix pipe with two taps per k-step — the
     * avd_conv3_weight_f32 image of a [out][27][base] tensor whose "tap" s (s = 0 .. 13) holds enc_net.0.0.weight[out][:, tap 2 s] in
     * channels 0 .. in_ch-1 and tap 2 s + 1 in channels 8 .. 8+in_ch-1 (everything else zero).  With it set, conv_terms 0 / 6, two conv
     * blocks and pooling (4, 8, 8), avd_vae_encode_f32 writes no fp32 activation at all (folded route, avd_tune_set "vae_fold"); NULL = fp32 first conv. */
    const void* conv0_pk_w3;
} avd_vae_encode_desc;
int64_t avd_vae_encode_workspace_bytes(const avd_vae_encode_desc* d);
int avd_vae_encode_f32(const avd_vae_encode_desc* d, const float* x, float* z, void* workspace,
                       int64_t workspace_bytes, avd_stream_t stream);

/* ---- next-2: AudioCodec layers — avdiff/models/encoders/audio_codec.py:88-133, :158-214.
 * NCL conv1d (odd k <= 15, zero padding k/2, stride 1) with optional nearest-neighbour upsampling of the INPUT by
 * `upsample` folded into the indexing (decode's F.interpolate(mode="nearest") x hop), then act in {none, GELU, tanh}.
 * x [B,Cin,Lin], w [Cout,Cin,k], bias [Cout] or NULL, out [B,Cout,Lin*upsample]. */
int avd_conv1d_act_f32(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                       int Lin, int upsample, int k, int act, avd_stream_t stream);
/* audio_codec.py:158-182: right-pad with zeros / crop to Fa*hop samples, then avg_pool1d(kernel = stride = hop).
 * x [rows, L] -> out [rows, Fa]. */
int avd_avgpool_frames_f32(const float* x, float* out, int rows, int L, int Fa, int hop, avd_stream_t stream);

/* ---- next-3: sliding-window stitching — avdiff/models/infer/stream_infer.py:85-116 (crossfade_audio),
 * :119-143 (crossfade_video).  N windows of L positions x `inner` values placed every `hop` positions, weighted by
 * w[L] (host-built fade table), divided by the summed weights clamped at 1e-6; out has (N-1)*hop + L positions.
 * The u8 form divides by 255 on input and clips / scales / truncates to uint8 on output like the reference. */
int avd_crossfade_f32(const float* chunks, const float* w, float* out, int N, int L, int hop, int64_t inner,
                      avd_stream_t stream);
int avd_crossfade_u8(const uint8_t* chunks, const float* w, uint8_t* out, int N, int L, int hop, int64_t inner,
                     avd_stream_t stream);

/* device-side sampling-schedule cursor so a captured step can be replayed without host writes:
 * t_now[b] = sched[*cursor], t_prev[b] = sched[*cursor+1] for all b, then (*cursor)++ . */
int avd_sched_advance(const int64_t* sched, int n_sched, int32_t* cursor, int64_t* t_now, int64_t* t_prev,
                      int B, avd_stream_t stream);

/* ---- measurement hooks (bench.py): when enabled, every kernel launch made by this library is bracketed by
 * hipEvents recorded on the launch stream and tagged with its kernel name (template arguments included, so the
 * tags line up with rocprofv3's per-kernel rows) and its algorithmic work (FLOPs for the MFMA kernels, bytes
 * for the HBM-bound ones).  Disabled by default; zero cost when off.  Not for use during graph capture. */
int         avd_prof_enable(int on);       /* on=1 start recording (clears previous records), on=0 stop */
int         avd_prof_num_tags(void);
const char* avd_prof_tag_name(int tag);
/* synchronises the recorded events and accumulates per tag: launches, total milliseconds, algorithmic work */
int         avd_prof_report(int64_t* launches, double* total_ms, double* work, int ntags);

#ifdef __cplusplus
}
#endif
#endif /* AVDIFF_HIP_H */
