// Library plumbing (errors, version) and the stateless composites: MMDiT forward, noise-head forward, the fused
// front end and one whole CFG denoising step.  A composite only enqueues the kernels of the other files, in
// order, on the caller's stream — one FFI call per module / per step, hipGraph-capturable.
#include "avd_common.h"

#include <stdlib.h>

#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace avd {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int LdsAttr::ensure(const void* kern, int lds_bytes, const char* what) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(AVD_ELAUNCH, "%s: hipGetDevice: %s", what, hipGetErrorString(e));
    const uint64_t bit = dev < 64 ? (uint64_t)1 << dev : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return AVD_OK;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return set_error(AVD_ELAUNCH, "%s: hipFuncSetAttribute: %s", what, hipGetErrorString(e));
    done.fetch_or(bit, std::memory_order_release);
    return AVD_OK;
}

// ---------------------------------------------------------------- measurement hooks (single host thread; see the header)
bool g_prof_on = false;
namespace {
struct ProfRec { hipEvent_t a, b; int tag; double work; };
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_pool;
std::vector<std::string> g_tags;
hipEvent_t take_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace
int prof_tag_id(const char* fmt, ...) {
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    for (size_t i = 0; i < g_tags.size(); ++i)
        if (g_tags[i] == buf) return (int)i;
    g_tags.emplace_back(buf);
    return (int)g_tags.size() - 1;
}
void prof_mark(int tag, double work, hipStream_t st, bool begin) {
    if (begin) {
        ProfRec r{take_event(), take_event(), tag, work};
        (void)hipEventRecord(r.a, st);
        g_recs.push_back(r);
    } else if (!g_recs.empty()) {
        (void)hipEventRecord(g_recs.back().b, st);
    }
}

int temb_f32(const int64_t* t, const float* freqs, float* out, int B, int dim, float max_period, hipStream_t st);
int tube_patch_f32(const float* z, float* tok, int B, int C, int T, int H, int W, int t, int h, int w, hipStream_t st);
int audio_tokens_f32(const float* z, float* tok, int B, int Ca, int F, int len, int stride, hipStream_t st);
int assemble_f32(float* X2, const float* temb, const float* Xp, int B, int N, int d, int tdim, int Nt, int Np,
                 int target_first, hipStream_t st);
int assemble_rows_f32(float* X2, const int64_t* t_now, const float* freqs, const float* Xp, float* ss, int B, int N, int d, int tdim,
                      int Nt, int Np, int target_first, float max_period, hipStream_t st);
int cfg_unpatch_ddim_f32(const float* eps2, const float* z, const int64_t* t_now, const int64_t* t_prev,
                         const float* abar, int T_train, float guidance, float eta, const float* noise, float* z_out,
                         int B, int C, int T, int H, int W, int t, int h, int w, hipStream_t st);
int cfg_untoken_ddim_audio_f32(const float* eps2, const float* z, const int64_t* t_now, const int64_t* t_prev,
                               const float* abar, int T_train, float guidance, float eta, const float* noise,
                               float* z_out, int B, int Ca, int F, int len, int stride, hipStream_t st);

static inline int64_t align_up(int64_t x) { return (x + 255) & ~(int64_t)255; }

struct Carver {
    char* base;
    int64_t used, cap;
    bool ok = true;       // false once a take() went past the capacity: the caller checks before it launches anything
    float* take(int64_t floats) {
        float* p = reinterpret_cast<float*>(base + used);
        used += align_up(floats * (int64_t)sizeof(float));
        if (used > cap) ok = false;
        return p;
    }
};

// ---------------------------------------------------------------- MMDiT.forward
// bf16x3 matmul path (gemm_bf16x3.hip, attn_bf16x3.hip): taken when every block carries split3 weight images and the batch has
// enough rows for the 256-row tiles
constexpr int64_t kSplitMinRows = 6144;    // 128x128 geometry at B=32 (8,512 rows) gains 1.34x; 64x64 (3,904 rows) does not fill the 256-row tiles
// the six-term mode on the 16x16x32 kernels has 64 .. 224-row blocks for its residual launches (round 4): it passes the fp32 MFMA path
// between 1,684 rows (606 against 680 steps/s) and 2,128 rows (631 against 516); 3,904 rows (C2): 549 against 462
constexpr int64_t kSplitMinRowsM16 = 2048;
static int64_t g_s3_min_rows = [] { const char* e = getenv("AVD_S3_MIN_ROWS"); return e ? (int64_t)atoll(e) : (int64_t)-1; }();
static bool g_no_fold = getenv("AVD_NO_FOLD") != nullptr;
// avd_tune_set "s3_min_rows" (measurement aid): -1 = the per-mode defaults above, >= 0 = that many rows in every mode
static int64_t split_min_rows(int terms = 3) {
    if (g_s3_min_rows >= 0) return g_s3_min_rows;
    return (terms == 0 || terms == 6) && g_s3_m16 ? kSplitMinRowsM16 : kSplitMinRows;
}
// what the weights and shapes allow, whatever the process-wide tunables say (workspace sizing: core_ws_bytes)
static bool core_split_capable(const avd_core_weights* w, int64_t M) {
    if (w->norm_kind != 0) return false;
    if (!gemm_bf16x3_supported(M, 3 * w->d, w->d) || !gemm_bf16x3_supported(M, w->d, w->d) ||
        !gemm_bf16x3_supported(M, w->mlp_hidden, w->d) || !gemm_bf16x3_supported(M, w->d, w->mlp_hidden))
        return false;
    for (int l = 0; l < w->n_layers; ++l) {
        const avd_block_weights& b = w->blocks[l];
        if (!b.in_proj_weight3 || !b.out_proj_weight3 || !b.fc1_weight3 || !b.fc2_weight3) return false;
    }
    return true;
}
// Rows of the WHOLE step when avd_denoise_step_f32 runs its cond / null halves as two calls on two streams (0 otherwise): the kernel
// family is chosen once per step from the stacked 2B x N row count and both halves take it, so the one-stream and the two-stream layouts
// stay bit-identical in the band where one half alone would fall under the row threshold (2,048 <= 2 B N < 4,096 rows).
static thread_local int64_t t_step_rows = 0;
static bool core_use_split(const avd_core_weights* w, int64_t M) {
    // (the reduced-precision one-term mode is an explicit request, not a speed heuristic: it takes the split kernels at any size)
    const int64_t Mt = t_step_rows > M ? t_step_rows : M;
    if (Mt < split_min_rows(w->split_terms) && w->split_terms != 1 && w->attn_mode != 1) return false;
    return core_split_capable(w, M);
}

// split path with RMSNorm folded into its neighbours (bf16 planes only: the un-normalised residual stream has no bound an fp16 image
// could be scaled by): needs the scale-carrying weight images and 64-column chunks for the sums of squares
static bool core_split_fold_capable(const avd_core_weights* w) {
    if (w->norm_kind != 0 || w->split_terms == 3 || w->d % 64 != 0) return false;
    for (int l = 0; l < w->n_layers; ++l)
        if (!w->blocks[l].in_proj_weight3n || !w->blocks[l].fc1_weight3n) return false;
    return true;
}
static bool core_use_split_fold(const avd_core_weights* w) { return !g_no_fold && core_split_fold_capable(w); }

// f16x2 path with the norms that follow a residual add finished inside that GEMM's epilogue (blocks that own whole rows)
static bool core_split_rownorm_capable(const avd_core_weights* w) {
    return w->norm_kind == 0 && w->split_terms == 3 && gemm_bf16x3_rownorm_supported(w->d, 3);
}
static bool core_use_split_rownorm(const avd_core_weights* w) { return !g_no_fold && core_split_rownorm_capable(w); }

// fp32 path with RMSNorm folded into the neighbouring GEMM epilogues: needs the scale-carrying weights and LDS-DMA-able shapes
static bool core_use_fold(const avd_core_weights* w) {
    if (g_no_fold || w->norm_kind != 0) return false;              // avd_tune_set "no_fold": measurement aid
    if (!gemm_f32_fold_supported(3 * w->d, w->d) || !gemm_f32_fold_supported(w->d, w->d) ||
        !gemm_f32_fold_supported(w->mlp_hidden, w->d) || !gemm_f32_fold_supported(w->d, w->mlp_hidden))
        return false;
    for (int l = 0; l < w->n_layers; ++l)
        if (!w->blocks[l].in_proj_weight_n || !w->blocks[l].fc1_weight_n) return false;
    return true;
}

// last-block dead-row elimination (avd_tune_set "core_trim" 0 switches it off: measurement aid / A-B in tests)
static bool g_core_trim = getenv("AVD_CORE_TRIM") ? atoi(getenv("AVD_CORE_TRIM")) != 0 : true;
static int64_t core_trim_floats(const avd_core_weights* w, int B, int N) { return (int64_t)B * N * w->d; }

// the wide scratch of the bf16x3 path holds the qkv3 image, then the fc1 image
static int64_t core_split_wide_bytes(const avd_core_weights* w, int B, int N) {
    const int64_t qkv3 = qkv3_bytes(B, N, w->n_heads), fc1 = split3_bytes((int64_t)B * N, w->mlp_hidden);
    return qkv3 > fc1 ? qkv3 : fc1;
}

// The requirement is the maximum over every variant the process-wide tunables (avd_tune_set: no_fold, s3_splitk, s3_m16, s3_min_rows)
// can select for these weights, so a workspace sized once (DenoiseEngine._bind_weights) stays large enough when a tunable is flipped
// afterwards — tests and tools do that between steps, also inside a graph capture.
static int64_t core_ws_bytes(const avd_core_weights* w, int B, int N) {
    const int64_t M = (int64_t)B * N;
    const int wide = 3 * w->d > w->mlp_hidden ? 3 * w->d : w->mlp_hidden;
    int64_t fp32_path = align_up(M * w->d * 4) + align_up(M * wide * 4) + 2 * align_up(M * (w->d / 32 + 1) * 4) +
                        align_up(gemm_f32_splitk_ws_max_floats(M, w->d, w->mlp_hidden) * 4);       // split-K partial sums of fc2 (tiny batches)
    if (!core_split_capable(w, M)) return fp32_path;
    // + split3 image of the norm / attention output, and the wide buffer must also hold the split3 image of the MLP hidden
    const int64_t wide_b = core_split_wide_bytes(w, B, N);
    // third region: (attn_mode 1) the fp8 attention's operand images
    const int64_t f8_b = w->attn_mode == 1 ? attn_fp8_ws_bytes(B, N, w->n_heads) : 0;
    // folded norms: a second [M][d] image (the residual stream's) and the table of its rows' sums of squares
    const int64_t fold_b = core_split_fold_capable(w) ? align_up(split3_bytes(M, w->d)) + align_up(M * (w->d / 64) * 4)
                         : core_split_rownorm_capable(w) ? align_up(split3_bytes(M, w->d)) : 0;        // the normalised stream's image
    // split-K partial sums of fc2 for batches that cannot fill the chip (folded bf16-plane path only), at the largest slice count
    const int64_t sk_b = core_split_fold_capable(w) ? align_up(gemm_bf16x3_splitk_ws_max_floats(M, w->d, w->mlp_hidden) * 4) : 0;
    const int64_t trim_b = core_split_fold_capable(w) ? align_up(core_trim_floats(w, B, N) * 4) : 0;      // compact stream of the last block
    const int64_t split_path = align_up(wide_b) + align_up(split3_bytes(M, w->d)) + align_up(f8_b) + fold_b + sk_b + trim_b;
    return split_path > fp32_path ? split_path : fp32_path;
}

static int check_core(const avd_core_weights* w) {
    AVD_REQUIRE(w && w->blocks && w->final_norm_scale, AVD_EINVAL, "core: null weight table");
    AVD_REQUIRE(w->d > 0 && w->n_layers > 0 && w->n_heads > 0 && w->mlp_hidden > 0, AVD_EINVAL, "core: bad dims");
    AVD_REQUIRE(w->d % w->n_heads == 0, AVD_EINVAL, "core: d_model %d not divisible by n_heads %d", w->d, w->n_heads);
    AVD_REQUIRE(w->attn_mode == 0 || w->attn_mode == 1, AVD_EINVAL, "core: attn_mode must be 0 (follow the matmul mode) or 1 (fp8)");
    AVD_REQUIRE(w->norm_kind == 0 || w->norm_kind == 1, AVD_EINVAL, "core: norm_kind must be 0 (RMSNorm) or 1 (LayerNorm)");
    AVD_REQUIRE(w->d / w->n_heads == 64, AVD_EUNSUPPORTED, "core: head_dim %d unsupported (64 only)", w->d / w->n_heads);
    AVD_REQUIRE(w->d % 4 == 0 && w->mlp_hidden % 4 == 0, AVD_EUNSUPPORTED, "core: widths must be multiples of 4");
    AVD_REQUIRE(w->split_terms == 0 || w->split_terms == 6 || w->split_terms == 9 || w->split_terms == 1 || w->split_terms == 3, AVD_EINVAL,
                "core: split_terms must be 0/6 (default), 9 (strict), 1 (plain bf16) or 3 (f16x2), got %d", w->split_terms);
    if (w->split_terms == 3) {
        for (int l = 0; l < w->n_layers; ++l)
            for (int i = 0; i < 8; ++i) {
                const float s = w->blocks[l].f16x2_scale[i];
                AVD_REQUIRE(s > 0.f && s < __builtin_inff(), AVD_EINVAL, "core: blocks[%d].f16x2_scale[%d] must be positive and finite", l, i);
            }
    }
    return AVD_OK;
}

// norm of the generic path: RMSNorm (mmdt.py:39-42) or, for norm="layernorm", nn.LayerNorm (mmdt.py:44-45)
static int core_norm(const avd_core_weights* w, const float* x, const float* scale, const float* bias, float* y, int64_t M, hipStream_t st) {
    const RowMap rd{w->d, 0, 0};
    if (w->norm_kind == 0) return rmsnorm_f32(x, rd, scale, y, rd, M, w->d, w->norm_eps, st);
    AVD_REQUIRE(bias, AVD_EINVAL, "core: norm_kind 1 (LayerNorm) needs the bias vectors");
    return layernorm_act_f32(x, scale, bias, y, M, w->d, w->norm_eps, AVD_ACT_NONE, st);
}

static int core_forward(const avd_core_weights* w, const float* x, float* y, int B, int N, int out_row0,
                        int n_out_rows, const unsigned char* kpm, void* ws, int64_t ws_bytes, hipStream_t st,
                        const float* ss_first = nullptr) {
    if (int rc = check_core(w)) return rc;
    AVD_REQUIRE(x && y && B > 0 && N > 0, AVD_EINVAL, "core: bad input");
    AVD_REQUIRE(out_row0 >= 0 && n_out_rows > 0 && out_row0 + n_out_rows <= N, AVD_EINVAL, "core: bad output row window");
    const int64_t M = (int64_t)B * N;
    AVD_REQUIRE(ws && ws_bytes >= core_ws_bytes(w, B, N), AVD_EWORKSPACE, "core: workspace %lld < %lld bytes",
                (long long)ws_bytes, (long long)core_ws_bytes(w, B, N));
    const int d = w->d, hid = w->mlp_hidden, H = w->n_heads;
    Carver cv{static_cast<char*>(ws), 0, ws_bytes};
    float* hbuf = cv.take(M * d);                                   // norm output, then attention output
    float* wide = cv.take(M * (3 * d > hid ? 3 * d : hid));         // packed qkv, then MLP hidden
    const RowMap rd{d, 0, 0}, r3{3 * d, 0, 0}, rh{hid, 0, 0};
    const float scale = 1.0f / sqrtf((float)(d / H));
    const float* cur = x;                                           // residual stream lives in y after the first write
    // a key-padding mask or norm="layernorm" keeps the whole forward on the fp32 kernels (the split-operand attention has no mask
    // input, the split producers are RMSNorm's): same results at fp32-MFMA speed, documented in the header.  The fp8 attention is an
    // explicit reduced-precision request and is refused rather than silently replaced.
    AVD_REQUIRE(!(w->attn_mode == 1 && (kpm || w->norm_kind != 0)), AVD_EUNSUPPORTED,
                "core: attn_mode 1 (fp8 attention) cannot be combined with a key_padding_mask or norm_kind 1 (LayerNorm)");
    AVD_REQUIRE(!(w->attn_mode == 1 && !core_use_split(w, M)), AVD_EUNSUPPORTED,
                "core: attn_mode 1 (fp8 attention) reads the q|k|v image of the split-operand projections, which these weights / shapes "
                "do not take (every block needs its *_weight3 images; 3 d, d and mlp_hidden must be multiples of 256, d of 16)");
    if (!kpm && core_use_split(w, M)) {
        // same op sequence with the four projections on the bf16 matrix pipe (gemm_bf16x3.hip); hs / wide3 are split3 images
        const int terms = w->split_terms;
        Carver cs{static_cast<char*>(ws), 0, ws_bytes};
        const int64_t wide_b = core_split_wide_bytes(w, B, N);
        float* qkv = cs.take((wide_b + 3) / 4);
        void* wide3 = qkv;
        void* hs = cs.take((split3_bytes(M, d) + 3) / 4);
        const int64_t f8_b = w->attn_mode == 1 ? attn_fp8_ws_bytes(B, N, H) : 0;
        float* f8w = cs.take((f8_b + 3) / 4);         // fp8 attention operands (attn_mode 1 only)
        if (core_use_split_fold(w)) {
            // RMSNorm folded into its neighbours: the residual epilogues (out_proj, fc2) write the new stream as fp32, as an operand
            // image and as per-row sums of squares; in_proj / fc1 run on that un-normalised image with weights that carry the norm's
            // scale and multiply their rows by 1 / (rms + eps) before the bias.  No norm kernel between the first split and the final norm.
            void* hx = cs.take((split3_bytes(M, d) + 3) / 4);           // image of the residual stream
            float* ss = cs.take(M * (d / 64));                           // its rows' sums of squares, [M][d / 64]
            const int ns = gemm_bf16x3_splitk_slices(M, d, hid, terms);
            float* part = ns ? cs.take(gemm_bf16x3_splitk_ws_floats(M, d, ns)) : nullptr;
            // dead-row elimination in the LAST block (sample_clip.py:375,379 consume the target rows only): its attention already
            // computes only the caller's row window; when that window starts at row 0 the attention writes those rows compactly and
            // out_proj / fc1 / fc2 / the final norm run on B * n_out_rows rows instead of B * N (C3: 384 of 421)
            const bool trim = g_core_trim && out_row0 == 0 && n_out_rows < N && w->attn_mode == 0 && gemm_bf16x3_resmap_supported(terms);
            float* yc = cs.take(core_trim_floats(w, B, N));              // compact fp32 stream of the last block (sized whether or not it is used)
            // fc1 -> GELU -> fc2 as one launch (mlp_bf16x3.hip; off by default: it measures slower)
            const bool fused = g_mlp_fused && !ns && mlp_bf16x3_supported(d, hid, terms) && gemm_bf16x3_resmap_supported(terms);
            AVD_REQUIRE(cs.ok, AVD_EWORKSPACE, "core: workspace carve %lld > %lld bytes", (long long)cs.used, (long long)cs.cap);
            if (int rc = split3_rows_f32(cur, rd, hx, M, d, st, 0.f, ss)) return rc;
            for (int l = 0; l < w->n_layers; ++l) {
                const avd_block_weights& b = w->blocks[l];
                const bool last = l == w->n_layers - 1;
                const int nq = (last && out_row0 == 0) ? n_out_rows : N;
                if (last && trim) {
                    const int64_t Mc = (int64_t)B * nq;
                    if (int rc = gemm_bf16x3_qkv3(hx, b.in_proj_weight3n, b.in_proj_bias, qkv, M, N, H, d, scale * 1.4426950408889634f, terms, st,
                                                  1.f, 1.f, ss, w->norm_eps)) return rc;
                    if (int rc = attn_bf16x3(qkv, nullptr, hs, B, N, H, nq, terms, st, 1.f, 1.f, nq)) return rc;       // rows b * nq + q
                    if (int rc = gemm_bf16x3(hs, b.out_proj_weight3, b.out_proj_bias, cur, yc, hx, Mc, d, d, AVD_ACT_NONE, terms, st, 1.f, 1.f,
                                             nullptr, 0.f, ss, nullptr, nq, N)) return rc;                                  // residual rows b * N + q
                    if (fused) {
                        if (int rc = mlp_bf16x3(hx, b.fc1_weight3n, b.fc1_bias, b.fc2_weight3, b.fc2_bias, ss, w->norm_eps, yc, yc, nullptr, nullptr, Mc,
                                                d, hid, st)) return rc;
                    } else {
                        if (int rc = gemm_bf16x3(hx, b.fc1_weight3n, b.fc1_bias, nullptr, nullptr, wide3, Mc, hid, d, AVD_ACT_GELU, terms, st, 1.f, 1.f,
                                                 ss, w->norm_eps)) return rc;
                        if (ns) {       // the same K slices as every other block: the window's rows keep their summation order
                            if (int rc = gemm_bf16x3_splitk(wide3, b.fc2_weight3, b.fc2_bias, yc, yc, nullptr, nullptr, Mc, d, hid, terms, ns, part,
                                                            st)) return rc;
                        } else {
                            if (int rc = gemm_bf16x3(wide3, b.fc2_weight3, b.fc2_bias, yc, yc, nullptr, Mc, d, hid, AVD_ACT_NONE, terms, st)) return rc;
                        }
                    }
                    // final norm from the compact rows into the caller's [B, N, d] layout (rows outside the window stay as they were)
                    return rmsnorm_f32(yc, rd, w->final_norm_scale, y, RowMap{d, nq, (int64_t)N * d}, Mc, d, w->norm_eps, st);
                }
                if (int rc = gemm_bf16x3_qkv3(hx, b.in_proj_weight3n, b.in_proj_bias, qkv, M, N, H, d, scale * 1.4426950408889634f, terms, st,
                                              1.f, 1.f, ss, w->norm_eps)) return rc;
                if (w->attn_mode == 1) {
                    if (int rc = attn_fp8(qkv, f8w, f8_b, nullptr, hs, B, N, H, nq, st, terms, 1.f, 1.f)) return rc;
                } else {
                    if (int rc = attn_bf16x3(qkv, nullptr, hs, B, N, H, nq, terms, st, 1.f, 1.f)) return rc;
                }
                if (int rc = gemm_bf16x3(hs, b.out_proj_weight3, b.out_proj_bias, cur, y, hx, M, d, d, AVD_ACT_NONE, terms, st, 1.f, 1.f, nullptr,
                                         0.f, ss)) return rc;
                cur = y;
                if (fused) {
                    if (int rc = mlp_bf16x3(hx, b.fc1_weight3n, b.fc1_bias, b.fc2_weight3, b.fc2_bias, ss, w->norm_eps, y, y, last ? nullptr : hx,
                                            last ? nullptr : ss, M, d, hid, st)) return rc;
                    continue;
                }
                if (int rc = gemm_bf16x3(hx, b.fc1_weight3n, b.fc1_bias, nullptr, nullptr, wide3, M, hid, d, AVD_ACT_GELU, terms, st, 1.f, 1.f, ss,
                                         w->norm_eps)) return rc;
                if (ns) {       // too few blocks for the chip: K slices + a deterministic reduction that writes what the epilogue would
                    if (int rc = gemm_bf16x3_splitk(wide3, b.fc2_weight3, b.fc2_bias, y, y, last ? nullptr : hx, last ? nullptr : ss, M, d, hid,
                                                    terms, ns, part, st)) return rc;
                } else if (last) {     // nothing reads the stream's image after the last block: the final norm takes the fp32 rows
                    if (int rc = gemm_bf16x3(wide3, b.fc2_weight3, b.fc2_bias, y, y, nullptr, M, d, hid, AVD_ACT_NONE, terms, st)) return rc;
                } else {
                    if (int rc = gemm_bf16x3(wide3, b.fc2_weight3, b.fc2_bias, y, y, hx, M, d, hid, AVD_ACT_NONE, terms, st, 1.f, 1.f, nullptr, 0.f,
                                             ss)) return rc;
                }
            }
            return rmsnorm_f32(y, rd, w->final_norm_scale, y, rd, M, d, w->norm_eps, st);
        }
        // f16x2 with d = 512: out_proj / fc2 own whole rows and write the NEXT norm's output image themselves (EPI_RES_NORM); the only
        // norm kernel left before the final norm is the first block's norm1
        const bool rown = core_use_split_rownorm(w);
        void* hn = rown ? cs.take((split3_bytes(M, d) + 3) / 4) : hs;      // image of the normalised stream (hs: attention output)
        AVD_REQUIRE(cs.ok, AVD_EWORKSPACE, "core: workspace carve %lld > %lld bytes", (long long)cs.used, (long long)cs.cap);
        for (int l = 0; l < w->n_layers; ++l) {
            const avd_block_weights& b = w->blocks[l];
            const bool last = l == w->n_layers - 1;
            const int nq = (last && out_row0 == 0) ? n_out_rows : N;
            // f16x2 (terms 3): the images carry the block's power-of-two scales; every other mode ignores them
            const bool h2 = terms == 3;
            const float* fs = b.f16x2_scale;
            const float s_n1 = h2 ? fs[4] : 0.f, s_qkv = h2 ? fs[5] : 1.f, s_n2 = h2 ? fs[6] : 0.f, s_fc1 = h2 ? fs[7] : 1.f;
            const float w_in = h2 ? fs[0] : 1.f, w_out = h2 ? fs[1] : 1.f, w_fc1 = h2 ? fs[2] : 1.f, w_fc2 = h2 ? fs[3] : 1.f;
            if (!rown || l == 0)
                if (int rc = rmsnorm_split3_f32(cur, b.norm1_scale, hn, M, d, w->norm_eps, st, s_n1)) return rc;
            if (int rc = gemm_bf16x3_qkv3(hn, b.in_proj_weight3, b.in_proj_bias, qkv, M, N, H, d, scale * 1.4426950408889634f, terms, st,
                                          h2 ? s_n1 * w_in : 1.f, s_qkv)) return rc;
            if (w->attn_mode == 1) {
                if (int rc = attn_fp8(qkv, f8w, f8_b, nullptr, hs, B, N, H, nq, st, terms, s_qkv, s_qkv)) return rc;
            } else {
                if (int rc = attn_bf16x3(qkv, nullptr, hs, B, N, H, nq, terms, st, s_qkv, s_qkv)) return rc;
            }
            if (rown) {
                // new stream (fp32) + image of norm2 of it, in one epilogue
                if (int rc = gemm_bf16x3(hs, b.out_proj_weight3, b.out_proj_bias, cur, y, hn, M, d, d, AVD_ACT_NONE, terms, st, s_qkv * w_out, s_n2,
                                         nullptr, w->norm_eps, nullptr, b.norm2_scale)) return rc;
                cur = y;
            } else {
                if (int rc = gemm_bf16x3(hs, b.out_proj_weight3, b.out_proj_bias, cur, y, nullptr, M, d, d, AVD_ACT_NONE, terms, st,
                                         h2 ? s_qkv * w_out : 1.f, 1.f)) return rc;
                cur = y;
                if (int rc = rmsnorm_split3_f32(y, b.norm2_scale, hn, M, d, w->norm_eps, st, s_n2)) return rc;
            }
            if (int rc = gemm_bf16x3(hn, b.fc1_weight3, b.fc1_bias, nullptr, nullptr, wide3, M, hid, d, AVD_ACT_GELU, terms, st,
                                     h2 ? s_n2 * w_fc1 : 1.f, s_fc1)) return rc;
            if (rown && !last) {
                // ... and the next block's norm1 image from fc2's epilogue
                const avd_block_weights& nb = w->blocks[l + 1];
                if (int rc = gemm_bf16x3(wide3, b.fc2_weight3, b.fc2_bias, y, y, hn, M, d, hid, AVD_ACT_NONE, terms, st, s_fc1 * w_fc2,
                                         nb.f16x2_scale[4], nullptr, w->norm_eps, nullptr, nb.norm1_scale)) return rc;
            } else {
                if (int rc = gemm_bf16x3(wide3, b.fc2_weight3, b.fc2_bias, y, y, nullptr, M, d, hid, AVD_ACT_NONE, terms, st,
                                         h2 ? s_fc1 * w_fc2 : 1.f, 1.f)) return rc;
            }
        }
        return rmsnorm_f32(y, rd, w->final_norm_scale, y, rd, M, d, w->norm_eps, st);
    }
    if (core_use_fold(w)) {
        // RMSNorm folded into its neighbours (GemmArgs in gemm_f32.hip): the residual epilogues emit per-row sums of squares, the
        // in_proj / fc1 GEMMs run on the un-normalised stream with scale-carrying weights and multiply their rows by 1/rms
        float* ssA = cv.take(M * (d / 32 + 1));     // sums of squares of the stream entering norm1
        float* ssB = cv.take(M * (d / 32 + 1));     // ... entering norm2
        const int nsf = gemm_f32_splitk_slices(M, d, hid);         // K slices of fc2 when its blocks cover less than half of the CUs
        float* partf = nsf ? cv.take((int64_t)nsf * M * d) : nullptr;
        AVD_REQUIRE(cv.ok, AVD_EWORKSPACE, "core: workspace carve %lld > %lld bytes", (long long)cv.used, (long long)cv.cap);
        const float sqrt_d = (float)sqrt((double)d);
        const float* ssA_in = ss_first;                // the front end may already have the rows' sums of squares
        if (!ssA_in) {
            if (int rc = rowss_f32(cur, ssA, M, d, st)) return rc;
            ssA_in = ssA;
        }
        int colsA = 1;
        for (int l = 0; l < w->n_layers; ++l) {
            const avd_block_weights& b = w->blocks[l];
            const bool last = l == w->n_layers - 1;
            const int nq = (last && out_row0 == 0) ? n_out_rows : N;
            if (int rc = gemm_f32_fold(cur, rd, b.in_proj_weight_n, b.in_proj_bias, nullptr, rd, wide, r3, M, 3 * d, d, AVD_ACT_NONE,
                                       l == 0 ? ssA_in : ssA, colsA, sqrt_d, w->norm_eps, nullptr, st)) return rc;
            if (int rc = attn_f32(wide, hbuf, B, N, H, d / H, scale, nq, kpm, st)) return rc;
            if (int rc = gemm_f32_fold(hbuf, rd, b.out_proj_weight, b.out_proj_bias, cur, rd, y, rd, M, d, d, AVD_ACT_NONE, nullptr, 0,
                                       1.f, 0.f, ssB, st)) return rc;
            cur = y;
            if (int rc = gemm_f32_fold(y, rd, b.fc1_weight_n, b.fc1_bias, nullptr, rd, wide, rh, M, hid, d, AVD_ACT_GELU, ssB, d / 32,
                                       sqrt_d, w->norm_eps, nullptr, st)) return rc;
            if (nsf) {
                if (int rc = gemm_f32_splitk(wide, rh, b.fc2_weight, b.fc2_bias, y, rd, y, rd, M, d, hid, nsf, partf, ssA, st)) return rc;
            } else {
                if (int rc = gemm_f32_fold(wide, rh, b.fc2_weight, b.fc2_bias, y, rd, y, rd, M, d, hid, AVD_ACT_NONE, nullptr, 0, 1.f, 0.f,
                                           ssA, st)) return rc;
            }
            colsA = d / 32;
        }
        return rmsnorm_f32(y, rd, w->final_norm_scale, y, rd, M, d, w->norm_eps, st);
    }
    for (int l = 0; l < w->n_layers; ++l) {
        const avd_block_weights& b = w->blocks[l];
        const bool last = l == w->n_layers - 1;
        // dead-row elimination: after the last block's K/V are formed only the rows the caller consumes matter
        const int nq = (last && out_row0 == 0) ? n_out_rows : N;
        if (int rc = core_norm(w, cur, b.norm1_scale, b.norm1_bias, hbuf, M, st)) return rc;
        if (int rc = gemm_f32(hbuf, rd, b.in_proj_weight, b.in_proj_bias, nullptr, rd, wide, r3, M, 3 * d, d, AVD_ACT_NONE, st)) return rc;
        if (int rc = attn_f32(wide, hbuf, B, N, H, d / H, scale, nq, kpm, st)) return rc;
        if (int rc = gemm_f32(hbuf, rd, b.out_proj_weight, b.out_proj_bias, cur, rd, y, rd, M, d, d, AVD_ACT_NONE, st)) return rc;
        cur = y;
        if (int rc = core_norm(w, y, b.norm2_scale, b.norm2_bias, hbuf, M, st)) return rc;
        if (int rc = gemm_f32(hbuf, rd, b.fc1_weight, b.fc1_bias, nullptr, rd, wide, rh, M, hid, d, AVD_ACT_GELU, st)) return rc;
        if (int rc = gemm_f32(wide, rh, b.fc2_weight, b.fc2_bias, y, rd, y, rd, M, d, hid, AVD_ACT_NONE, st)) return rc;
    }
    return core_norm(w, y, w->final_norm_scale, w->final_norm_bias, y, M, st);
}

// ---------------------------------------------------------------- MultiModalNoiseHead (one modality path)
// split-operand mode of the head (gemm_bf16x3.hip): all four kinds of Linear must fit the 256-column tiles
static bool head_split_capable(const avd_head_weights* w, int64_t rows) {
    if (w->split_terms == 0) return false;
    if (!w->input_proj_weight3 || !w->out_proj_weight3 || (w->n_shared > 0 && !w->shared_lin_weight3)) return false;
    for (int j = 0; j < w->n_shared; ++j)
        if (!w->shared_lin_weight3[j]) return false;
    if (w->split_terms == 3 && !w->f16x2_scale) return false;
    return gemm_bf16x3_supported(rows, w->hidden, w->d_in) && gemm_bf16x3_supported(rows, w->hidden, w->hidden) &&
           gemm_bf16x3_supported(rows, w->d_out, w->hidden);
}
static thread_local int64_t t_step_head_rows = 0;      // as t_step_rows, for the noise head's row count
static bool head_use_split(const avd_head_weights* w, int64_t rows) {
    return (t_step_head_rows > rows ? t_step_head_rows : rows) >= split_min_rows() && head_split_capable(w, rows);
}

static int64_t head_ws_bytes(const avd_head_weights* w, int64_t rows) {
    const int64_t fp32_path = 2 * align_up(rows * w->hidden * 4);
    if (!head_split_capable(w, rows)) return fp32_path;      // sized for either path: "s3_min_rows" may move between calls
    const int wide = w->d_in > w->hidden ? w->d_in : w->hidden;
    const int64_t split_path = 2 * align_up(split3_bytes(rows, wide)) + align_up(rows * w->hidden * 4);
    return split_path > fp32_path ? split_path : fp32_path;
}

static int head_forward(const avd_head_weights* w, const float* h, RowMap hm, int64_t rows, float* out, void* ws,
                        int64_t ws_bytes, hipStream_t st) {
    AVD_REQUIRE(w && h && out, AVD_EINVAL, "head: null pointer");
    AVD_REQUIRE(w->d_in > 0 && w->hidden > 0 && w->d_out > 0 && w->n_shared >= 0, AVD_EINVAL, "head: bad dims");
    AVD_REQUIRE(w->input_proj_weight && w->out_proj_weight, AVD_EINVAL, "head: null weights");
    AVD_REQUIRE(w->n_shared == 0 || (w->shared_lin_weight && w->shared_lin_bias && w->shared_ln_weight && w->shared_ln_bias),
                AVD_EINVAL, "head: null shared-trunk tables");
    AVD_REQUIRE(w->split_terms == 0 || w->split_terms == 6 || w->split_terms == 9 || w->split_terms == 1 || w->split_terms == 3, AVD_EINVAL,
                "head: split_terms must be 0 (fp32), 6, 9, 1 or 3, got %d", w->split_terms);
    AVD_REQUIRE(ws && ws_bytes >= head_ws_bytes(w, rows), AVD_EWORKSPACE, "head: workspace too small");
    Carver cv{static_cast<char*>(ws), 0, ws_bytes};
    const RowMap rh{w->hidden, 0, 0}, ro{w->d_out, 0, 0};
    if (head_use_split(w, rows)) {
        // the same chain with every Linear on the split-operand kernels; activations travel as operand images
        const int terms = w->split_terms, n = w->n_shared;
        const bool h2 = terms == 3;
        const float* fs = w->f16x2_scale;
        if (h2)
            for (int i = 0; i < 2 * (n + 2); ++i)
                AVD_REQUIRE(fs[i] > 0.f && fs[i] < __builtin_inff(), AVD_EINVAL, "head: f16x2_scale[%d] must be positive and finite", i);
        auto wsc = [&](int i) { return h2 ? fs[i] : 1.f; };                  // weight image i: input_proj, shared 0.., out_proj
        auto asc = [&](int i) { return h2 ? fs[n + 2 + i] : 0.f; };          // activation image i: input, input_proj out, LN 0..
        const int wide = w->d_in > w->hidden ? w->d_in : w->hidden;
        void* imgA = cv.take((split3_bytes(rows, wide) + 3) / 4);
        void* imgB = cv.take((split3_bytes(rows, wide) + 3) / 4);
        float* t2 = cv.take(rows * w->hidden);
        if (int rc = split3_rows_f32(h, hm, imgA, rows, w->d_in, st, asc(0))) return rc;
        if (int rc = gemm_bf16x3(imgA, w->input_proj_weight3, w->input_proj_bias, nullptr, nullptr, imgB, rows, w->hidden, w->d_in, AVD_ACT_NONE,
                                 terms, st, h2 ? asc(0) * wsc(0) : 1.f, h2 ? asc(1) : 1.f)) return rc;
        void* cur = imgB;
        void* nxt = imgA;
        for (int j = 0; j < n; ++j) {
            if (int rc = gemm_bf16x3(cur, w->shared_lin_weight3[j], w->shared_lin_bias[j], nullptr, t2, nullptr, rows, w->hidden, w->hidden,
                                     AVD_ACT_NONE, terms, st, h2 ? asc(1 + j) * wsc(1 + j) : 1.f, 1.f)) return rc;
            if (int rc = layernorm_act_split3_f32(t2, w->shared_ln_weight[j], w->shared_ln_bias[j], nxt, rows, w->hidden, w->ln_eps, w->act, st,
                                                  asc(2 + j))) return rc;
            void* t = cur; cur = nxt; nxt = t;
        }
        return gemm_bf16x3(cur, w->out_proj_weight3, w->out_proj_bias, nullptr, out, nullptr, rows, w->d_out, w->hidden, AVD_ACT_NONE, terms, st,
                           h2 ? asc(1 + n) * wsc(1 + n) : 1.f, 1.f);
    }
    float* t1 = cv.take(rows * w->hidden);
    float* t2 = cv.take(rows * w->hidden);
    if (int rc = gemm_f32(h, hm, w->input_proj_weight, w->input_proj_bias, nullptr, rh, t1, rh, rows, w->hidden, w->d_in, AVD_ACT_NONE, st)) return rc;
    for (int j = 0; j < w->n_shared; ++j) {
        if (int rc = gemm_f32(t1, rh, w->shared_lin_weight[j], w->shared_lin_bias[j], nullptr, rh, t2, rh, rows, w->hidden, w->hidden, AVD_ACT_NONE, st)) return rc;
        if (int rc = layernorm_act_f32(t2, w->shared_ln_weight[j], w->shared_ln_bias[j], t1, rows, w->hidden, w->ln_eps, w->act, st)) return rc;
    }
    return gemm_f32(t1, rh, w->out_proj_weight, w->out_proj_bias, nullptr, ro, out, ro, rows, w->d_out, w->hidden, AVD_ACT_NONE, st);
}

// ---------------------------------------------------------------- fused front end
static int check_embed(const avd_embed_desc* e) {
    AVD_REQUIRE(e, AVD_EINVAL, "embed: null descriptor");
    AVD_REQUIRE(e->B > 0 && e->d > 0 && e->tdim >= 0, AVD_EINVAL, "embed: bad widths");
    AVD_REQUIRE(e->temb_add ? e->tdim == e->d : e->tdim < e->d, AVD_EINVAL,
                "embed: timestep width %d does not fit token width %d in %s mode", e->tdim, e->d, e->temb_add ? "add" : "concat");
    AVD_REQUIRE(e->d % 4 == 0 && e->tdim % 4 == 0, AVD_EUNSUPPORTED, "embed: d and tdim must be multiples of 4");
    AVD_REQUIRE(e->Nt > 0 && e->Np >= 0, AVD_EINVAL, "embed: bad token counts");
    if (e->target_kind == 0) {
        AVD_REQUIRE(e->p0 > 0 && e->p1 > 0 && e->p2 > 0 && e->T % e->p0 == 0 && e->H % e->p1 == 0 && e->W % e->p2 == 0,
                    AVD_EINVAL, "tube sizes must divide latent dims");
        AVD_REQUIRE(e->Nt == (e->T / e->p0) * (e->H / e->p1) * (e->W / e->p2), AVD_EINVAL, "embed: token count mismatch");
    } else if (e->target_kind == 1) {
        AVD_REQUIRE(e->p0 > 0 && e->p1 > 0 && e->T >= e->p0, AVD_EINVAL, "embed: bad audio chunking");
        AVD_REQUIRE(e->Nt == (e->T - e->p0) / e->p1 + 1, AVD_EINVAL, "embed: token count mismatch");
    } else {
        return set_error(AVD_EINVAL, "embed: unknown target_kind %d", e->target_kind);
    }
    return AVD_OK;
}
static int embed_tok_dim(const avd_embed_desc* e) {
    return e->target_kind == 0 ? e->C * e->p0 * e->p1 * e->p2 : e->C * e->p0;
}
static int64_t embed_ws_floats(const avd_embed_desc* e) {
    return align_up((int64_t)e->B * e->Nt * embed_tok_dim(e) * 4) / 4 + align_up((int64_t)e->B * (e->tdim > 0 ? e->tdim : 1) * 4) / 4;
}

// ss_out (optional, concat mode): per-row sums of squares of the finished X2, [2B*N] — spares MMDiT's first folded norm its pass
static int embed_cfg_pair(const avd_embed_desc* e, const float* z, const float* Wt, const float* bt,
                          const int64_t* t_now, const float* Xp, float* tok_ws, float* X2, hipStream_t st, float* ss_out = nullptr) {
    if (int rc = check_embed(e)) return rc;
    AVD_REQUIRE(z && Wt && t_now && tok_ws && X2 && (e->Np == 0 || Xp), AVD_EINVAL, "embed: null pointer");
    const int B = e->B, d = e->d, N = e->Nt + e->Np, D = embed_tok_dim(e);
    float* tok = tok_ws;
    float* temb = tok_ws + align_up((int64_t)B * e->Nt * D * 4) / 4;
    // video target in the sampler's concat mode: the tube patch is the adapter GEMM's A-operand load (no token matrix is written)
    const bool fuse_patch = e->target_kind == 0 && !e->temb_add && e->p2 % 4 == 0 && e->W % 4 == 0 && D % 4 == 0 && aligned16(z);
    if (e->target_kind == 0) {
        if (!fuse_patch)
            if (int rc = tube_patch_f32(z, tok, B, e->C, e->T, e->H, e->W, e->p0, e->p1, e->p2, st)) return rc;
    } else {
        if (int rc = audio_tokens_f32(z, tok, B, e->C, e->T, e->p0, e->p1, st)) return rc;
    }
    if (e->tdim > 0 && e->temb_add)
        if (int rc = temb_f32(t_now, e->temb_freqs, temb, B, e->tdim, 10000.f, st)) return rc;
    // adapter GEMM straight into the cond half's target rows (segmented C: one segment per sample)
    float* c0 = X2 + (e->target_first ? 0 : (int64_t)e->Np * d);
    const RowMap cm{d, e->Nt, (int64_t)N * d};
    if (e->temb_add) {
        // trainer-style embedding (train/trainer.py:45-49): tokens + temb, both d wide.  The per-sample embedding row is
        // the GEMM's residual operand through a row map with zero row stride (one segment of Nt rows per sample).
        const RowMap rm{0, e->Nt, (int64_t)e->tdim};
        if (int rc = gemm_f32(tok, RowMap{D, 0, 0}, Wt, bt, temb, rm, c0, cm, (int64_t)B * e->Nt, d, D, AVD_ACT_NONE, st)) return rc;
        return assemble_f32(X2, temb, Xp, B, N, d, 0, e->Nt, e->Np, e->target_first, st);
    }
    if (fuse_patch) {
        const TubeGather tg{e->T, e->H, e->W, e->p0, e->p1, e->p2, e->H / e->p1, e->W / e->p2, e->Nt, (int64_t)e->C * e->T * e->H * e->W};
        if (int rc = gemm_f32_tube(z, tg, Wt, bt, c0, cm, (int64_t)B * e->Nt, d - e->tdim, D, st)) return rc;
    } else {
        if (int rc = gemm_f32(tok, RowMap{D, 0, 0}, Wt, bt, nullptr, cm, c0, cm, (int64_t)B * e->Nt, d - e->tdim, D, AVD_ACT_NONE, st)) return rc;
    }
    // timestep columns, null-half copies, prompt rows and the rows' sums of squares in one pass
    return assemble_rows_f32(X2, t_now, e->temb_freqs, Xp, ss_out, B, N, d, e->tdim, e->Nt, e->Np, e->target_first, 10000.f, st);
}

// ---------------------------------------------------------------- one CFG denoising step
#define AVD_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess) return set_error(AVD_ELAUNCH, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// second stream + fork/join events of the split-stream step, one set per (device, caller stream): created on first use (the one
// place the library allocates, outside any capture), never shared between two caller streams
struct AuxState { hipStream_t aux = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
static std::mutex g_aux_mu;
static std::map<std::pair<int, hipStream_t>, AuxState> g_aux_map;
static int ensure_aux(hipStream_t st, AuxState& out) {
    int dev = 0;
    AVD_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_aux_mu);
    AuxState& a = g_aux_map[{dev, st}];
    if (!a.aux) {
        AVD_HIP(hipStreamCreateWithFlags(&a.aux, hipStreamNonBlocking));
        AVD_HIP(hipEventCreateWithFlags(&a.fork, hipEventDisableTiming));
        AVD_HIP(hipEventCreateWithFlags(&a.join, hipEventDisableTiming));
    }
    out = a;
    return AVD_OK;
}

struct StepPlan {
    int64_t x2, tok, core, head, eps, ss, total;
    int N, D;
    int64_t rows;     // 2B*Nt
};
static int plan_step(const avd_step_desc* s, StepPlan& p) {
    AVD_REQUIRE(s && s->core && s->head, AVD_EINVAL, "step: null descriptor");
    if (int rc = check_embed(&s->embed)) return rc;
    if (int rc = check_core(s->core)) return rc;
    const avd_embed_desc& e = s->embed;
    AVD_REQUIRE(e.d == s->core->d && s->head->d_in == e.d, AVD_EINVAL, "step: width mismatch between embed/core/head");
    p.N = e.Nt + e.Np;
    p.D = embed_tok_dim(&e);
    AVD_REQUIRE(s->head->d_out == p.D, AVD_EINVAL, "step: head d_out %d != token dim %d", s->head->d_out, p.D);
    p.rows = (int64_t)2 * e.B * e.Nt;
    p.x2 = align_up((int64_t)2 * e.B * p.N * e.d * 4);
    p.tok = align_up(embed_ws_floats(&e) * 4);
    // one slice per CFG half (they may run on two streams), or the stacked 2B batch in one piece, whichever is larger
    const int64_t halves = 2 * core_ws_bytes(s->core, e.B, p.N), whole = core_ws_bytes(s->core, 2 * e.B, p.N);
    p.core = halves > whole ? halves : whole;
    const int64_t head_halves = 2 * head_ws_bytes(s->head, p.rows / 2), head_whole = head_ws_bytes(s->head, p.rows);
    p.head = head_halves > head_whole ? head_halves : head_whole;
    p.eps = align_up(p.rows * p.D * 4);
    p.ss = align_up((int64_t)2 * e.B * p.N * 4);
    p.total = p.x2 + p.tok + p.core + p.head + p.ss + p.eps;      // eps stays last (DenoiseEngine.eps_tokens reads the tail)
    return AVD_OK;
}

}  // namespace avd

using namespace avd;

extern "C" int avd_abi_version(void) { return AVD_ABI_VERSION; }
extern "C" const char* avd_last_error(void) { return g_err; }

extern "C" int avd_tune_set(const char* key, int64_t value) {
    AVD_REQUIRE(key, AVD_EINVAL, "tune_set: null key");
    if (!strcmp(key, "s3_tile")) { g_s3_tile = (int)value; return AVD_OK; }
    if (!strcmp(key, "s3_m16")) { g_s3_m16 = (int)value; return AVD_OK; }
    if (!strcmp(key, "s3_rt")) { g_s3_rt = (int)value; return AVD_OK; }
    if (!strcmp(key, "s3_rt4")) {
        AVD_REQUIRE(value == 0 || (value >= 2 && value <= 8), AVD_EINVAL, "tune_set: s3_rt4 must be 0 (automatic) or 2 .. 8");
        g_s3_rt4 = (int)value;
        return AVD_OK;
    }
    if (!strcmp(key, "s3_deep4")) { g_s3_deep4 = value != 0; return AVD_OK; }
    if (!strcmp(key, "s3_w128")) { g_s3_w128 = (int)value; return AVD_OK; }
    if (!strcmp(key, "s3_splitk")) {
        AVD_REQUIRE(value >= 0 && value <= kS3SplitKMax, AVD_EINVAL, "tune_set: s3_splitk must be in [0, %d]", kS3SplitKMax);
        g_s3_splitk = (int)value;
        return AVD_OK;
    }
    if (!strcmp(key, "s3_stagger")) { g_s3_stagger = (int)value; return AVD_OK; }
    if (!strcmp(key, "cfg_rows")) { g_cfg_rows = value != 0; return AVD_OK; }
    if (!strcmp(key, "vae_lat")) { g_vae_lat = value != 0; return AVD_OK; }
    if (!strcmp(key, "vae_fold")) { g_vae_fold = value != 0; return AVD_OK; }
    if (!strcmp(key, "codec_mfma")) { g_codec_mfma = value != 0; return AVD_OK; }
    if (!strcmp(key, "s3_sn")) { g_s3_sn = (int)value; return AVD_OK; }
    if (!strcmp(key, "s3_super4")) { g_s3_super4 = (int)value; return AVD_OK; }
    if (!strcmp(key, "s3_super8")) { g_s3_super8 = (int)value; return AVD_OK; }
    if (!strcmp(key, "attn_pipe")) { g_attn_pipe = (int)value; return AVD_OK; }
    if (!strcmp(key, "attn_m16")) { g_attn_m16 = (int)value; return AVD_OK; }
    if (!strcmp(key, "core_trim")) { g_core_trim = value != 0; return AVD_OK; }
    if (!strcmp(key, "mlp_fused")) { g_mlp_fused = (int)value; return AVD_OK; }
    if (!strcmp(key, "gemm_tile")) { g_gemm_force_tile = (int)value; return AVD_OK; }
    if (!strcmp(key, "gemm_stages")) { g_gemm_stages = (int)value; return AVD_OK; }
    if (!strcmp(key, "gemm_splitk")) {
        AVD_REQUIRE(value >= 0 && value <= kGemmSplitKMax, AVD_EINVAL, "tune_set: gemm_splitk must be in [0, %d]", kGemmSplitKMax);
        g_gemm_splitk = (int)value;
        return AVD_OK;
    }
    if (!strcmp(key, "s3_min_rows")) { g_s3_min_rows = value; return AVD_OK; }
    if (!strcmp(key, "no_fold")) { g_no_fold = value != 0; return AVD_OK; }
    return set_error(AVD_EINVAL, "tune_set: unknown key '%s'", key);
}

extern "C" int avd_device_arch(char* buf, int buflen) {
    AVD_REQUIRE(buf && buflen > 0, AVD_EINVAL, "device_arch: bad buffer");
    hipDeviceProp_t prop;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return set_error(AVD_ELAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return set_error(AVD_ELAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    strncpy(buf, prop.gcnArchName, buflen - 1);
    buf[buflen - 1] = 0;
    return AVD_OK;
}

extern "C" int64_t avd_core_workspace_bytes(const avd_core_weights* w, int B, int N) {
    if (!w || B <= 0 || N <= 0) return -1;
    return core_ws_bytes(w, B, N);
}
extern "C" int avd_core_forward_f32(const avd_core_weights* w, const float* x, float* y, int B, int N, int out_row0,
                                    int n_out_rows, const uint8_t* key_padding_mask, void* workspace, int64_t workspace_bytes,
                                    avd_stream_t stream) {
    return core_forward(w, x, y, B, N, out_row0, n_out_rows, key_padding_mask, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream));
}

extern "C" int64_t avd_head_workspace_bytes(const avd_head_weights* w, int64_t rows) {
    if (!w || rows < 0) return -1;
    return head_ws_bytes(w, rows);
}
extern "C" int avd_head_forward_f32(const avd_head_weights* w, const float* h, int64_t ldh, int64_t seg_rows,
                                    int64_t seg_stride, int64_t rows, float* out, void* workspace,
                                    int64_t workspace_bytes, avd_stream_t stream) {
    return head_forward(w, h, RowMap{ldh, seg_rows, seg_stride}, rows, out, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream));
}

extern "C" int64_t avd_embed_workspace_floats(const avd_embed_desc* e) {
    if (check_embed(e)) return -1;
    return embed_ws_floats(e);
}
extern "C" int avd_embed_cfg_pair_f32(const avd_embed_desc* desc, const float* z_target, const float* Wt,
                                      const float* bt, const int64_t* t_now, const float* Xp, float* tok_ws, float* X2,
                                      avd_stream_t stream) {
    return embed_cfg_pair(desc, z_target, Wt, bt, t_now, Xp, tok_ws, X2, static_cast<hipStream_t>(stream));
}

extern "C" int64_t avd_step_workspace_bytes(const avd_step_desc* s) {
    StepPlan p;
    if (plan_step(s, p)) return -1;
    return p.total;
}

extern "C" int avd_denoise_step_f32(const avd_step_desc* s, const float* z, const float* Xp, const int64_t* t_now,
                                    const int64_t* t_prev, const float* noise, float* z_out, void* workspace,
                                    int64_t workspace_bytes, avd_stream_t stream) {
    StepPlan p;
    if (int rc = plan_step(s, p)) return rc;
    AVD_REQUIRE(z && z_out && t_now && t_prev && s->alpha_bar && s->adapt_w, AVD_EINVAL, "step: null pointer");
    AVD_REQUIRE(z != z_out, AVD_EINVAL, "step: z_out must not alias z");
    AVD_REQUIRE(workspace && workspace_bytes >= p.total, AVD_EWORKSPACE, "step: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)p.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const avd_embed_desc& e = s->embed;
    char* w = static_cast<char*>(workspace);
    float* X2 = reinterpret_cast<float*>(w);
    float* tok = reinterpret_cast<float*>(w + p.x2);
    void* core_ws = w + p.x2 + p.tok;
    void* head_ws = w + p.x2 + p.tok + p.core;
    float* ssx = reinterpret_cast<float*>(w + p.x2 + p.tok + p.core + p.head);
    float* eps2 = reinterpret_cast<float*>(w + p.x2 + p.tok + p.core + p.head + p.ss);
    const bool have_ss = !e.temb_add;      // the fused concat front end leaves the rows' sums of squares behind

    if (int rc = embed_cfg_pair(&e, z, s->adapt_w, s->adapt_b, t_now, Xp, tok, X2, st, ssx)) return rc;
    const int row0 = e.target_first ? 0 : e.Np;
    // head over the target rows only (per-token independent, so skipping prompt rows is exact)
    const RowMap hm{e.d, e.Nt, (int64_t)p.N * e.d};
    if (!s->split_streams) {
        if (int rc = core_forward(s->core, X2, X2, 2 * e.B, p.N, row0, e.Nt, nullptr, core_ws, p.core, st, have_ss ? ssx : nullptr)) return rc;
        if (int rc = head_forward(s->head, X2 + (int64_t)row0 * e.d, hm, p.rows, eps2, head_ws, p.head, st)) return rc;
    } else {
        // the cond and null halves are independent until the CFG combine: run them as two kernel chains on two
        // streams so one chain's partially-filled last rounds overlap the other chain's kernels
        AuxState ax;
        if (int rc = ensure_aux(st, ax)) return rc;
        hipStream_t g_aux = ax.aux;
        hipEvent_t g_fork = ax.fork, g_join = ax.join;
        const int64_t half_rows = (int64_t)e.B * p.N * e.d;
        const int64_t hc = p.core / 2, hh = p.head / 2;
        AVD_HIP(hipEventRecord(g_fork, st));
        AVD_HIP(hipStreamWaitEvent(g_aux, g_fork, 0));
        struct TwoStreams {
            TwoStreams(int64_t rows, int64_t head_rows) { t_s3_two_streams = true; t_step_rows = rows; t_step_head_rows = head_rows; }
            ~TwoStreams() { t_s3_two_streams = false; t_step_rows = 0; t_step_head_rows = 0; }
        } two_streams_scope((int64_t)2 * e.B * p.N, p.rows);
        for (int half = 0; half < 2; ++half) {
            hipStream_t hs = half ? g_aux : st;
            float* xh = X2 + half * half_rows;
            if (int rc = core_forward(s->core, xh, xh, e.B, p.N, row0, e.Nt, nullptr, static_cast<char*>(core_ws) + half * hc, hc, hs,
                                      have_ss ? ssx + (int64_t)half * e.B * p.N : nullptr)) return rc;
            if (int rc = head_forward(s->head, xh + (int64_t)row0 * e.d, hm, p.rows / 2, eps2 + half * (p.rows / 2) * p.D,
                                      static_cast<char*>(head_ws) + half * hh, hh, hs)) return rc;
        }
        AVD_HIP(hipEventRecord(g_join, g_aux));
        AVD_HIP(hipStreamWaitEvent(st, g_join, 0));
    }
    if (e.target_kind == 0)
        return cfg_unpatch_ddim_f32(eps2, z, t_now, t_prev, s->alpha_bar, s->T_train, s->guidance, s->eta, noise, z_out,
                                    e.B, e.C, e.T, e.H, e.W, e.p0, e.p1, e.p2, st);
    return cfg_untoken_ddim_audio_f32(eps2, z, t_now, t_prev, s->alpha_bar, s->T_train, s->guidance, s->eta, noise, z_out,
                                      e.B, e.C, e.T, e.p0, e.p1, st);
}

extern "C" int avd_prof_enable(int on) {
    if (on) {   // recycle the previous run's events; records survive a disable so they can be reported
        for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
        g_recs.clear();
    }
    g_prof_on = on != 0;
    return AVD_OK;
}

extern "C" int avd_prof_num_tags(void) { return (int)g_tags.size(); }
extern "C" const char* avd_prof_tag_name(int tag) { return (tag >= 0 && tag < (int)g_tags.size()) ? g_tags[tag].c_str() : ""; }

extern "C" int avd_prof_report(int64_t* launches, double* total_ms, double* work, int ntags) {
    AVD_REQUIRE(launches && total_ms && work && ntags >= (int)g_tags.size(), AVD_EINVAL, "prof_report: need %d tags",
                (int)g_tags.size());
    for (int i = 0; i < ntags; ++i) { launches[i] = 0; total_ms[i] = 0.0; work[i] = 0.0; }
    for (auto& r : g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return set_error(AVD_ELAUNCH, "prof_report: %s", hipGetErrorString(e));
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return set_error(AVD_ELAUNCH, "prof_report: %s", hipGetErrorString(e));
        launches[r.tag] += 1; total_ms[r.tag] += ms; work[r.tag] += r.work;
    }
    return AVD_OK;
}
