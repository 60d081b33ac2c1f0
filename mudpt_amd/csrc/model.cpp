// Host side of libmudpt_hip.so: owns the frozen weights and activations in HBM and sequences the
// gfx950 kernels of one MuDPT forward / forward+backward on a HIP stream.
//
// Data layout in HBM (B images, C class prompts, ViT-B/16 numbers in brackets):
//   tokens are batch-first [seq][L][d], never padded: vision L = 1 + P + n_ctx [201], text L = ctx_len [77];
//   the residual stream and its gradient are fp32; GEMM / attention operands (LN output, qkv, attention
//   output, MLP pre-activation, their gradients) are T = bf16 or fp16;
//   frozen Linear weights are stored twice in T: [out,in] for the forward GEMM and [in,out] for the dX GEMM
//   (weights are frozen, so the transposed copy is made once at load; no dW is ever computed);
//   per block the backward needs x_in, x_mid (fp32), LN statistics, qkv, attention output + LSE and the
//   MLP pre-activation: 0.95 GB per block at B = 256, 11.4 GB for the vision tower (288 GB HBM: no recompute).
// Prompt rows are ordinary rows of the token buffer: the reference's torch.cat splices
// (clip/model.py:281-297) are in-place row writes, their backward a fixed-order reduction over the batch.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/mudpt.h"
#include "kernels.h"

namespace mudpt {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// ---- host-side conversion to the operand dtype -----------------------------------------------------
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline uint16_t f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

// OCP e4m3 (fn: no infinities, maximum 448), round to nearest even, saturating: the weight copies of the e4m3 second pass (common.h LO_F8)
static inline uint8_t f32_to_e4m3(float f) {
    const uint8_t sign = std::signbit(f) ? 0x80 : 0;
    const float a = std::fabs(f);
    if (!(a == a)) return 0x7f;
    if (a >= 464.f) return sign | 0x7e;                   // past the midpoint to the (non-existent) next value: the maximum 448
    if (a <= std::ldexp(1.f, -10)) return sign;           // at most half the smallest subnormal 2^-9: zero (the tie goes to even)
    int e;
    (void)std::frexp(a, &e);                              // a = m 2^e, m in [0.5, 1)
    int E = e - 1 < -6 ? -6 : e - 1;                      // binade (subnormals share 2^-6)
    float r = std::nearbyint(a / std::ldexp(1.f, E - 3)); // in units of 2^(E - 3): [8, 16) normal, [0, 8) subnormal; RNE (default rounding mode)
    if (r >= 16.f) { r = 8.f; E += 1; }
    uint8_t bits = r < 8.f ? (uint8_t)r : (uint8_t)(((E + 7) << 3) | ((int)r - 8));
    if ((bits & 0x7f) == 0x7f) bits = 0x7e;               // 480 would be the NaN code: saturate
    return sign | bits;
}

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct BlockW {
    void *w_in = nullptr, *w_in_t = nullptr, *w_out = nullptr, *w_out_t = nullptr;
    void *w_fc = nullptr, *w_fc_t = nullptr, *w_proj = nullptr, *w_proj_t = nullptr;
    // e4m3 copies of the four forward weights for the second pass of a LO_F8 split tower (common.h): rows of the T copy's length IN BYTES
    // (the first `in` bytes used), values W 2^shift with the per-tensor shift that brings max |W| just below 448; s_* = the MX block scale
    // (E8M0) 2^-shift the matrix instruction applies
    void *w_in8 = nullptr, *w_out8 = nullptr, *w_fc8 = nullptr, *w_proj8 = nullptr;
    int s_in8 = 127, s_out8 = 127, s_fc8 = 127, s_proj8 = 127;
    float *b_in = nullptr, *b_out = nullptr, *b_fc = nullptr, *b_proj = nullptr;
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
};

struct BlockAct {
    float *x_in = nullptr, *x_mid = nullptr;  // fp32 [M, d]
    float *mean1 = nullptr, *rstd1 = nullptr, *mean2 = nullptr, *rstd2 = nullptr;
    void *qkv = nullptr, *attn = nullptr, *u = nullptr;  // T
    float* lse = nullptr;
};

struct Tower {
    int d = 0, layers = 0, heads = 0, L = 0, max_seq = 0, Lp = 0;
    bool causal = false;
    // Split operands (common.h LoMode).  The forward GEMMs' A operands -- both LayerNorm outputs, the attention output and QuickGELU(u) --
    // are stored as hi = T(v) plus the remainder lo in a second buffer, and the GEMM contracts lo in a second pass: against the same
    // weights (LO_F16: 22 bits; CLIP's weights are fp16-exact, checkpoints are fp16-stored) or, as e4m3 bytes, against an e4m3 copy of
    // the weights on the fp8 matrix pipe (LO_F8: ~15 bits at half the cost of the pass).  Which the modes use, and what each site buys:
    // DESIGN.md 2 (tests/precision_ablation.py: the text tower needs every bit at every site, the vision tower's error is c_fc / c_proj first).
    int split = LO_NONE;
    bool may_split = false;   // low-half buffers (and, vision tower, the e4m3 weights) exist: split / sites / exact_attn are knobs
    int sites = 0x1f;         // knob: sites that take part -- bit 0 ln_1 -> in_proj, 1 attention -> out_proj, 2 ln_2 -> c_fc, 3 QuickGELU -> c_proj, 4 pixels -> patch embed
    void *h_lo = nullptr, *g_lo = nullptr, *attn_lo = nullptr;               // low halves of h, g, the attention output: rows of 2 d / 8 d / 2 d bytes
    void *h_sel_lo = nullptr, *g_sel_lo = nullptr, *attn_sel_lo = nullptr;   // ... of the last block's tail
    int prompt_row0 = 0;  // first prompt row inside a sequence (vision: L - n, text: 1)
    std::vector<BlockW> w;
    std::vector<BlockAct> a;  // layers entries; the last block's output exists on the tail rows only (xout_sel)
    // scratch shared by all blocks
    void *h = nullptr, *g = nullptr;                    // T [M,d], T [M,4d]
    // Attention forward in fp32 (attention_exact.hip; the parity mode's text tower, knob for its vision tower): in_proj writes q | k | v in
    // fp32 to qkv32, the attention kernel leaves the fp16 copy the backward reads.
    bool exact_attn = false;
    float* qkv32 = nullptr;                             // fp32 [M, 3d] scratch (allocated when may_split && the mode is MUDPT_F32)
    float* dx = nullptr; void* dx_lp = nullptr;         // gradient residual stream fp32 + T copy
    void *dattn = nullptr, *dqkv = nullptr;             // T
    float* delta = nullptr;
    float* upd = nullptr;  // fp32 [M, d]: out_proj / c_proj result, added to the stream by the next LayerNorm kernel
    // Tail of the last block.  Only ONE row per sequence of the last block's output is ever used (the CLS token,
    // clip/model.py:549, or the EOT token, trainers/mudpt.py:154), so everything after that block's attention -- out_proj,
    // ln_2, the MLP, the residual adds and their backward -- runs on those max_seq rows only ("sel": compact [max_seq, *]).
    // Results are identical to the reference's: the other rows of its last block output are computed and dropped.
    const int* tail_rows = nullptr;  // [nseq] token row (b * L + position) of the used row of every sequence
    float *xin_sel = nullptr, *xmid_sel = nullptr, *xout_sel = nullptr;  // fp32 [S, d]
    void *attn_sel = nullptr, *h_sel = nullptr, *u_sel = nullptr, *g_sel = nullptr, *dattn_sel = nullptr;  // T
    float* dsel = nullptr; void* dsel_lp = nullptr;  // gradient of the residual stream on the selected rows (fp32 / T)
    // single-query attention of the last block (attention_single.hip): the one query per sequence, its gradient, its log-sum-exp
    void *q_sel = nullptr, *dq_sel = nullptr, *dqx_sel = nullptr;  // T [S, d]
    float* lse_sel = nullptr;                                       // [S, heads]
    // Head of the backward pass: block 0's input gradient is only needed on the n_ctx prompt rows of every sequence (the
    // other rows of the tower input have no trainable ancestor), so its in_proj dX GEMM and ln_1 backward run on those rows.
    const int* head_rows = nullptr;  // [nseq * n_ctx] token rows of the prompt tokens
    int head_n = 0;                  // rows per sequence
    void *hd_dqkv = nullptr, *hd_h = nullptr;  // T [max_seq * n_ctx, 3 d], [max_seq * n_ctx, d]
    std::vector<void*> act_allocs;  // activation / scratch buffers (sized for max_seq sequences of L rows): re-made when L changes
    // Length buckets (text tower with many classes): the sequences are sorted by length and packed bucket after bucket, each bucket with
    // its own row count per sequence, so the row-wise kernels (GEMMs, LayerNorm) run once over `rows` packed rows while the kernels that
    // know about sequences (attention, prompt splice, prompt-row reductions) run once per bucket.  Empty = one bucket of max_seq x L.
    struct Seg { int row0 = 0, seq0 = 0, nseq = 0, L = 0; size_t lse0 = 0; };
    std::vector<Seg> segs;
    int rows = 0;                        // packed rows (segs non-empty)
    const int* tail_local = nullptr;     // [nseq] tail row of a sequence relative to its bucket's first row (single-query attention)
};

// rows of a tower pass over nseq sequences; its buckets (one pseudo-bucket when the tower is not packed)
static int tower_rows(const Tower& t, int nseq) { return t.segs.empty() ? nseq * t.L : t.rows; }
static std::vector<Tower::Seg> tower_segs(const Tower& t, int nseq) {
    if (!t.segs.empty()) return t.segs;
    Tower::Seg one; one.nseq = nseq; one.L = t.L;
    return {one};
}

}  // namespace mudpt

using namespace mudpt;

struct mudpt_model {
    mudpt_config cfg;
    int dtype = 0;
    bool exact = false;  // mudpt_config.dtype == MUDPT_F32: fp16 split operands everywhere on the forward + fp32 attention forward
    std::vector<void*> allocs;
    std::vector<std::string> missing;  // weight keys not yet set
    bool prompts_set = false;
    bool text_valid = false;  // txt_f holds the text features of the currently bound parameter values' last forward

    Tower vis, txt;
    // vision stem / head
    void* conv_w = nullptr;  // T [dv, K0]
    void* conv_w8 = nullptr; int conv_s8 = 127;  // parity mode: its e4m3 copy (rows of 2 K0 bytes) + block scale, BlockW::w_in8
    void* patches_lo = nullptr;                  // ... and the low halves of the pixels
    float *cls = nullptr, *vpos = nullptr, *ln_pre_g = nullptr, *ln_pre_b = nullptr, *ln_post_g = nullptr, *ln_post_b = nullptr;
    float* vproj = nullptr;  // [dv, e]
    void* patches = nullptr; // T [B P, 3 p p]
    float *xpre = nullptr, *pre_mean = nullptr, *pre_rstd = nullptr;
    float *f_ln = nullptr, *post_mean = nullptr, *post_rstd = nullptr, *df_ln = nullptr;
    int *cls_rows = nullptr, *vprompt_rows = nullptr;
    // text stem / head
    float *tpos = nullptr, *ln_fin_g = nullptr, *ln_fin_b = nullptr, *tproj = nullptr;
    float* emb_pos = nullptr;  // [C, Lt, dt] class token embeddings + positional embedding
    int *eot_rows = nullptr, *tprompt_rows = nullptr;
    int *eot_local = nullptr, *class_perm = nullptr;  // EOT row relative to the sequence's length bucket; packed position -> local class
    float *txt_sorted = nullptr, *dtxt_sorted = nullptr;  // [C, e] text features / their gradient in packed (length-sorted) order
    float *t_ln = nullptr, *fin_mean = nullptr, *fin_rstd = nullptr, *dt_ln = nullptr;
    float scale = 1.f;
    // prompt learner intermediates (fp32)
    float *shared = nullptr, *t2v = nullptr, *v2t = nullptr, *vis_deep = nullptr, *txt_deep = nullptr;
    float *d_vis_deep = nullptr, *d_txt_deep = nullptr, *d_vprompt0 = nullptr;
    float* vsplice = nullptr;  // [max_batch][(depth - 1) n_ctx][dv] fp32: per-image gradients of the spliced vision prompt rows (vision_backward)
    // head
    float *img_f = nullptr, *txt_f = nullptr, *img_n = nullptr, *txt_n = nullptr, *img_inv = nullptr, *txt_inv = nullptr;
    float *logits = nullptr, *dlogits = nullptr, *row_loss = nullptr, *dimg = nullptr, *dtxt = nullptr, *loss = nullptr;
    // parameters
    float *params = nullptr, *grads = nullptr, *momentum = nullptr;
    bool sgd_first = true;
    size_t off[10];
    size_t numel[10];
    size_t total = 0;
    // CoCoOp variant (trainers/cocoop.py): 5 trainables, vanilla vision tower (forward only), B * C text sequences
    bool cocoop = false;
    int nparams = 10;
    // Class-parallel text tower (second axis): this handle encodes classes [c0, c0 + ct) of the n_cls only; the
    // [n_cls, e] text-feature table is completed by the caller's exchange between the mudpt_cp_* phases.  Default: all classes.
    int c0 = 0, ct = 0;
    bool sharded = false;
    float cp_unscale = 0.f;  // of the step in flight between mudpt_cp_head and mudpt_cp_backward
    int cp_B = 0, cp_stage = 0;  // 1 = towers forward done, 2 = head (training) done
    int hid = 0;  // meta_net hidden width = embed_dim / 16 (trainers/cocoop.py:104)
    float *mn_hid = nullptr, *mn_bias = nullptr, *mn_dbias = nullptr, *mn_dhid = nullptr;  // [B, hid], [B, dt], [B, dt], [B, hid]
    float loss_scale = 128.f;  // static, power of two; see mudpt_forward_backward
    // bf16 mode keeps the gradient of the residual stream in T only (the fp32 copy costs 237 MB of HBM traffic per LayerNorm
    // backward); fp16 mode -- the parity configuration -- keeps it in fp32.  mudpt_model_set("lp_grad", 1) moves fp16 mode to fp16
    // activation gradients as well (what the reference's own fp16 model has; the static loss scale keeps them normal): measured on
    // the fixtures, logits unchanged (forward only), gradient errors +30 % (ViT-B/16 worst max-error 1.8e-2 -> 2.5e-2 of the tensor's
    // RMS; CoCoOp's meta_net gradients reach the edge of the test's error model), step 26.5 -> 25.6 ms.  Not the default: the parity
    // mode exists for accuracy.
    bool lp_grad = false;
    // bf16 mode: c_fc stores QuickGELU'(u) in 8 bits for the backward instead of u in T (common.h gelu_grad_q8x4): -158 MB written and
    // -158 MB read per MLP and step at an error of 2.4e-3 on a factor in [-0.1, 1.1] -- bf16's own grade.  Knob gelu_q8.
    bool gelu_q8 = false;
    bool lp_upd = false;   // the forward's update stream (out_proj / c_proj results added by the next LayerNorm) in T instead of fp32: bf16 mode
    // per-handle tuning knobs (mudpt_model_set): nothing here is process-global, two models in one process do not interfere
    int gemm_variant = 0;
    bool txt_trim = true;  // run the text tower on positions 0..max(eot) only (read by mudpt_set_class_prompts)
    int txt_bucket_cost = 1024;  // knob: what one more bucket costs in the cut search, in token rows (its extra launches per block)
    int txt_buckets = 3;   // knob: at most this many length buckets for the class prompts (1 = every prompt runs to the longest EOT)
    bool attn_fused_w1 = false;
    bool split_k = true;      // knob: split K for the small-grid, long-K store GEMMs (small batches)
    bool fwd_split_k = true;  // knob: ... of the forward too, up to kFwdSplitTiles tiles (gemm_call) -- in a TRAINING step's forward only
    bool train_fwd = false;   // the forward in flight belongs to a training step (mudpt_forward_backward, mudpt_cp_forward with MUDPT_FWD_TRAINING):
                              // inference forwards never split K, so eval logits of an image do not depend on the size of its (last, partial) test batch
    static constexpr size_t kFwdSplitTiles = 320;
    static constexpr size_t kScratchElems = (size_t)4 << 20;  // 4 slices x 128 tiles of 128 x 64 fp32
    float *gemm_scratch = nullptr, *gemm_scratch2 = nullptr;
    bool attn_window = true;  // knob: block 0's attention backward computes the 16-row blocks of the prompt rows only (0 = all rows)
    bool last_single = true;  // knob: single-query attention in the last block (0 = the general kernels on all rows)
    bool attn_two_kernels = false;  // knob: attention backward as the dQ + dK/dV kernel pair instead of the fused single pass
    int cocoop_chunk = 0;  // knob: cap on the images per text-tower pass (0 = as many as the memory budget allows)
    int txt_chunk = 1;     // CoCoOp: images per text-tower pass, set by mudpt_set_class_prompts
    bool any_weight_set = false;
    // side stream for the text tower (forks after the prompt learner / head backward, joins before the head /
    // prompt-learner backward)
    hipStream_t s2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork_b = nullptr, ev_join_b = nullptr;
    // optional HIP-event timing of the big vision-tower launches (bench.py's roofline legs), by kernel class
    bool prof = false;
    int prof_mask = 0;  // bit c: launches of class c are bracketed (an event pair costs ~5 us of queue time per launch)
    int prof_stride = 1;  // knob: every prof_stride-th gemm_pp launch is bracketed (launch counter runs across steps: with a launch
                          // count per step coprime to the stride, every launch site is sampled equally often over `stride` steps)
    long pp_seen = 0;
    std::vector<hipEvent_t> ev;  // pairs
    size_t ev_used = 0;
    struct ProfRec { int cls; double work; };
    std::vector<ProfRec> ev_rec;  // one per event pair: class and algorithmic work (FLOPs for the GEMM, bytes for the HBM-bound kernels)
    double exec_flop = 0;         // executed MFMA FLOPs (every GEMM and attention launch of both towers) since profile_enable / read
};
enum ProfClass : int { PC_GEMM = 0, PC_LN_FWD = 1, PC_LN_BWD = 2, PC_ATTN_FWD = 3, PC_ATTN_BWD = 4, PC_COUNT = 5 };

// Next event pair for a bracketed launch of class cls (nullptr-filled when profiling is off).
static int prof_next(mudpt_model* m, int cls, double work, LaunchProf* out) {
    *out = LaunchProf();
    if (!m->prof || !(m->prof_mask & (1 << cls))) return MUDPT_OK;
    if (m->ev_used + 2 > m->ev.size()) {
        for (int i = 0; i < 512; ++i) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            m->ev.push_back(e);
        }
    }
    out->start = m->ev[m->ev_used];
    out->stop = m->ev[m->ev_used + 1];
    m->ev_used += 2;
    m->ev_rec.push_back({cls, work});
    return MUDPT_OK;
}

// Every MFMA GEMM of the path goes through here; with profiling on, the launch is bracketed by HIP events on
// the launch stream and its algorithmic FLOPs (2 M N K) are recorded.
// bwd: a GEMM of the backward pass.  Those may always split K; a FORWARD GEMM only where the caller allows it (fwd_split: the vision
// tower's out_proj / c_proj, never with split operands) and its grid is at most kFwdSplitTiles 64 x 64 tiles (ViT-B: up to 8 images -- the
// reference's own training batch of 4, where c_proj's 48-step chain on 156 workgroups was the longest kernel of the step).  The text
// tower never splits, forward or backward (its call sites pass bwd = !t.causal): its features are bit-identical and its gradients equal to
// the order of the fp32 sums over classes however the class prompts are bucketed (a tested property; a split decision that depends on the
// row count of a bucketing moved the fp16 gradients by 1e-2 of their rms, the size of the fp16 backward's whole rounding noise).  A split sum has a different fp32 association than the sequential one and whether a shape splits depends on M, so the
// tested property "logits of a batch equal the logits of its chunks bit for bit" holds for chunks ABOVE that size (knob fwd_split_k = 0
// restores it for every size); gradients are only held to agree to rounding.
static int gemm_call(mudpt_model* m, int epi, const GemmArgs& a, hipStream_t s, bool bwd = false, bool fwd_split = false) {
    // only the dominant kernel is bracketed: gemm_pp_kernel launches (the vision tower's big GEMMs, main stream).  The text
    // tower's small GEMMs run on the side stream, where an event pair would mostly measure queueing behind the other stream.
    GemmOpts o;
    o.variant = m->gemm_variant;
    const bool fwd_small = fwd_split && m->fwd_split_k && m->train_fwd && (size_t)((a.M + 63) / 64) * ((a.N + 63) / 64) <= mudpt_model::kFwdSplitTiles;
    if (m->split_k && (bwd || fwd_small)) {  // split-K partials: one scratch per stream (the towers run concurrently)
        o.scratch = s == m->s2 ? m->gemm_scratch2 : m->gemm_scratch;
        o.scratch_elems = mudpt_model::kScratchElems;
    }
    if (m->prof) m->exec_flop += 2.0 * a.M * a.N * a.K;
    if (!m->prof || !gemm_uses_pp(epi, a, o.variant)) return launch_gemm(m->dtype, epi, a, s, o);
    if (m->prof_stride > 1 && (m->pp_seen++ % m->prof_stride) != 0) return launch_gemm(m->dtype, epi, a, s, o);
    LaunchProf lp;
    if (int rc = prof_next(m, PC_GEMM, 2.0 * a.M * a.N * a.K, &lp)) return rc;
    o.ev_start = lp.start;
    o.ev_stop = lp.stop;
    return launch_gemm(m->dtype, epi, a, s, o);
}

// LayerNorm / attention launches of the path.  The big vision-tower ones (main stream) are bracketed when profiling is on; their
// work figure is the ALGORITHMIC HBM bytes of the launch: every operand read once, every result written once.
static bool prof_big(const mudpt_model* m, const Tower& t, int rows) { return m->prof && &t == &m->vis && rows >= 4096; }
static int ln_fwd_call(mudpt_model* m, const Tower& t, const LnFwdArgs& a, hipStream_t s) {
    if (!prof_big(m, t, a.rows)) return launch_ln_fwd(m->dtype, a, s);
    const double per_elem = 4.0 + (a.add ? 4.0 : 0.0) + (a.add_lp ? 2.0 : 0.0) + (a.xout ? 4.0 : 0.0) + (a.out_f32 ? 4.0 : 2.0) + (a.out_lo ? 2.0 : 0.0);
    LaunchProf lp;
    if (int rc = prof_next(m, PC_LN_FWD, per_elem * a.rows * a.d, &lp)) return rc;
    return launch_ln_fwd(m->dtype, a, s, &lp);
}
static int ln_bwd_call(mudpt_model* m, const Tower& t, const LnBwdArgs& a, hipStream_t s) {
    if (!prof_big(m, t, a.rows)) return launch_ln_bwd(m->dtype, a, s);
    const double per_elem = (a.dy_f32 ? 4.0 : 2.0) + 4.0 + (a.dres ? 4.0 : 0.0) + (a.dres_lp ? 2.0 : 0.0) + (a.dx ? 4.0 : 0.0) + (a.dx_lp ? 2.0 : 0.0);
    LaunchProf lp;
    if (int rc = prof_next(m, PC_LN_BWD, per_elem * a.rows * a.d, &lp)) return rc;
    return launch_ln_bwd(m->dtype, a, s, &lp);
}
static int attn_call(mudpt_model* m, const Tower& t, const AttnArgs& a0, bool bwd, hipStream_t s) {
    AttnArgs a = a0;
    a.two_kernels = m->attn_two_kernels;
    a.fused_w1 = m->attn_fused_w1;
    // executed MFMA FLOPs: forward S = QK^T and PV (2 products of 2 L^2 64 each per head); backward 7 products (dQ sweep: S, dP, dQ;
    // dK/dV sweep: S, dP, dV, dK); the causal tower does about half of each
    const double prod = 2.0 * a.L * (double)a.L * 64.0 * a.H * a.B * (a.causal ? 0.5 : 1.0);
    const bool exact_fwd = !bwd && a.qkv32;  // fp32 attention forward: fp32 matrix-core FLOPs are not counted as executed bf16 / fp16 MFMA work
    if (m->prof && !a.sel_rows && !exact_fwd) m->exec_flop += (bwd ? 7.0 : 2.0) * prod;
    if (exact_fwd) return launch_attn_fwd_exact(a, s);
    if (!prof_big(m, t, a.B * a.L) || a.sel_rows) return bwd ? launch_attn_bwd(m->dtype, a, s) : launch_attn_fwd(m->dtype, a, s);
    // algorithmic bytes: forward reads q, k, v and writes o (+ its low half in split mode); backward reads q, k, v, o, do and writes dq, dk, dv
    const double tok = (double)a.B * a.L * a.H * 128.0;
    LaunchProf lp;
    if (int rc = prof_next(m, bwd ? PC_ATTN_BWD : PC_ATTN_FWD, bwd ? 8.0 * tok : (4.0 + (a.out_lo ? 1.0 : 0.0)) * tok, &lp)) return rc;
    return bwd ? launch_attn_bwd(m->dtype, a, s, &lp) : launch_attn_fwd(m->dtype, a, s, &lp);
}

static const char* kParamNames[10] = {
    "mudpt_prompt_learner.ctx",
    "mudpt_prompt_learner.deep_prompts",
    "mudpt_prompt_learner.embed_projection.weight",
    "mudpt_prompt_learner.embed_projection.bias",
    "mudpt_prompt_learner.deep_projections.weight",
    "mudpt_prompt_learner.deep_projections.bias",
    "image_encoder.visual_ctx",
    "image_encoder.visual_ctx_deep_prompts",
    "image_encoder.visual_ctx_deep_projections.weight",
    "image_encoder.visual_ctx_deep_projections.bias",
};
enum { P_CTX = 0, P_DEEP, P_EW, P_EB, P_DW, P_DB, P_VCTX, P_VDEEP, P_VW, P_VB };
// CoCoOp: names under CustomCLIP (trainers/cocoop.py:96-107,176; the reference registers the prompt_learner sub-module)
static const char* kCocoopNames[5] = {
    "prompt_learner.ctx",
    "prompt_learner.meta_net.linear1.weight",
    "prompt_learner.meta_net.linear1.bias",
    "prompt_learner.meta_net.linear2.weight",
    "prompt_learner.meta_net.linear2.bias",
};
enum { Q_CTX = 0, Q_W1, Q_B1, Q_W2, Q_B2 };


static int dev_alloc(mudpt_model* m, void** out, size_t bytes) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    m->allocs.push_back(p);
    *out = p;
    return MUDPT_OK;
}
#define ALLOC(ptr, bytes)                                              \
    do {                                                               \
        if (int _e = dev_alloc(m, (void**)&(ptr), (size_t)(bytes))) return _e; \
    } while (0)

// Frozen weights of a tower (device copies; filled by mudpt_set_weight).
static int alloc_tower_weights(mudpt_model* m, Tower& t, int d, int layers, int heads, bool causal, int prompt_row0, bool f8_weights) {
    t.d = d; t.layers = layers; t.heads = heads; t.causal = causal; t.prompt_row0 = prompt_row0;
    t.w.resize(layers);
    t.a.resize(layers);
    for (int i = 0; i < layers; ++i) {
        BlockW& w = t.w[i];
        ALLOC(w.w_in, (size_t)3 * d * d * 2); ALLOC(w.w_in_t, (size_t)3 * d * d * 2);
        ALLOC(w.w_out, (size_t)d * d * 2); ALLOC(w.w_out_t, (size_t)d * d * 2);
        ALLOC(w.w_fc, (size_t)4 * d * d * 2); ALLOC(w.w_fc_t, (size_t)4 * d * d * 2);
        ALLOC(w.w_proj, (size_t)4 * d * d * 2); ALLOC(w.w_proj_t, (size_t)4 * d * d * 2);
        if (f8_weights) {  // e4m3 copies at the T copies' row length in bytes
            ALLOC(w.w_in8, (size_t)3 * d * d * 2); ALLOC(w.w_out8, (size_t)d * d * 2);
            ALLOC(w.w_fc8, (size_t)4 * d * d * 2); ALLOC(w.w_proj8, (size_t)d * 4 * d * 2);
        }
        ALLOC(w.b_in, 3 * d * 4); ALLOC(w.b_out, d * 4); ALLOC(w.b_fc, 4 * d * 4); ALLOC(w.b_proj, d * 4);
        ALLOC(w.ln1_g, d * 4); ALLOC(w.ln1_b, d * 4); ALLOC(w.ln2_g, d * 4); ALLOC(w.ln2_b, d * 4);
    }
    return MUDPT_OK;
}

// Bytes of activations + scratch per token row of a tower (what alloc_tower_acts takes per row; sizes the CoCoOp chunk).
static size_t tower_bytes_per_row(const Tower& t) {
    const size_t d = t.d, sp = 2;  // low halves counted always (upper bound)
    const size_t per_layer = d * 4 * 2 + d * 3 * 2 + d * 2 + d * 4 * 2 + 16 + (size_t)t.heads * 4 * 2;
    const size_t shared = d * 2 * sp * 2 + d * 4 * 2 * sp + d * 4 + d * 2 + d * 2 + d * 3 * 2 + d * 4 + (size_t)t.heads * 4 * 2 + d * 3 * 4;
    return per_layer * t.layers + shared;
}

// Activations saved for the backward + scratch, for max_seq sequences of L rows.  The vision tower's are made at create; the
// text tower's when the class prompts (and with them its trimmed length) are known -- and again if those change.
static int alloc_tower_acts(mudpt_model* m, Tower& t, int L, int max_seq) {
    for (void* p : t.act_allocs) (void)hipFree(p);
    t.act_allocs.clear();
#define ALLOC_T(ptr, bytes)                                                      \
    do {                                                                         \
        void* _p = nullptr;                                                      \
        HIP_TRY(hipMalloc(&_p, (size_t)(bytes) ? (size_t)(bytes) : 16));         \
        t.act_allocs.push_back(_p);                                              \
        *(void**)&(ptr) = _p;                                                    \
    } while (0)
    const int d = t.d, heads = t.heads;
    t.L = L; t.max_seq = max_seq;
    t.Lp = attn_padded_len(L);
    const size_t M = (size_t)max_seq * L;
    for (int i = 0; i < t.layers; ++i) {
        BlockAct& a = t.a[i];
        ALLOC_T(a.x_in, M * d * 4); ALLOC_T(a.x_mid, M * d * 4);
        ALLOC_T(a.mean1, M * 4); ALLOC_T(a.rstd1, M * 4); ALLOC_T(a.mean2, M * 4); ALLOC_T(a.rstd2, M * 4);
        ALLOC_T(a.qkv, M * 3 * d * 2); ALLOC_T(a.attn, M * d * 2); ALLOC_T(a.u, M * 4 * d * 2);
        ALLOC_T(a.lse, (size_t)max_seq * heads * t.Lp * 4);
    }
    ALLOC_T(t.h, M * d * 2); ALLOC_T(t.g, M * 4 * d * 2);
    if (t.may_split) {  // the low halves of the split operands: the row length of their T counterparts in bytes, whatever their form
        ALLOC_T(t.h_lo, M * d * 2); ALLOC_T(t.g_lo, M * 4 * d * 2); ALLOC_T(t.attn_lo, M * d * 2);
    }
    ALLOC_T(t.dx, M * d * 4); ALLOC_T(t.dx_lp, M * d * 2);
    ALLOC_T(t.dattn, M * d * 2); ALLOC_T(t.dqkv, M * 3 * d * 2);
    ALLOC_T(t.delta, (size_t)max_seq * heads * t.Lp * 4);
    ALLOC_T(t.upd, M * d * 4);
    if (t.may_split && m->exact) ALLOC_T(t.qkv32, M * 3 * d * 4);
    const size_t S = (size_t)max_seq;
    ALLOC_T(t.xin_sel, S * d * 4); ALLOC_T(t.xmid_sel, S * d * 4); ALLOC_T(t.xout_sel, S * d * 4);
    ALLOC_T(t.attn_sel, S * d * 2); ALLOC_T(t.h_sel, S * d * 2); ALLOC_T(t.u_sel, S * 4 * d * 2); ALLOC_T(t.g_sel, S * 4 * d * 2); ALLOC_T(t.dattn_sel, S * d * 2);
    if (t.may_split) { ALLOC_T(t.attn_sel_lo, S * d * 2); ALLOC_T(t.h_sel_lo, S * d * 2); ALLOC_T(t.g_sel_lo, S * 4 * d * 2); }
    ALLOC_T(t.dsel, S * d * 4); ALLOC_T(t.dsel_lp, S * d * 2);
    ALLOC_T(t.q_sel, S * d * 2); ALLOC_T(t.dq_sel, S * d * 2); ALLOC_T(t.dqx_sel, S * d * 2); ALLOC_T(t.lse_sel, S * heads * 4);
    t.head_n = m->cfg.n_ctx;
    ALLOC_T(t.hd_dqkv, S * t.head_n * 3 * d * 2); ALLOC_T(t.hd_h, S * t.head_n * d * 2);
#undef ALLOC_T
    return MUDPT_OK;
}

static void expect_block_keys(mudpt_model* m, const std::string& prefix, int layers) {
    static const char* names[] = {"ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight",
                                  "attn.out_proj.bias", "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias",
                                  "mlp.c_proj.weight", "mlp.c_proj.bias"};
    for (int i = 0; i < layers; ++i)
        for (const char* n : names) m->missing.push_back(prefix + ".resblocks." + std::to_string(i) + "." + n);
}

extern "C" int mudpt_abi_version(void) { return MUDPT_ABI_VERSION; }
extern "C" const char* mudpt_last_error(void) { return get_error(); }

extern "C" int mudpt_create(const mudpt_config* c, mudpt_model** out) {
    ARG_CHECK(c && out, "create: null argument");
    ARG_CHECK(c->dtype == MUDPT_BF16 || c->dtype == MUDPT_F16 || c->dtype == MUDPT_F32, "create: dtype must be MUDPT_BF16, MUDPT_F16 or MUDPT_F32");
    ARG_CHECK(c->variant == MUDPT_VARIANT_MUDPT || c->variant == MUDPT_VARIANT_COCOOP, "create: unknown variant %d", c->variant);
    const bool cocoop = c->variant == MUDPT_VARIANT_COCOOP;
    ARG_CHECK(cocoop || c->depth > 0, "PROMPT_DEPTH should be > 0");  // trainers/mudpt.py:52
    ARG_CHECK(c->n_ctx > 0 && c->n_cls > 0 && c->max_batch > 0, "create: n_ctx, n_cls, max_batch must be positive");
    ARG_CHECK(c->patch > 0 && c->image_size % c->patch == 0, "create: image_size %d / patch %d unsupported", c->image_size, c->patch);
    ARG_CHECK(c->v_width == c->v_heads * 64 && c->t_width == c->t_heads * 64, "create: head dim must be 64");
    ARG_CHECK(c->v_width % 64 == 0 && c->t_width % 64 == 0 && c->v_width <= 1024 && c->t_width <= 1024, "create: widths must be multiples of 64, <= 1024");
    ARG_CHECK(c->embed_dim == c->t_width, "create: embed_dim must equal t_width (visual_ctx_deep_projections output is added to text prompts)");
    ARG_CHECK(1 + c->n_ctx < c->ctx_len, "create: n_ctx too large for ctx_len");
    const int P = (c->image_size / c->patch) * (c->image_size / c->patch);
    const int Lv = 1 + P + (cocoop ? 0 : c->n_ctx);  // CoCoOp's image encoder is the vanilla ViT (trainers/cocoop.py:38, clip/model.py:443-496)
    ARG_CHECK(Lv <= 4096 && c->ctx_len <= 4096, "create: sequence length %d/%d exceeds the attention limit (4096)", Lv, c->ctx_len);

    mudpt_model* m = new mudpt_model();
    m->cfg = *c;
    // MUDPT_F32 (the parity mode): the kernels' operand type is fp16.  Text tower: every forward GEMM operand a 22-bit
    // (hi, lo) pair and the attention forward in fp32 -- each of its rounding sites alone costs 2.5e-3 on the logits at logit scale 100.
    // Vision tower: every forward GEMM operand (and the pixels) hi + an e4m3 remainder contracted on the fp8 matrix pipe, attention in
    // fp16.  Knobs vis_lo / vis_exact_attn / vis_sites / txt_sites select the other points of the ablation (vis_lo = 1, vis_exact_attn = 1:
    // round 3's "exact" mode).  MUDPT_F16 splits the text tower's GEMM operands only (the round-2 mode, kept as it was).
    m->exact = c->dtype == MUDPT_F32;
    m->dtype = m->exact ? (int)MUDPT_F16 : c->dtype;
    m->vis.may_split = m->exact;
    m->txt.may_split = m->dtype == MUDPT_F16;
    m->vis.split = m->exact ? LO_F8 : LO_NONE;
    m->txt.split = m->txt.may_split ? LO_F16 : LO_NONE;
    m->txt.exact_attn = m->exact;
    // the gradient of the residual stream in T: bf16 mode, and the parity mode (its bound is on the LOGITS; its backward is the fp16 mode's
    // with fp16 activation gradients, as the reference's own fp16 model has them: gradient errors +30 %, -0.9 ms per step; knob lp_grad = 0)
    m->lp_grad = (c->dtype == MUDPT_BF16 || c->dtype == MUDPT_F32);
    m->lp_upd = (c->dtype == MUDPT_BF16);
    m->gelu_q8 = (c->dtype == MUDPT_BF16);
    m->cocoop = cocoop;
    m->ct = c->n_cls;
    if (cocoop) m->cfg.depth = 1;  // no deep prompts
    const int dv = c->v_width, dt = c->t_width, e = c->embed_dim, n = c->n_ctx, D1 = m->cfg.depth - 1, B = c->max_batch, C = c->n_cls;
    const int TS = cocoop ? B * C : C;  // text sequences per step: one per (image, class) pair in CoCoOp (trainers/cocoop.py:187-194)
    auto fail = [&](int code) { mudpt_destroy(m); return code; };
    if (int r = alloc_tower_weights(m, m->vis, dv, c->v_layers, c->v_heads, false, cocoop ? Lv : Lv - n, m->exact)) return fail(r);  // e4m3 weight copies: vision tower of the parity mode
    if (int r = alloc_tower_acts(m, m->vis, Lv, B)) return fail(r);
    // the text tower's activations are sized by mudpt_set_class_prompts: its trimmed length (max(eot) + 1 of ctx_len positions) and, for
    // CoCoOp, the number of images whose B * C prompts fit the memory budget at once are only known there
    if (int r = alloc_tower_weights(m, m->txt, dt, c->t_layers, c->t_heads, true, 1, false)) return fail(r);
    m->txt.L = c->ctx_len; m->txt.Lp = attn_padded_len(c->ctx_len);
    auto body = [&]() -> int {
        const int K0 = (3 * c->patch * c->patch + 63) / 64 * 64;  // conv-as-GEMM K, zero-padded to the GEMM's granularity (ViT-L/14: 588 -> 640)
        ALLOC(m->conv_w, (size_t)dv * K0 * 2);
        if (m->exact) { ALLOC(m->conv_w8, (size_t)dv * K0 * 2); ALLOC(m->patches_lo, (size_t)B * P * K0 * 2); }  // split pixels (vision tower's site 4)
        ALLOC(m->cls, dv * 4); ALLOC(m->vpos, (size_t)(1 + P) * dv * 4);
        ALLOC(m->ln_pre_g, dv * 4); ALLOC(m->ln_pre_b, dv * 4); ALLOC(m->ln_post_g, dv * 4); ALLOC(m->ln_post_b, dv * 4);
        ALLOC(m->vproj, (size_t)dv * e * 4);
        ALLOC(m->patches, (size_t)B * P * K0 * 2);
        ALLOC(m->xpre, (size_t)B * Lv * dv * 4); ALLOC(m->pre_mean, (size_t)B * Lv * 4); ALLOC(m->pre_rstd, (size_t)B * Lv * 4);
        ALLOC(m->f_ln, (size_t)B * dv * 4); ALLOC(m->post_mean, B * 4); ALLOC(m->post_rstd, B * 4); ALLOC(m->df_ln, (size_t)B * dv * 4);
        ALLOC(m->cls_rows, B * 4); ALLOC(m->vprompt_rows, (size_t)B * n * 4);
        ALLOC(m->tpos, (size_t)c->ctx_len * dt * 4); ALLOC(m->ln_fin_g, dt * 4); ALLOC(m->ln_fin_b, dt * 4);
        ALLOC(m->tproj, (size_t)dt * e * 4);
        ALLOC(m->emb_pos, (size_t)C * c->ctx_len * dt * 4); ALLOC(m->eot_rows, TS * 4); ALLOC(m->eot_local, TS * 4); ALLOC(m->class_perm, C * 4);
        ALLOC(m->txt_sorted, (size_t)C * e * 4); ALLOC(m->dtxt_sorted, (size_t)C * e * 4);
        ALLOC(m->t_ln, (size_t)TS * dt * 4); ALLOC(m->fin_mean, TS * 4); ALLOC(m->fin_rstd, TS * 4); ALLOC(m->dt_ln, (size_t)TS * dt * 4);
        const size_t dn = (size_t)(D1 > 0 ? D1 : 1) * n;
        ALLOC(m->shared, (size_t)n * dv * 4); ALLOC(m->t2v, dn * dv * 4); ALLOC(m->v2t, dn * e * 4);
        ALLOC(m->vis_deep, dn * dv * 4); ALLOC(m->txt_deep, dn * dt * 4);
        ALLOC(m->vsplice, (size_t)B * dn * dv * 4);
        ALLOC(m->d_vis_deep, dn * dv * 4); ALLOC(m->d_txt_deep, dn * dt * 4); ALLOC(m->d_vprompt0, (size_t)n * dv * 4);
        ALLOC(m->img_f, (size_t)B * e * 4); ALLOC(m->txt_f, (size_t)TS * e * 4); ALLOC(m->img_n, (size_t)B * e * 4); ALLOC(m->txt_n, (size_t)TS * e * 4);
        ALLOC(m->img_inv, B * 4); ALLOC(m->txt_inv, TS * 4);
        ALLOC(m->logits, (size_t)B * C * 4); ALLOC(m->dlogits, (size_t)B * C * 4); ALLOC(m->row_loss, B * 4);
        ALLOC(m->dimg, (size_t)B * e * 4); ALLOC(m->dtxt, (size_t)TS * e * 4); ALLOC(m->loss, 16);
        if (cocoop) {
            m->hid = e / 16;
            ALLOC(m->mn_hid, (size_t)B * m->hid * 4); ALLOC(m->mn_dhid, (size_t)B * m->hid * 4);
            ALLOC(m->mn_bias, (size_t)B * dt * 4); ALLOC(m->mn_dbias, (size_t)B * dt * 4);
        }
        ALLOC(m->gemm_scratch, mudpt_model::kScratchElems * 4); ALLOC(m->gemm_scratch2, mudpt_model::kScratchElems * 4);
        HIP_TRY(hipStreamCreateWithFlags(&m->s2, hipStreamNonBlocking));
        for (hipEvent_t* e : {&m->ev_fork, &m->ev_join, &m->ev_fork_b, &m->ev_join_b}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
        // row index tables
        std::vector<int> cr(B), pr((size_t)B * n);
        for (int b = 0; b < B; ++b) {
            cr[b] = b * Lv;
            for (int i = 0; i < n; ++i) pr[(size_t)b * n + i] = b * Lv + (Lv - n) + i;
        }
        HIP_TRY(hipMemcpy(m->cls_rows, cr.data(), cr.size() * 4, hipMemcpyHostToDevice));
        m->vis.tail_rows = m->cls_rows;
        m->vis.head_rows = cocoop ? nullptr : m->vprompt_rows;
        ALLOC(m->tprompt_rows, (size_t)TS * n * 4);
        m->txt.head_rows = m->tprompt_rows;  // ctx rows of every prompt; this table and the next are filled by mudpt_set_class_prompts
        m->txt.tail_rows = m->eot_rows;
        HIP_TRY(hipMemcpy(m->vprompt_rows, pr.data(), pr.size() * 4, hipMemcpyHostToDevice));
        return MUDPT_OK;
    };
    if (int r = body()) return fail(r);

    // flat bucket layout, reference order/shapes: trainers/mudpt.py:71-81, clip/model.py:512-519
    const size_t shapes[10] = {(size_t)n * dt, (size_t)D1 * n * dt, (size_t)dv * dt, (size_t)dv, (size_t)dv * dt, (size_t)dv,
                               (size_t)n * dv, (size_t)D1 * n * dv, (size_t)e * dv, (size_t)e};
    // CoCoOp: ctx [n, dt], meta_net.linear1 [e/16, e] + [e/16], meta_net.linear2 [dt, e/16] + [dt]  (trainers/cocoop.py:96-107)
    const size_t cshapes[5] = {(size_t)n * dt, (size_t)(e / 16) * e, (size_t)(e / 16), (size_t)dt * (e / 16), (size_t)dt};
    m->nparams = cocoop ? 5 : 10;
    size_t o = 0;
    for (int i = 0; i < m->nparams; ++i) { m->off[i] = o; m->numel[i] = cocoop ? cshapes[i] : shapes[i]; o += m->numel[i]; }
    m->total = o;
    if (int r = dev_alloc(m, (void**)&m->momentum, o * 4)) return fail(r);

    // frozen weights the path needs before it may run
    for (const char* k : {"visual.conv1.weight", "visual.class_embedding", "visual.positional_embedding", "visual.ln_pre.weight",
                          "visual.ln_pre.bias", "visual.ln_post.weight", "visual.ln_post.bias", "visual.proj", "positional_embedding",
                          "ln_final.weight", "ln_final.bias", "text_projection", "logit_scale"})
        m->missing.push_back(k);
    expect_block_keys(m, "visual.transformer", c->v_layers);
    expect_block_keys(m, "transformer", c->t_layers);
    *out = m;
    return MUDPT_OK;
}

extern "C" int mudpt_destroy(mudpt_model* m) {
    if (!m) return MUDPT_OK;
    for (void* p : m->allocs) (void)hipFree(p);
    for (Tower* t : {&m->vis, &m->txt})
        for (void* p : t->act_allocs) (void)hipFree(p);
    for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : {m->ev_fork, m->ev_join, m->ev_fork_b, m->ev_join_b})
        if (e) (void)hipEventDestroy(e);
    if (m->s2) (void)hipStreamDestroy(m->s2);
    delete m;
    return MUDPT_OK;
}

// ---- weight ingestion -----------------------------------------------------------------------------------
static int upload_f32(float* dst, const float* src, size_t n) {
    HIP_TRY(hipMemcpy(dst, src, n * 4, hipMemcpyHostToDevice));
    return MUDPT_OK;
}
// W [rows, cols] fp32 host -> T device, plain and (optionally) transposed copies
static int upload_lp(int dtype, void* dst, void* dst_t, const float* src, size_t rows, size_t cols) {
    std::vector<uint16_t> tmp(rows * cols);
    auto cv = [&](float f) { return dtype == DT_BF16 ? f32_to_bf16(f) : f32_to_f16(f); };
    for (size_t i = 0; i < rows * cols; ++i) tmp[i] = cv(src[i]);
    HIP_TRY(hipMemcpy(dst, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    if (dst_t) {
        for (size_t r = 0; r < rows; ++r)
            for (size_t c = 0; c < cols; ++c) tmp[c * rows + r] = cv(src[r * cols + c]);
        HIP_TRY(hipMemcpy(dst_t, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    }
    return MUDPT_OK;
}

// W [rows, cols] fp32 host -> e4m3 device copy for the second pass of a LO_F8 split GEMM: rows of 2 * cols BYTES (the T copy's row length, so
// that byte offsets into W and W8 agree), the first cols of them W 2^shift in e4m3; *scale_e8m0 = the block scale 2^-shift for the MFMA
static int upload_e4m3(void* dst, int* scale_e8m0, const float* src, size_t rows, size_t cols) {
    float mx = 0.f;
    for (size_t i = 0; i < rows * cols; ++i) mx = std::max(mx, std::fabs(src[i]));
    int shift = 0;
    if (mx > 0.f && std::isfinite(mx)) {
        shift = (int)std::floor(std::log2(448.f / mx));
        while (std::ldexp(mx, shift) > 448.f) --shift;
        shift = std::max(-100, std::min(100, shift));
    }
    std::vector<uint8_t> tmp(rows * 2 * cols, 0);
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c) tmp[r * 2 * cols + c] = f32_to_e4m3(std::ldexp(src[r * cols + c], shift));
    HIP_TRY(hipMemcpy(dst, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    *scale_e8m0 = 127 - shift;
    return MUDPT_OK;
}

static int set_block_weight(mudpt_model* m, Tower& t, int layer, const std::string& name, const float* data, size_t numel) {
    ARG_CHECK(layer >= 0 && layer < t.layers, "set_weight: layer %d out of range", layer);
    BlockW& w = t.w[layer];
    const size_t d = t.d;
    struct F32 { const char* n; float* p; size_t sz; };
    const F32 f32s[] = {{"ln_1.weight", w.ln1_g, d}, {"ln_1.bias", w.ln1_b, d}, {"ln_2.weight", w.ln2_g, d}, {"ln_2.bias", w.ln2_b, d},
                        {"attn.in_proj_bias", w.b_in, 3 * d}, {"attn.out_proj.bias", w.b_out, d}, {"mlp.c_fc.bias", w.b_fc, 4 * d},
                        {"mlp.c_proj.bias", w.b_proj, d}};
    for (const F32& f : f32s)
        if (name == f.n) {
            ARG_CHECK(numel == f.sz, "set_weight: %s expects %zu elements, got %zu", f.n, f.sz, numel);
            return upload_f32(f.p, data, numel);
        }
    struct LP { const char* n; void* p; void* pt; void* p8; int* s8; size_t rows, cols; };
    const LP lps[] = {{"attn.in_proj_weight", w.w_in, w.w_in_t, w.w_in8, &w.s_in8, 3 * d, d}, {"attn.out_proj.weight", w.w_out, w.w_out_t, w.w_out8, &w.s_out8, d, d},
                      {"mlp.c_fc.weight", w.w_fc, w.w_fc_t, w.w_fc8, &w.s_fc8, 4 * d, d}, {"mlp.c_proj.weight", w.w_proj, w.w_proj_t, w.w_proj8, &w.s_proj8, d, 4 * d}};
    for (const LP& l : lps)
        if (name == l.n) {
            ARG_CHECK(numel == l.rows * l.cols, "set_weight: %s expects %zu elements, got %zu", l.n, l.rows * l.cols, numel);
            if (l.p8) { if (int rc = upload_e4m3(l.p8, l.s8, data, l.rows, l.cols)) return rc; }
            return upload_lp(m->dtype, l.p, l.pt, data, l.rows, l.cols);
        }
    set_error("set_weight: unknown block tensor '%s'", name.c_str());
    return MUDPT_ERR_ARG;
}

extern "C" int mudpt_set_weight(mudpt_model* m, const char* key, const float* data, size_t numel) {
    ARG_CHECK(m && key && data, "set_weight: null argument");
    const mudpt_config& c = m->cfg;
    const std::string k(key);
    const size_t dv = c.v_width, dt = c.t_width, e = c.embed_dim;
    const size_t P = (size_t)(c.image_size / c.patch) * (c.image_size / c.patch);
    int rc = MUDPT_OK;
    auto blk = [&](const char* prefix, Tower& t) -> int {
        const std::string rest = k.substr(strlen(prefix));
        const size_t dot = rest.find('.');
        ARG_CHECK(dot != std::string::npos, "set_weight: malformed key %s", key);
        return set_block_weight(m, t, atoi(rest.substr(0, dot).c_str()), rest.substr(dot + 1), data, numel);
    };
#define EXPECT(n) ARG_CHECK(numel == (size_t)(n), "set_weight: %s expects %zu elements, got %zu", key, (size_t)(n), numel)
    if (k.rfind("visual.transformer.resblocks.", 0) == 0) rc = blk("visual.transformer.resblocks.", m->vis);
    else if (k.rfind("transformer.resblocks.", 0) == 0) rc = blk("transformer.resblocks.", m->txt);
    else if (k == "visual.conv1.weight") {
        EXPECT(dv * 3 * c.patch * c.patch);
        const size_t k0 = (size_t)3 * c.patch * c.patch, k0p = (k0 + 63) / 64 * 64;
        std::vector<float> padded(dv * k0p, 0.f);  // rows zero-padded like the im2col rows
        for (size_t r = 0; r < dv; ++r) memcpy(&padded[r * k0p], data + r * k0, k0 * 4);
        rc = upload_lp(m->dtype, m->conv_w, nullptr, padded.data(), dv, k0p);
        if (!rc && m->conv_w8) rc = upload_e4m3(m->conv_w8, &m->conv_s8, padded.data(), dv, k0p);
    }
    else if (k == "visual.class_embedding") { EXPECT(dv); rc = upload_f32(m->cls, data, numel); }
    else if (k == "visual.positional_embedding") { EXPECT((1 + P) * dv); rc = upload_f32(m->vpos, data, numel); }
    else if (k == "visual.ln_pre.weight") { EXPECT(dv); rc = upload_f32(m->ln_pre_g, data, numel); }
    else if (k == "visual.ln_pre.bias") { EXPECT(dv); rc = upload_f32(m->ln_pre_b, data, numel); }
    else if (k == "visual.ln_post.weight") { EXPECT(dv); rc = upload_f32(m->ln_post_g, data, numel); }
    else if (k == "visual.ln_post.bias") { EXPECT(dv); rc = upload_f32(m->ln_post_b, data, numel); }
    else if (k == "visual.proj") { EXPECT(dv * e); rc = upload_f32(m->vproj, data, numel); }
    else if (k == "positional_embedding") { EXPECT((size_t)c.ctx_len * dt); rc = upload_f32(m->tpos, data, numel); m->prompts_set = false; }
    else if (k == "ln_final.weight") { EXPECT(dt); rc = upload_f32(m->ln_fin_g, data, numel); }
    else if (k == "ln_final.bias") { EXPECT(dt); rc = upload_f32(m->ln_fin_b, data, numel); }
    else if (k == "text_projection") { EXPECT(dt * e); rc = upload_f32(m->tproj, data, numel); }
    else if (k == "logit_scale") { EXPECT(1); m->scale = std::exp(data[0]); }  // trainers/mudpt.py:181
    else if (k == "token_embedding.weight" || k == "input_resolution" || k == "context_length" || k == "vocab_size") return MUDPT_OK;
    else { set_error("set_weight: unknown key '%s'", key); return MUDPT_ERR_ARG; }
#undef EXPECT
    if (rc) return rc;
    m->any_weight_set = true;
    for (size_t i = 0; i < m->missing.size(); ++i)
        if (m->missing[i] == k) { m->missing.erase(m->missing.begin() + i); break; }
    return MUDPT_OK;
}

extern "C" int mudpt_set_class_prompts(mudpt_model* m, const float* emb, const int32_t* eot) {
    ARG_CHECK(m && emb && eot, "set_class_prompts: null argument");
    for (const std::string& k : m->missing)
        if (k == "positional_embedding") { set_error("set_class_prompts: set 'positional_embedding' first"); return MUDPT_ERR_STATE; }
    const mudpt_config& c = m->cfg;
    // emb / eot describe ALL n_cls classes on every rank; a class-sharded handle (mudpt_set_class_shard) keeps its own [c0, c0 + C) only
    const size_t C = (size_t)m->ct, c0 = (size_t)m->c0, L = c.ctx_len, d = c.t_width;
    for (int cc = 0; cc < c.n_cls; ++cc) ARG_CHECK(eot[cc] >= 0 && eot[cc] < (int)L, "set_class_prompts: eot index %d out of range", eot[cc]);
    emb += c0 * L * d;
    eot += c0;
    // The text tower is causal (clip/model.py:407-413 build_attention_mask) and only the EOT row of each prompt is used
    // (trainers/mudpt.py:154): positions behind the last EOT of the class set influence neither a used output nor a gradient,
    // so the tower runs on the first Le = max(eot) + 1 positions of every prompt ("a photo of a <name>." ends at position 7-9
    // of 77).  Row-wise operators and causal attention make the kept rows bit-identical to the full-length run;
    // mudpt_model_set("txt_trim", 0) keeps all ctx_len positions (A/B runs, tests).
    int max_eot = 0;
    for (size_t cc = 0; cc < C; ++cc) max_eot = std::max(max_eot, (int)eot[cc]);
    const size_t Le = m->txt_trim ? (size_t)std::max(max_eot + 1, c.n_ctx + 2) : L;
    // Sequences per text-tower pass.  MuDPT: the C class prompts.  CoCoOp: every image has its own C prompts (trainers/cocoop.py:187-194
    // loops over the images, C sequences at a time); here a CHUNK of images goes through the tower at once -- as many as fit a
    // memory budget (activations for the backward are ~150 KB per token at width 512) and the kernels' 32-bit offsets -- and
    // cocoop_forward / cocoop_forward_backward loop over the chunks.  The reference's own config (train batch 1, test batch 100, up to
    // 1000 classes) therefore needs C * Le tokens of activations at least, never max_batch * C * ctx_len.
    size_t chunk = 1;
    if (m->cocoop) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        for (void* p : m->txt.act_allocs) (void)hipFree(p);  // a previous sizing does not count against the budget
        m->txt.act_allocs.clear();
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        const size_t per_row = tower_bytes_per_row(m->txt), budget = (size_t)((double)free_b * 0.6);
        size_t rows_max = budget / per_row;
        // the widest row of a pass (QuickGELU(u) and its low half: 2 x 8 d bytes) stays inside the kernels' 32-bit byte offsets
        const size_t rows_cap = (size_t)0x7fffffff / ((size_t)16 * d) - 1;
        if (rows_max > rows_cap) rows_max = rows_cap;
        chunk = rows_max / (C * Le);
        if (chunk > (size_t)c.max_batch) chunk = (size_t)c.max_batch;
        if (m->cocoop_chunk > 0 && chunk > (size_t)m->cocoop_chunk) chunk = (size_t)m->cocoop_chunk;
        if (chunk < 1) {
            set_error("set_class_prompts: one image's %zu class prompts x %zu positions (%zu tokens, %.1f GB of text-tower activations) "
                      "exceed the budget of %.1f GB / %zu tokens per pass", C, Le, C * Le, (double)(C * Le * per_row) / 1e9, (double)budget / 1e9, rows_cap);
            return MUDPT_ERR_ARG;
        }
    }
    m->txt_chunk = (int)chunk;
    if (int rc = alloc_tower_acts(m, m->txt, (int)Le, (int)(m->cocoop ? chunk * C : C))) return rc;
    // Length buckets (MuDPT, many classes): "a photo of a <name>." ends at position 7-9 for most ImageNet names and at 19 for a few; the
    // trim above runs EVERY prompt to the longest.  Sorting the prompts by length and cutting the sorted list into <= txt_buckets groups,
    // each run to its own longest member, removes most of the padding (C = 1000 synthetic names: 19 000 -> ~11 000 rows).  Sequences are
    // independent in every kernel of the tower, so the kept rows are bit-identical to the single-bucket run (tests).  A bucket costs a
    // handful of extra launches per block (attention, splice, reductions: ~1 000 rows' worth of time), which the cut search charges.
    Tower& X = m->txt;
    X.segs.clear();
    std::vector<int> order(C);  // packed position -> local class
    for (size_t cc = 0; cc < C; ++cc) order[cc] = (int)cc;
    auto len_of = [&](int cc) { return std::max(eot[cc] + 1, c.n_ctx + 2); };
    if (!m->cocoop && m->txt_trim && m->txt_buckets > 1 && C * Le >= 2048) {
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len_of(x) < len_of(y); });
        std::vector<int> dl, cnt;  // distinct lengths ascending, sequences per length
        for (int cc : order) {
            if (dl.empty() || dl.back() != len_of(cc)) { dl.push_back(len_of(cc)); cnt.push_back(0); }
            ++cnt.back();
        }
        const int nd = (int)dl.size(), K = std::min(m->txt_buckets, nd);
        const long PEN = m->txt_bucket_cost, INF = 1L << 60;
        std::vector<long> pre(nd + 1, 0);
        for (int j = 0; j < nd; ++j) pre[j + 1] = pre[j] + cnt[j];
        // best[b][j]: rows (+ penalties) of covering lengths 0..j-1 with b buckets, the last one ending at length j-1
        std::vector<std::vector<long>> best(K + 1, std::vector<long>(nd + 1, INF));
        std::vector<std::vector<int>> from(K + 1, std::vector<int>(nd + 1, 0));
        best[0][0] = 0;
        for (int bk = 1; bk <= K; ++bk)
            for (int j = 1; j <= nd; ++j)
                for (int i0 = bk - 1; i0 < j; ++i0) {
                    if (best[bk - 1][i0] >= INF) continue;
                    const long v = best[bk - 1][i0] + (pre[j] - pre[i0]) * dl[j - 1] + PEN;
                    if (v < best[bk][j]) { best[bk][j] = v; from[bk][j] = i0; }
                }
        int kb = 1;
        for (int bk = 2; bk <= K; ++bk) if (best[bk][nd] < best[kb][nd]) kb = bk;
        if (kb > 1) {
            std::vector<int> cuts;  // bucket boundaries in distinct-length indices
            for (int bk = kb, j = nd; bk >= 1; --bk) { cuts.push_back(j); j = from[bk][j]; }
            std::reverse(cuts.begin(), cuts.end());
            int j0 = 0, row0 = 0;
            size_t lse0 = 0;
            for (int j1 : cuts) {
                Tower::Seg g; g.row0 = row0; g.seq0 = (int)pre[j0]; g.nseq = (int)(pre[j1] - pre[j0]); g.L = dl[j1 - 1]; g.lse0 = lse0;
                X.segs.push_back(g);
                row0 += g.nseq * g.L;
                lse0 += (size_t)g.nseq * X.heads * attn_padded_len(g.L);
                j0 = j1;
            }
            X.rows = row0;
        } else {
            for (size_t cc = 0; cc < C; ++cc) order[cc] = (int)cc;  // one bucket: keep the caller's order
        }
    }
    std::vector<float> pos(L * d);
    HIP_TRY(hipMemcpy(pos.data(), m->tpos, pos.size() * 4, hipMemcpyDeviceToHost));
    const size_t reps = chunk;  // CoCoOp: sequence i * C + c for every image i of a chunk (the tables are chunk-local, reused per chunk)
    const std::vector<Tower::Seg> segs = tower_segs(X, (int)C);
    const size_t packed_rows = X.segs.empty() ? C * Le : (size_t)X.rows;
    std::vector<float> ep(packed_rows * d);
    std::vector<int> rows(C * reps), rows_local(C * reps), tr(C * reps * c.n_ctx), perm(C);
    for (const Tower::Seg& g : segs)
        for (int j = 0; j < g.nseq; ++j) {
            const int sq = g.seq0 + j, cc = order[sq];
            const size_t r0 = (size_t)g.row0 + (size_t)j * g.L;
            perm[sq] = cc;
            for (size_t i = 0; i < reps; ++i) {  // reps > 1 (CoCoOp) only with one bucket: image i's prompts follow image i - 1's
                rows[i * C + sq] = (int)(i * C * Le + r0) + eot[cc];
                rows_local[i * C + sq] = (int)(i * C * Le + r0 - g.row0) + eot[cc];
                for (int k = 0; k < c.n_ctx; ++k) tr[(i * C + sq) * c.n_ctx + k] = (int)(i * C * Le + r0) + 1 + k;  // ctx rows 1..n (trainers/mudpt.py:97-115)
            }
            for (size_t i = 0; i < (size_t)g.L * d; ++i) ep[r0 * d + i] = emb[(size_t)cc * L * d + i] + pos[i];  // trainers/mudpt.py:143
        }
    HIP_TRY(hipMemcpy(m->emb_pos, ep.data(), ep.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->eot_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->eot_local, rows_local.data(), rows_local.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->tprompt_rows, tr.data(), tr.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->class_perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
    X.tail_local = m->eot_local;
    m->prompts_set = true;
    m->text_valid = false;
    return MUDPT_OK;
}

extern "C" int mudpt_text_layout(const mudpt_model* m, int32_t* rows, int32_t* buckets, int32_t* max_len) {
    ARG_CHECK(m && m->prompts_set, "text_layout: call mudpt_set_class_prompts first");
    const int nseq = m->cocoop ? m->txt_chunk * m->cfg.n_cls : m->ct;
    if (rows) *rows = tower_rows(m->txt, nseq);
    if (buckets) *buckets = m->txt.segs.empty() ? 1 : (int)m->txt.segs.size();
    if (max_len) *max_len = m->txt.L;
    return MUDPT_OK;
}

extern "C" int mudpt_param_count(const mudpt_model* m) { return m ? m->nparams : 0; }
extern "C" size_t mudpt_param_numel(const mudpt_model* m) { return m ? m->total : 0; }
extern "C" int mudpt_param_info(const mudpt_model* m, int i, const char** name, size_t* offset, size_t* numel, int32_t* ndim, int64_t shape[3]) {
    ARG_CHECK(m && i >= 0 && i < m->nparams, "param_info: bad index %d", i);
    const mudpt_config& c = m->cfg;
    const int64_t n = c.n_ctx, D1 = c.depth - 1, dt = c.t_width, dv = c.v_width, e = c.embed_dim;
    if (m->cocoop) {
        const int64_t hd = e / 16;
        const int64_t cshp[5][3] = {{n, dt, 0}, {hd, e, 0}, {hd, 0, 0}, {dt, hd, 0}, {dt, 0, 0}};
        const int cnd[5] = {2, 2, 1, 2, 1};
        if (name) *name = kCocoopNames[i];
        if (offset) *offset = m->off[i];
        if (numel) *numel = m->numel[i];
        if (ndim) *ndim = cnd[i];
        if (shape) for (int k = 0; k < 3; ++k) shape[k] = cshp[i][k];
        return MUDPT_OK;
    }
    const int64_t shp[10][3] = {{n, dt, 0}, {D1, n, dt}, {dv, dt, 0}, {dv, 0, 0}, {dv, dt, 0}, {dv, 0, 0}, {n, dv, 0}, {D1, n, dv}, {e, dv, 0}, {e, 0, 0}};
    const int nd[10] = {2, 3, 2, 1, 2, 1, 2, 3, 2, 1};
    if (name) *name = kParamNames[i];
    if (offset) *offset = m->off[i];
    if (numel) *numel = m->numel[i];
    if (ndim) *ndim = nd[i];
    if (shape) for (int k = 0; k < 3; ++k) shape[k] = shp[i][k];
    return MUDPT_OK;
}
extern "C" int mudpt_bind_params(mudpt_model* m, float* p, float* g) {
    ARG_CHECK(m && p, "bind_params: null argument");
    ARG_CHECK((uintptr_t)p % 16 == 0 && (uintptr_t)g % 16 == 0, "bind_params: buckets must be 16-byte aligned");
    m->params = p;
    m->grads = g;
    return MUDPT_OK;
}

// ---- the path ---------------------------------------------------------------------------------------------
#define TRY(expr)                      \
    do {                               \
        if (int _e = (expr)) return _e; \
    } while (0)

static int ready(mudpt_model* m, int B, bool need_grads) {
    ARG_CHECK(m, "null model");
    if (!m->missing.empty()) {
        set_error("model not ready: %zu frozen weights unset (first: %s)", m->missing.size(), m->missing[0].c_str());
        return MUDPT_ERR_STATE;
    }
    if (!m->prompts_set) { set_error("model not ready: call mudpt_set_class_prompts"); return MUDPT_ERR_STATE; }
    if (!m->params) { set_error("model not ready: call mudpt_bind_params"); return MUDPT_ERR_STATE; }
    if (need_grads && !m->grads) { set_error("model not ready: no gradient bucket bound"); return MUDPT_ERR_STATE; }
    ARG_CHECK(B > 0 && B <= m->cfg.max_batch, "batch %d outside 1..max_batch=%d", B, m->cfg.max_batch);
    return MUDPT_OK;
}

// Split operands: the form of the low half at GEMM site `site` of tower t whose contraction length is K (common.h LoMode) -- what the
// producing kernel writes and what the consuming GEMM's second pass reads.  The e4m3 pass needs K % 128 == 0 (one matrix instruction
// contracts 128 k): narrower sites of a LO_F8 tower (the 192-wide test shapes) fall back to the fp16 pair.
enum Site : int { SITE_QKV = 0, SITE_OUT = 1, SITE_FC = 2, SITE_PROJ = 3, SITE_PATCH = 4 };
static int site_mode(const Tower& t, int site, int K) {
    if (t.split == LO_NONE || !((t.sites >> site) & 1)) return LO_NONE;
    return (t.split == LO_F8 && K % 128 == 0) ? LO_F8 : LO_F16;
}
// ... and the second-pass fields of the consuming GEMM: the low half of A, and for the e4m3 form the e4m3 weights (same row offset as B) + scale
static void split_operand(GemmArgs& g, int mode, const void* A_lo, const void* B8, int b8_scale) {
    if (mode == LO_NONE) return;
    g.A_lo = A_lo; g.lo_mode = mode;
    if (mode == LO_F8) { g.B8 = B8; g.b8_scale = b8_scale; }
}

// Forward of the last block after its attention, on the one used row of every sequence (Tower::tail_rows): gathers the
// rows, then out_proj (+ residual in the small GEMM's epilogue), ln_2, c_fc + QuickGELU, c_proj (+ residual) -> t.xout_sel.
static int block_fwd_tail(mudpt_model* m, Tower& t, int nseq, hipStream_t s, bool attn_sel_ready = false) {
    const int i = t.layers - 1, S = nseq, d = t.d, dt = m->dtype;
    BlockW& w = t.w[i];
    BlockAct& a = t.a[i];
    const size_t esz = 2;
    const int m_out = site_mode(t, SITE_OUT, d), m_fc = site_mode(t, SITE_FC, d), m_proj = site_mode(t, SITE_PROJ, 4 * d);
    if (!attn_sel_ready) {
        TRY(launch_gather_rows(a.attn, (size_t)d * esz, t.tail_rows, t.attn_sel, (size_t)d * esz, S, d * (int)esz, s));
        if (m_out != LO_NONE) TRY(launch_gather_rows(t.attn_lo, (size_t)d * esz, t.tail_rows, t.attn_sel_lo, (size_t)d * esz, S, d * (int)esz, s));
    }
    TRY(launch_gather_rows(a.x_in, (size_t)d * 4, t.tail_rows, t.xin_sel, (size_t)d * 4, S, d * 4, s));
    GemmArgs o; o.A = t.attn_sel; o.lda = d; o.B = w.w_out; o.ldb = d; o.M = S; o.N = d; o.K = d; o.bias = w.b_out;
    split_operand(o, m_out, t.attn_sel_lo, w.w_out8, w.s_out8);
    o.out0 = t.xmid_sel; o.ldo0 = d; o.aux = t.xin_sel; o.ldaux = d;
    TRY(gemm_call(m, EPI_RESIDUAL, o, s));
    LnFwdArgs l2; l2.x = t.xmid_sel; l2.ldx = d; l2.gamma = w.ln2_g; l2.beta = w.ln2_b; l2.out = t.h_sel; l2.ldo = d; l2.mean = a.mean2; l2.rstd = a.rstd2; l2.rows = S; l2.d = d;
    if (m_fc != LO_NONE) { l2.out_lo = t.h_sel_lo; l2.lo_mode = m_fc; }
    TRY(launch_ln_fwd(dt, l2, s));
    GemmArgs f; f.A = t.h_sel; f.lda = d; f.B = w.w_fc; f.ldb = d; f.M = S; f.N = 4 * d; f.K = d; f.bias = w.b_fc;
    split_operand(f, m_fc, t.h_sel_lo, w.w_fc8, w.s_fc8);
    f.out0 = t.u_sel; f.ldo0 = 4 * d; f.out1 = t.g_sel; f.ldo1 = 4 * d;
    f.gelu_q8 = m->gelu_q8 && t.split == LO_NONE;
    if (m_proj != LO_NONE) { f.out1_lo = t.g_sel_lo; f.out1_lo_mode = m_proj; }
    TRY(gemm_call(m, EPI_GELU, f, s));
    GemmArgs p; p.A = t.g_sel; p.lda = 4 * d; p.B = w.w_proj; p.ldb = 4 * d; p.M = S; p.N = d; p.K = 4 * d; p.bias = w.b_proj;
    split_operand(p, m_proj, t.g_sel_lo, w.w_proj8, w.s_proj8);
    p.out0 = t.xout_sel; p.ldo0 = d; p.aux = t.xmid_sel; p.ldaux = d;
    TRY(gemm_call(m, EPI_RESIDUAL, p, s));
    return MUDPT_OK;
}

// Block i of a tower.  The residual adds are NOT in the GEMM epilogues: out_proj / c_proj write their fp32 result
// (+ bias) to t.upd and the FOLLOWING LayerNorm kernel adds it to the stream while it reads it (the stream has to
// pass through that kernel anyway; a GEMM epilogue that loads the residual stalls behind its own stores, since
// vmcnt retires in order).  So LN1 of block i >= 1 computes x_in[i] = x_mid[i-1] + upd, with the deep-prompt rows
// spliced in (splice != null), and writes it for the backward; LN2 computes x_mid[i] = x_in[i] + upd.
static int block_fwd(mudpt_model* m, Tower& t, int i, int nseq, const float* splice, hipStream_t s) {
    const int M = tower_rows(t, nseq), d = t.d, dt = m->dtype, n = m->cfg.n_ctx;
    const std::vector<Tower::Seg> segs = tower_segs(t, nseq);
    // bf16 mode: the update stream (out_proj / c_proj results) is kept in T like the gradient stream -- half the store time
    // of those GEMMs and 2 bytes less per element in the LayerNorm that adds it.  The last block's c_proj stays fp32 (launch_add).
    const bool lp = m->lp_upd;
    BlockW& w = t.w[i];
    BlockAct& a = t.a[i];
    const size_t esz = 2;
    // split operands (Tower::split): the form of each site's low half (LO_NONE: the site runs on T alone)
    const int m_qkv = site_mode(t, SITE_QKV, d), m_out = site_mode(t, SITE_OUT, d), m_fc = site_mode(t, SITE_FC, d), m_proj = site_mode(t, SITE_PROJ, 4 * d);
    LnFwdArgs l1; l1.x = a.x_in; l1.ldx = d; l1.gamma = w.ln1_g; l1.beta = w.ln1_b; l1.out = t.h; l1.ldo = d; l1.mean = a.mean1; l1.rstd = a.rstd1; l1.rows = M; l1.d = d;
    if (m_qkv != LO_NONE) { l1.out_lo = t.h_lo; l1.lo_mode = m_qkv; }
    if (i > 0) {
        l1.x = t.a[i - 1].x_mid; l1.ldadd = d; l1.xout = a.x_in; l1.ldxout = d;
        if (lp) l1.add_lp = t.upd; else l1.add = t.upd;
        if (splice) { l1.ov_rows = splice; l1.ov_row0 = t.prompt_row0; l1.ov_n = n; l1.ov_L = t.L; }
    }
    if (splice && i > 0 && segs.size() > 1) {
        // the splice replaces rows by their position inside a sequence: one launch per length bucket
        for (const Tower::Seg& g : segs) {
            LnFwdArgs b = l1;
            const size_t r0 = (size_t)g.row0;
            b.x = l1.x + r0 * l1.ldx; b.xout = l1.xout + r0 * l1.ldxout;
            if (l1.add) b.add = l1.add + r0 * l1.ldadd;
            if (l1.add_lp) b.add_lp = (const char*)l1.add_lp + r0 * l1.ldadd * esz;
            b.out = (char*)l1.out + r0 * l1.ldo * esz;
            if (l1.out_lo) b.out_lo = (char*)l1.out_lo + r0 * l1.ldo * esz;
            b.mean = l1.mean + r0; b.rstd = l1.rstd + r0; b.rows = g.nseq * g.L; b.ov_L = g.L;
            TRY(ln_fwd_call(m, t, b, s));
        }
    } else {
        TRY(ln_fwd_call(m, t, l1, s));
    }
    if (i + 1 == t.layers && m->last_single && !t.exact_attn && t.tail_rows) {
        // Last block: only ONE query per sequence is ever used (CLS / EOT row).  K and V for every row (the k, v thirds of in_proj: rows
        // d .. 3d of its weight, written into the k, v thirds of the packed qkv buffer), q for the selected rows only, single-query attention
        // straight into the compact attn_sel the tail works on.  (With the fp32 attention forward the general kernel runs instead: the
        // single-query kernels take fp16 q, k, v.)
        GemmArgs kv; kv.A = t.h; kv.lda = d; kv.B = (const char*)w.w_in + (size_t)d * d * esz; kv.ldb = d; kv.M = M; kv.N = 2 * d; kv.K = d; kv.bias = w.b_in + d;
        split_operand(kv, m_qkv, t.h_lo, w.w_in8 ? (const char*)w.w_in8 + (size_t)d * d * esz : nullptr, w.s_in8);
        kv.out0 = (char*)a.qkv + (size_t)d * esz; kv.ldo0 = 3 * d;
        TRY(gemm_call(m, EPI_STORE, kv, s));
        TRY(launch_gather_rows(t.h, (size_t)d * esz, t.tail_rows, t.h_sel, (size_t)d * esz, nseq, d * (int)esz, s));
        if (m_qkv != LO_NONE) TRY(launch_gather_rows(t.h_lo, (size_t)d * esz, t.tail_rows, t.h_sel_lo, (size_t)d * esz, nseq, d * (int)esz, s));
        GemmArgs qs; qs.A = t.h_sel; qs.lda = d; qs.B = w.w_in; qs.ldb = d; qs.M = nseq; qs.N = d; qs.K = d; qs.bias = w.b_in;
        split_operand(qs, m_qkv, t.h_sel_lo, w.w_in8, w.s_in8);
        qs.out0 = t.q_sel; qs.ldo0 = d;
        TRY(gemm_call(m, EPI_STORE, qs, s));
        for (const Tower::Seg& g : segs) {
            AttnArgs at; at.qkv = (const char*)a.qkv + (size_t)g.row0 * 3 * d * esz; at.B = g.nseq; at.L = g.L; at.H = t.heads; at.causal = t.causal;
            at.sel_rows = segs.size() > 1 ? t.tail_local + g.seq0 : t.tail_rows;
            at.lo_mode = m_out;
            TRY(launch_attn_fwd_single(dt, at, (const char*)t.q_sel + (size_t)g.seq0 * d * esz, (char*)t.attn_sel + (size_t)g.seq0 * d * esz,
                                       m_out != LO_NONE ? (char*)t.attn_sel_lo + (size_t)g.seq0 * d * esz : nullptr, d, t.lse_sel + (size_t)g.seq0 * t.heads, s));
        }
        return block_fwd_tail(m, t, nseq, s, true);
    }
    GemmArgs q; q.A = t.h; q.lda = d; q.B = w.w_in; q.ldb = d; q.M = M; q.N = 3 * d; q.K = d; q.bias = w.b_in; q.out0 = a.qkv; q.ldo0 = 3 * d;
    split_operand(q, m_qkv, t.h_lo, w.w_in8, w.s_in8);
    if (t.exact_attn) q.out0 = t.qkv32;  // fp32 q | k | v for the fp32 attention forward, which leaves their fp16 copy in a.qkv for the backward
    TRY(gemm_call(m, t.exact_attn ? EPI_STORE_F32 : EPI_STORE, q, s));
    for (const Tower::Seg& g : segs) {
        AttnArgs at; at.qkv = (const char*)a.qkv + (size_t)g.row0 * 3 * d * esz; at.out = (char*)a.attn + (size_t)g.row0 * d * esz; at.lse = a.lse + g.lse0;
        at.B = g.nseq; at.L = g.L; at.H = t.heads; at.causal = t.causal;
        if (m_out != LO_NONE) { at.out_lo = (char*)t.attn_lo + (size_t)g.row0 * d * esz; at.lo_mode = m_out; }
        if (t.exact_attn) { at.qkv32 = t.qkv32 + (size_t)g.row0 * 3 * d; at.qkv_lp = (char*)a.qkv + (size_t)g.row0 * 3 * d * esz; }
        TRY(attn_call(m, t, at, false, s));
    }
    if (i + 1 == t.layers) return block_fwd_tail(m, t, nseq, s);
    GemmArgs o; o.A = a.attn; o.lda = d; o.B = w.w_out; o.ldb = d; o.M = M; o.N = d; o.K = d; o.bias = w.b_out; o.out0 = t.upd; o.ldo0 = d;
    split_operand(o, m_out, t.attn_lo, w.w_out8, w.s_out8);
    const bool fs = !t.causal && t.split == LO_NONE;  // forward split K: the vision tower's out_proj / c_proj at tiny batches only (gemm_call), never with split operands
    TRY(gemm_call(m, lp ? EPI_STORE : EPI_STORE_F32, o, s, false, fs));
    LnFwdArgs l2; l2.x = a.x_in; l2.ldx = d; if (lp) l2.add_lp = t.upd; else l2.add = t.upd; l2.ldadd = d; l2.xout = a.x_mid; l2.ldxout = d; l2.gamma = w.ln2_g; l2.beta = w.ln2_b; l2.out = t.h; l2.ldo = d;
    if (m_fc != LO_NONE) { l2.out_lo = t.h_lo; l2.lo_mode = m_fc; }
    l2.mean = a.mean2; l2.rstd = a.rstd2; l2.rows = M; l2.d = d;
    TRY(ln_fwd_call(m, t, l2, s));
    GemmArgs f; f.A = t.h; f.lda = d; f.B = w.w_fc; f.ldb = d; f.M = M; f.N = 4 * d; f.K = d; f.bias = w.b_fc; f.out0 = a.u; f.ldo0 = 4 * d;
    split_operand(f, m_fc, t.h_lo, w.w_fc8, w.s_fc8);
    f.out1 = t.g; f.ldo1 = 4 * d;
    f.gelu_q8 = m->gelu_q8 && t.split == LO_NONE;  // a.u then holds byte codes of QuickGELU'(u) (rows of 4 d bytes)
    if (m_proj != LO_NONE) { f.out1_lo = t.g_lo; f.out1_lo_mode = m_proj; }
    TRY(gemm_call(m, EPI_GELU, f, s));
    GemmArgs p; p.A = t.g; p.lda = 4 * d; p.B = w.w_proj; p.ldb = 4 * d; p.M = M; p.N = d; p.K = 4 * d; p.bias = w.b_proj; p.out0 = t.upd; p.ldo0 = d;
    split_operand(p, m_proj, t.g_lo, w.w_proj8, w.s_proj8);
    TRY(gemm_call(m, lp ? EPI_STORE : EPI_STORE_F32, p, s, false, fs));
    return MUDPT_OK;
}

// Backward of the last block's tail (see Tower::tail_rows).  in: t.dsel / t.dsel_lp = gradient w.r.t. the selected rows of
// the tower output; out: t.dx / t.dx_lp = gradient w.r.t. the last block's input, all rows.
static int block_bwd_tail(mudpt_model* m, Tower& t, int nseq, hipStream_t s) {
    const int i = t.layers - 1, M = tower_rows(t, nseq), S = nseq, d = t.d, dt = m->dtype;
    const std::vector<Tower::Seg> segs = tower_segs(t, nseq);
    const size_t esz = 2;
    BlockW& w = t.w[i];
    BlockAct& a = t.a[i];
    GemmArgs g1; g1.A = t.dsel_lp; g1.lda = d; g1.B = w.w_proj_t; g1.ldb = d; g1.M = S; g1.N = 4 * d; g1.K = d; g1.out0 = t.g_sel; g1.ldo0 = 4 * d; g1.aux = t.u_sel; g1.ldaux = 4 * d;
    g1.gelu_q8 = m->gelu_q8 && t.split == LO_NONE;  // as the forward stored it (block_fwd_tail)
    TRY(gemm_call(m, EPI_GELU_BWD, g1, s, !t.causal));
    GemmArgs g2; g2.A = t.g_sel; g2.lda = 4 * d; g2.B = w.w_fc_t; g2.ldb = 4 * d; g2.M = S; g2.N = d; g2.K = 4 * d; g2.out0 = t.h_sel; g2.ldo0 = d;
    TRY(gemm_call(m, EPI_STORE, g2, s, !t.causal));
    LnBwdArgs b2; b2.dy = t.h_sel; b2.lddy = d; b2.x = t.xmid_sel; b2.ldx = d; b2.mean = a.mean2; b2.rstd = a.rstd2; b2.gamma = w.ln2_g; b2.lddres = d;
    if (m->lp_grad) b2.dres_lp = t.dsel_lp; else { b2.dres = t.dsel; b2.dx = t.dsel; }
    b2.lddx = d; b2.dx_lp = t.dsel_lp; b2.lddx_lp = d; b2.rows = S; b2.d = d;
    TRY(launch_ln_bwd(dt, b2, s));  // t.dsel(_lp) = gradient w.r.t. x_mid on the selected rows
    GemmArgs g3; g3.A = t.dsel_lp; g3.lda = d; g3.B = w.w_out_t; g3.ldb = d; g3.M = S; g3.N = d; g3.K = d; g3.out0 = t.dattn_sel; g3.ldo0 = d;
    TRY(gemm_call(m, EPI_STORE, g3, s, !t.causal));
    if (m->last_single && !t.exact_attn) {
        // single-query attention backward: dK, dV of every row (k, v thirds of t.dqkv) and dq of the one query per sequence
        for (const Tower::Seg& g : segs) {
            AttnArgs at; at.qkv = (const char*)a.qkv + (size_t)g.row0 * 3 * d * esz; at.dqkv = (char*)t.dqkv + (size_t)g.row0 * 3 * d * esz;
            at.sel_rows = segs.size() > 1 ? t.tail_local + g.seq0 : t.tail_rows;
            at.B = g.nseq; at.L = g.L; at.H = t.heads; at.causal = t.causal;
            TRY(launch_attn_bwd_single(dt, at, (const char*)t.q_sel + (size_t)g.seq0 * d * esz, (const char*)t.attn_sel + (size_t)g.seq0 * d * esz, d,
                                       (const char*)t.dattn_sel + (size_t)g.seq0 * d * esz, t.lse_sel + (size_t)g.seq0 * t.heads,
                                       (char*)t.dq_sel + (size_t)g.seq0 * d * esz, s));
        }
        // d(ln_1 output) = dK, dV rows . W_kv  (K range d .. 3d of the transposed in_proj weight)  +  on the selected rows  dq . W_q
        GemmArgs g4; g4.A = (char*)t.dqkv + (size_t)d * esz; g4.lda = 3 * d; g4.B = (char*)w.w_in_t + (size_t)d * esz; g4.ldb = 3 * d; g4.M = M; g4.N = d; g4.K = 2 * d;
        g4.out0 = t.h; g4.ldo0 = d;
        TRY(gemm_call(m, EPI_STORE, g4, s, !t.causal));
        GemmArgs g5; g5.A = t.dq_sel; g5.lda = d; g5.B = w.w_in_t; g5.ldb = 3 * d; g5.M = S; g5.N = d; g5.K = d; g5.out0 = t.dqx_sel; g5.ldo0 = d;
        TRY(gemm_call(m, EPI_STORE, g5, s, !t.causal));
        TRY(launch_add_rows(dt, t.dqx_sel, t.tail_rows, t.h, S, d, s));
    } else {
    // attention backward over all keys: d(attention output) is zero except on the selected query rows
    HIP_TRY(hipMemsetAsync(t.dattn, 0, (size_t)M * d * esz, s));
    TRY(launch_scatter_rows(t.dattn_sel, (size_t)d * esz, t.tail_rows, t.dattn, (size_t)d * esz, S, d * (int)esz, s));
    for (const Tower::Seg& g : segs) {
        AttnArgs at; at.qkv = (const char*)a.qkv + (size_t)g.row0 * 3 * d * esz; at.out = (char*)a.attn + (size_t)g.row0 * d * esz; at.lse = a.lse + g.lse0;
        at.dout = (const char*)t.dattn + (size_t)g.row0 * d * esz; at.dqkv = (char*)t.dqkv + (size_t)g.row0 * 3 * d * esz; at.delta = t.delta + g.lse0;
        at.B = g.nseq; at.L = g.L; at.H = t.heads; at.causal = t.causal;
        at.sel_rows = segs.size() > 1 ? t.tail_local + g.seq0 : t.tail_rows;  // d(attention output) is zero except on those rows: the kernels skip the all-zero query blocks
        TRY(attn_call(m, t, at, true, s));
    }
    GemmArgs g4; g4.A = t.dqkv; g4.lda = 3 * d; g4.B = w.w_in_t; g4.ldb = 3 * d; g4.M = M; g4.N = d; g4.K = 3 * d; g4.out0 = t.h; g4.ldo0 = d;
    TRY(gemm_call(m, EPI_STORE, g4, s, !t.causal));
    }
    // the residual path into ln_1's input: d(x_mid), zero except on the selected rows
    HIP_TRY(hipMemsetAsync(t.dx_lp, 0, (size_t)M * d * esz, s));
    TRY(launch_scatter_rows(t.dsel_lp, (size_t)d * esz, t.tail_rows, t.dx_lp, (size_t)d * esz, S, d * (int)esz, s));
    if (!m->lp_grad) {
        HIP_TRY(hipMemsetAsync(t.dx, 0, (size_t)M * d * 4, s));
        TRY(launch_scatter_rows(t.dsel, (size_t)d * 4, t.tail_rows, t.dx, (size_t)d * 4, S, d * 4, s));
    }
    LnBwdArgs b1; b1.dy = t.h; b1.lddy = d; b1.x = a.x_in; b1.ldx = d; b1.mean = a.mean1; b1.rstd = a.rstd1; b1.gamma = w.ln1_g; b1.lddres = d;
    if (m->lp_grad) b1.dres_lp = t.dx_lp; else { b1.dres = t.dx; b1.dx = t.dx; }
    b1.lddx = d; b1.dx_lp = t.dx_lp; b1.lddx_lp = d; b1.rows = M; b1.d = d;
    TRY(ln_bwd_call(m, t, b1, s));
    return MUDPT_OK;
}

// in: t.dx / t.dx_lp = gradient w.r.t. the block output; out: the same buffers = gradient w.r.t. x_in
// side != null: block i's input had deep-prompt rows spliced in; their gradient goes to side[seq][n_ctx][d] (row stride side_ldb per
// sequence) and the stream gets zeros on those rows (LnBwdArgs::side)
static int block_bwd(mudpt_model* m, Tower& t, int i, int nseq, hipStream_t s, float* side = nullptr, size_t side_ldb = 0) {
    const int M = tower_rows(t, nseq), d = t.d, dt = m->dtype;
    BlockW& w = t.w[i];
    BlockAct& a = t.a[i];
    GemmArgs g1; g1.A = t.dx_lp; g1.lda = d; g1.B = w.w_proj_t; g1.ldb = d; g1.M = M; g1.N = 4 * d; g1.K = d; g1.out0 = t.g; g1.ldo0 = 4 * d; g1.aux = a.u; g1.ldaux = 4 * d;
    g1.gelu_q8 = m->gelu_q8 && t.split == LO_NONE;  // as the forward stored it (block_fwd)
    TRY(gemm_call(m, EPI_GELU_BWD, g1, s, !t.causal));
    GemmArgs g2; g2.A = t.g; g2.lda = 4 * d; g2.B = w.w_fc_t; g2.ldb = 4 * d; g2.M = M; g2.N = d; g2.K = 4 * d; g2.out0 = t.h; g2.ldo0 = d;
    TRY(gemm_call(m, EPI_STORE, g2, s, !t.causal));
    LnBwdArgs b2; b2.dy = t.h; b2.lddy = d; b2.x = a.x_mid; b2.ldx = d; b2.mean = a.mean2; b2.rstd = a.rstd2; b2.gamma = w.ln2_g; b2.lddres = d;
    if (m->lp_grad) b2.dres_lp = t.dx_lp; else { b2.dres = t.dx; b2.dx = t.dx; }
    b2.lddx = d; b2.dx_lp = t.dx_lp; b2.lddx_lp = d; b2.rows = M; b2.d = d;
    TRY(ln_bwd_call(m, t, b2, s));
    GemmArgs g3; g3.A = t.dx_lp; g3.lda = d; g3.B = w.w_out_t; g3.ldb = d; g3.M = M; g3.N = d; g3.K = d; g3.out0 = t.dattn; g3.ldo0 = d;
    TRY(gemm_call(m, EPI_STORE, g3, s, !t.causal));
    for (const Tower::Seg& g : tower_segs(t, nseq)) {
        const size_t esz = 2;
        AttnArgs at; at.qkv = (const char*)a.qkv + (size_t)g.row0 * 3 * d * esz; at.out = (char*)a.attn + (size_t)g.row0 * d * esz; at.lse = a.lse + g.lse0;
        at.dout = (const char*)t.dattn + (size_t)g.row0 * d * esz; at.dqkv = (char*)t.dqkv + (size_t)g.row0 * 3 * d * esz; at.delta = t.delta + g.lse0;
        at.B = g.nseq; at.L = g.L; at.H = t.heads; at.causal = t.causal;
        // block 0: only the prompt rows of dqkv are read below (Tower::head_rows)
        if (i == 0 && t.head_rows && t.layers > 1 && m->attn_window) { at.win_row0 = t.prompt_row0; at.win_n = t.head_n; }
        TRY(attn_call(m, t, at, true, s));
    }
    if (i == 0 && t.head_rows && t.layers > 1) {
        // block 0: d(x_in) on the prompt rows only (Tower::head_rows); the other rows of t.dx / t.dx_lp are left stale and
        // nothing reads them (the splice reductions and ln_pre's backward touch prompt rows only)
        const int R = nseq * t.head_n;
        TRY(launch_gather_rows(t.dqkv, (size_t)3 * d * 2, t.head_rows, t.hd_dqkv, (size_t)3 * d * 2, R, 3 * d * 2, s));
        GemmArgs g4; g4.A = t.hd_dqkv; g4.lda = 3 * d; g4.B = w.w_in_t; g4.ldb = 3 * d; g4.M = R; g4.N = d; g4.K = 3 * d; g4.out0 = t.hd_h; g4.ldo0 = d;
        TRY(gemm_call(m, EPI_STORE, g4, s, !t.causal));
        LnBwdArgs b1; b1.dy = t.hd_h; b1.lddy = d; b1.x = a.x_in; b1.ldx = d; b1.row_index = t.head_rows; b1.stats_by_token = true;
        b1.mean = a.mean1; b1.rstd = a.rstd1; b1.gamma = w.ln1_g; b1.lddres = d;
        if (m->lp_grad) b1.dres_lp = t.dx_lp; else { b1.dres = t.dx; b1.dx = t.dx; }
        b1.lddx = d; b1.dx_lp = t.dx_lp; b1.lddx_lp = d; b1.rows = R; b1.d = d;
        TRY(launch_ln_bwd(dt, b1, s));
        return MUDPT_OK;
    }
    GemmArgs g4; g4.A = t.dqkv; g4.lda = 3 * d; g4.B = w.w_in_t; g4.ldb = 3 * d; g4.M = M; g4.N = d; g4.K = 3 * d; g4.out0 = t.h; g4.ldo0 = d;
    TRY(gemm_call(m, EPI_STORE, g4, s, !t.causal));
    LnBwdArgs b1; b1.dy = t.h; b1.lddy = d; b1.x = a.x_in; b1.ldx = d; b1.mean = a.mean1; b1.rstd = a.rstd1; b1.gamma = w.ln1_g; b1.lddres = d;
    if (m->lp_grad) b1.dres_lp = t.dx_lp; else { b1.dres = t.dx; b1.dx = t.dx; }
    b1.lddx = d; b1.dx_lp = t.dx_lp; b1.lddx_lp = d; b1.rows = M; b1.d = d;
    if (side) { b1.side = side; b1.side_row0 = t.prompt_row0; b1.side_n = m->cfg.n_ctx; b1.side_L = t.L; b1.side_ldb = side_ldb; }
    TRY(ln_bwd_call(m, t, b1, s));
    return MUDPT_OK;
}

// Vision tower forward, clip/model.py:526-553 (MuDPT: prompt rows appended before ln_pre, deep prompts spliced per block) or
// clip/model.py:478-496 (CoCoOp: the vanilla ViT, no prompt rows) -> m->img_f [B, e]
static int vision_forward(mudpt_model* m, const float* images, int B, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dv = c.v_width, e = c.embed_dim, n = c.n_ctx, D1 = c.depth - 1;
    const int P = (c.image_size / c.patch) * (c.image_size / c.patch), Lv = m->vis.L, K0 = (3 * c.patch * c.patch + 63) / 64 * 64;
    float* Pm = m->params;
    const int m_patch = site_mode(m->vis, SITE_PATCH, K0);  // parity mode: split pixels (an fp16 pixel alone carries 2.4e-4 of rounding into block 0)
    if (m_patch != LO_NONE) TRY(launch_patchify_split(m->dtype, images, m->patches, m->patches_lo, m_patch, B, c.image_size, c.patch, K0, s));
    else TRY(launch_patchify(m->dtype, images, m->patches, B, c.image_size, c.patch, K0, s));
    GemmArgs pe; pe.A = m->patches; pe.lda = K0; pe.B = m->conv_w; pe.ldb = K0; pe.M = B * P; pe.N = dv; pe.K = K0; pe.out0 = m->xpre; pe.ldo0 = dv;
    split_operand(pe, m_patch, m->patches_lo, m->conv_w8, m->conv_s8);
    pe.patches = P; pe.seq_len = Lv; pe.pos = m->vpos;
    TRY(gemm_call(m, EPI_PATCH, pe, s));
    TRY(launch_set_rows(m->xpre, B, Lv, dv, 0, 1, m->cls, m->vpos, s));
    if (!m->cocoop) TRY(launch_set_rows(m->xpre, B, Lv, dv, Lv - n, n, Pm + m->off[P_VCTX], m->shared, s));
    LnFwdArgs lp; lp.x = m->xpre; lp.ldx = dv; lp.gamma = m->ln_pre_g; lp.beta = m->ln_pre_b; lp.out = m->vis.a[0].x_in; lp.ldo = dv; lp.out_f32 = true;
    lp.mean = m->pre_mean; lp.rstd = m->pre_rstd; lp.rows = B * Lv; lp.d = dv;
    TRY(launch_ln_fwd(m->dtype, lp, s));
    for (int i = 0; i < m->vis.layers; ++i) {
        TRY(block_fwd(m, m->vis, i, B, (i >= 1 && i - 1 < D1) ? m->vis_deep + (size_t)(i - 1) * n * dv : nullptr, s));
    }
    LnFwdArgs lq; lq.x = m->vis.xout_sel; lq.ldx = dv; lq.gamma = m->ln_post_g; lq.beta = m->ln_post_b; lq.out = m->f_ln; lq.ldo = dv;
    lq.out_f32 = true; lq.mean = m->post_mean; lq.rstd = m->post_rstd; lq.rows = B; lq.d = dv;
    TRY(launch_ln_fwd(m->dtype, lq, s));
    TRY(launch_sgemm(false, false, B, e, dv, 1.f, m->f_ln, dv, m->vproj, e, 0.f, m->img_f, e, nullptr, s));
    return MUDPT_OK;
}

// ---- CoCoOp (trainers/cocoop.py) ------------------------------------------------------------------------------------
// forward (:178-198): image features of the frozen vanilla ViT -> meta_net bias per image (:141-146) -> one text-tower pass
// over all B * C (image, class) prompts at once (the reference loops over the images, :187-194) -> logits [B, C].
// text tower over the prompts of images [i0, i0 + nb): prompts (:148-165) + positional embedding (:52), 12 causal blocks, ln_final on
// the EOT rows, text_projection -> rows i0 * C .. of m->txt_f
static int cocoop_text_chunk(mudpt_model* m, int i0, int nb, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dt = c.t_width, e = c.embed_dim, n = c.n_ctx, C = c.n_cls, Lt = m->txt.L, TS = nb * C;
    const size_t r0 = (size_t)i0 * C;
    TRY(launch_cocoop_prompts(m->txt.a[0].x_in, m->emb_pos, m->params + m->off[Q_CTX], m->mn_bias + (size_t)i0 * dt, m->tpos, nb, C, Lt, dt, n, s));
    for (int i = 0; i < m->txt.layers; ++i) TRY(block_fwd(m, m->txt, i, TS, nullptr, s));
    LnFwdArgs lf; lf.x = m->txt.xout_sel; lf.ldx = dt; lf.gamma = m->ln_fin_g; lf.beta = m->ln_fin_b; lf.out = m->t_ln + r0 * dt; lf.ldo = dt;
    lf.out_f32 = true; lf.mean = m->fin_mean + r0; lf.rstd = m->fin_rstd + r0; lf.rows = TS; lf.d = dt;
    TRY(launch_ln_fwd(m->dtype, lf, s));
    TRY(launch_sgemm(false, false, TS, e, dt, 1.f, m->t_ln + r0 * dt, dt, m->tproj, e, 0.f, m->txt_f + r0 * e, e, nullptr, s));
    return MUDPT_OK;
}

static HeadArgs cocoop_head_args(mudpt_model* m, int i0, int nb, int B) {
    const mudpt_config& c = m->cfg;
    const size_t C = c.n_cls, e = c.embed_dim, r0 = (size_t)i0 * C;
    HeadArgs h; h.img = m->img_f + (size_t)i0 * e; h.txt = m->txt_f + r0 * e; h.scale = m->scale; h.logits = m->logits + r0;
    h.img_n = m->img_n + (size_t)i0 * e; h.txt_n = m->txt_n + r0 * e; h.img_inv = m->img_inv + i0; h.txt_inv = m->txt_inv + r0;
    h.dlogits = m->dlogits + r0; h.row_loss = m->row_loss + i0; h.dtxt = m->dtxt + r0 * e; h.loss = m->loss;
    h.B = nb; h.B_total = B; h.C = (int)C; h.e = (int)e;
    return h;
}

// image features of the frozen vanilla ViT, normalised (:183), and meta_net: linear1 -> ReLU -> linear2 (:103-107, fp32) -> the
// per-image context shift m->mn_bias [B, dt]
static int cocoop_image_side(mudpt_model* m, const float* images, int B, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dt = c.t_width, e = c.embed_dim, hd = m->hid;
    float* Pm = m->params;
    TRY(vision_forward(m, images, B, s));
    TRY(launch_l2norm(m->img_f, m->img_n, m->img_inv, B, e, s));
    TRY(launch_sgemm(false, true, B, hd, e, 1.f, m->img_n, e, Pm + m->off[Q_W1], e, 0.f, m->mn_hid, hd, Pm + m->off[Q_B1], s));
    TRY(launch_relu(m->mn_hid, (size_t)B * hd, s));
    TRY(launch_sgemm(false, true, B, dt, hd, 1.f, m->mn_hid, hd, Pm + m->off[Q_W2], hd, 0.f, m->mn_bias, dt, Pm + m->off[Q_B2], s));
    return MUDPT_OK;
}

// forward (:178-198): image features -> meta_net bias per image (:141-146) -> the text tower over the (image, class) prompts, a chunk
// of images per pass (the reference loops image by image, :187-194) -> logits [B, C].
static int cocoop_forward(mudpt_model* m, const float* images, int B, hipStream_t s) {
    TRY(cocoop_image_side(m, images, B, s));
    for (int i0 = 0; i0 < B; i0 += m->txt_chunk) {
        const int nb = B - i0 < m->txt_chunk ? B - i0 : m->txt_chunk;
        TRY(cocoop_text_chunk(m, i0, nb, s));
        TRY(launch_pair_head_fwd(cocoop_head_args(m, i0, nb, B), s));
    }
    return MUDPT_OK;
}

// forward + F.cross_entropy (:196-197) + backward w.r.t. ctx and meta_net (:222-226 freeze rule: "prompt_learner" only).  Per chunk of
// images: text forward, logits, CE rows, text backward, ctx / bias gradients accumulated in chunk order (fixed: reproducible).
static int cocoop_forward_backward(mudpt_model* m, const float* images, const int64_t* labels, int B, float grad_scale, float* loss, float* logits, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dt = c.t_width, e = c.embed_dim, n = c.n_ctx, C = c.n_cls, Lt = m->txt.L, hd = m->hid;
    float *Pm = m->params, *G = m->grads;
    TRY(cocoop_image_side(m, images, B, s));
    HIP_TRY(hipMemsetAsync(G, 0, m->total * 4, s));
    const float unscale = grad_scale / ((float)B * m->loss_scale);  // static loss scaling, as in mudpt_forward_backward
    Tower& X = m->txt;
    for (int i0 = 0; i0 < B; i0 += m->txt_chunk) {
        const int nb = B - i0 < m->txt_chunk ? B - i0 : m->txt_chunk, TS = nb * C;
        const size_t r0 = (size_t)i0 * C;
        TRY(cocoop_text_chunk(m, i0, nb, s));
        HeadArgs h = cocoop_head_args(m, i0, nb, B);
        h.labels = labels + i0; h.grad_scale = m->loss_scale * (float)B;
        TRY(launch_pair_head_fwd(h, s));
        TRY(launch_pair_head_bwd(h, s));  // CE rows, dlogits, d(text features) of this chunk; the mean over all B rows follows the loop
        TRY(launch_sgemm(false, true, TS, dt, e, 1.f, m->dtxt + r0 * e, e, m->tproj, e, 0.f, m->dt_ln + r0 * dt, dt, nullptr, s));
        LnBwdArgs bf; bf.dy = m->dt_ln + r0 * dt; bf.lddy = dt; bf.dy_f32 = true; bf.x = X.xout_sel; bf.ldx = dt; bf.mean = m->fin_mean + r0; bf.rstd = m->fin_rstd + r0;
        bf.gamma = m->ln_fin_g; bf.dx = m->lp_grad ? nullptr : X.dsel; bf.lddx = dt; bf.dx_lp = X.dsel_lp; bf.lddx_lp = dt; bf.rows = TS; bf.d = dt;
        TRY(launch_ln_bwd(m->dtype, bf, s));
        for (int i = X.layers - 1; i >= 0; --i) {
            if (i == X.layers - 1) TRY(block_bwd_tail(m, X, TS, s)); else TRY(block_bwd(m, X, i, TS, s));
        }
        // d ctx += sum over the chunk's (image, class) prompts of the context rows' gradient; d bias[i] = the same sum over image i's prompts
        TRY(launch_reduce_rows(m->dtype, m->lp_grad ? nullptr : X.dx, m->lp_grad ? X.dx_lp : nullptr, TS, Lt, dt, 1, n, G + m->off[Q_CTX], false, true, unscale, s));
        TRY(launch_cocoop_dbias(m->dtype, m->lp_grad ? nullptr : X.dx, m->lp_grad ? X.dx_lp : nullptr, m->mn_dbias + (size_t)i0 * dt, nb, C, Lt, dt, n, unscale, s));
    }
    TRY(launch_mean(m->row_loss, B, m->loss, s));
    HIP_TRY(hipMemcpyAsync(loss, m->loss, 4, hipMemcpyDeviceToDevice, s));
    if (logits) HIP_TRY(hipMemcpyAsync(logits, m->logits, (size_t)B * C * 4, hipMemcpyDeviceToDevice, s));
    // meta_net backward (fp32, tiny): linear2, ReLU, linear1; its input (the normalised image features) is a constant
    TRY(launch_sgemm(true, false, dt, hd, B, 1.f, m->mn_dbias, dt, m->mn_hid, hd, 0.f, G + m->off[Q_W2], hd, nullptr, s));
    TRY(launch_colsum(m->mn_dbias, B, dt, dt, G + m->off[Q_B2], false, s));
    TRY(launch_sgemm(false, false, B, hd, dt, 1.f, m->mn_dbias, dt, Pm + m->off[Q_W2], hd, 0.f, m->mn_dhid, hd, nullptr, s));
    TRY(launch_relu_bwd(m->mn_dhid, m->mn_hid, (size_t)B * hd, s));
    TRY(launch_sgemm(true, false, hd, e, B, 1.f, m->mn_dhid, hd, m->img_n, e, 0.f, G + m->off[Q_W1], e, nullptr, s));
    TRY(launch_colsum(m->mn_dhid, B, hd, hd, G + m->off[Q_B1], false, s));
    return MUDPT_OK;
}

// ---- the MuDPT step in pieces (the monolithic entry points and the class-parallel phases share them) ------------------------------
// prompt learner, trainers/mudpt.py:117-130 + clip/model.py:534-539
static int prompt_learner_forward(mudpt_model* m, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dv = c.v_width, dt = c.t_width, e = c.embed_dim, n = c.n_ctx, D1 = c.depth - 1;
    float* Pm = m->params;
    TRY(launch_sgemm(false, true, n, dv, dt, 1.f, Pm + m->off[P_CTX], dt, Pm + m->off[P_EW], dt, 0.f, m->shared, dv, Pm + m->off[P_EB], s));
    if (D1 > 0) {
        TRY(launch_sgemm(false, true, D1 * n, dv, dt, 1.f, Pm + m->off[P_DEEP], dt, Pm + m->off[P_DW], dt, 0.f, m->t2v, dv, Pm + m->off[P_DB], s));
        TRY(launch_sgemm(false, true, D1 * n, e, dv, 1.f, Pm + m->off[P_VDEEP], dv, Pm + m->off[P_VW], dv, 0.f, m->v2t, e, Pm + m->off[P_VB], s));
        TRY(launch_add(m->t2v, Pm + m->off[P_VDEEP], m->vis_deep, (size_t)D1 * n * dv, s));
        TRY(launch_add(m->v2t, Pm + m->off[P_DEEP], m->txt_deep, (size_t)D1 * n * dt, s));
    }
    return MUDPT_OK;
}

// text tower, trainers/mudpt.py:142-156, over this handle's classes [c0, c0 + ct): rows c0.. of the [n_cls, e] feature table
static int text_forward(mudpt_model* m, hipStream_t s2) {
    const mudpt_config& c = m->cfg;
    const int dt = c.t_width, e = c.embed_dim, n = c.n_ctx, D1 = c.depth - 1, Ct = m->ct;
    float* Pm = m->params;
    const std::vector<Tower::Seg> segs = tower_segs(m->txt, Ct);
    const bool packed = segs.size() > 1;
    HIP_TRY(hipMemcpyAsync(m->txt.a[0].x_in, m->emb_pos, (size_t)tower_rows(m->txt, Ct) * dt * 4, hipMemcpyDeviceToDevice, s2));
    for (const Tower::Seg& g : segs)
        TRY(launch_set_rows(m->txt.a[0].x_in + (size_t)g.row0 * dt, g.nseq, g.L, dt, 1, n, Pm + m->off[P_CTX], m->tpos + dt, s2));
    for (int i = 0; i < m->txt.layers; ++i) {
        TRY(block_fwd(m, m->txt, i, Ct, (i >= 1 && i - 1 < D1) ? m->txt_deep + (size_t)(i - 1) * n * dt : nullptr, s2));
    }
    LnFwdArgs lf; lf.x = m->txt.xout_sel; lf.ldx = dt; lf.gamma = m->ln_fin_g; lf.beta = m->ln_fin_b; lf.out = m->t_ln; lf.ldo = dt;
    lf.out_f32 = true; lf.mean = m->fin_mean; lf.rstd = m->fin_rstd; lf.rows = Ct; lf.d = dt;
    TRY(launch_ln_fwd(m->dtype, lf, s2));
    if (m->sharded) HIP_TRY(hipMemsetAsync(m->txt_f, 0, (size_t)c.n_cls * e * 4, s2));  // other ranks' rows: zero, so a sum completes the table
    float* feat = m->txt_f + (size_t)m->c0 * e;
    TRY(launch_sgemm(false, false, Ct, e, dt, 1.f, m->t_ln, dt, m->tproj, e, 0.f, packed ? m->txt_sorted : feat, e, nullptr, s2));
    if (packed) TRY(launch_scatter_rows(m->txt_sorted, (size_t)e * 4, m->class_perm, feat, (size_t)e * 4, Ct, e * 4, s2));  // back to the caller's class order
    m->text_valid = true;
    return MUDPT_OK;
}

// Both towers' forward.  The text tower is independent of the vision tower once the prompt learner has run, so it goes to the side
// stream (enqueued first): its ~200 small launch-latency-bound kernels fill the CUs the big vision kernels leave idle (tails of the
// persistent GEMMs, memory-bound LayerNorms) instead of serialising behind them.  With reuse_text (inference with unchanged
// parameters: the reference recomputes the text tower for every test batch, trainers/mudpt.py:170-184) the text
// features of the previous call are kept.
static int towers_forward(mudpt_model* m, const float* images, int B, hipStream_t s, bool reuse_text) {
    TRY(prompt_learner_forward(m, s));
    if (!reuse_text) {
        HIP_TRY(hipEventRecord(m->ev_fork, s));
        HIP_TRY(hipStreamWaitEvent(m->s2, m->ev_fork, 0));
        TRY(text_forward(m, m->s2));
        HIP_TRY(hipEventRecord(m->ev_join, m->s2));
    }
    TRY(vision_forward(m, images, B, s));  // clip/model.py:526-553
    if (!reuse_text) HIP_TRY(hipStreamWaitEvent(s, m->ev_join, 0));
    return MUDPT_OK;
}

// cosine logits, trainers/mudpt.py:178-182 (needs both towers' features; txt_f must hold all n_cls rows)
static int head_forward(mudpt_model* m, int B, bool reuse_text, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    HeadArgs h; h.img = m->img_f; h.txt = m->txt_f; h.scale = m->scale; h.logits = m->logits; h.img_n = m->img_n; h.txt_n = m->txt_n;
    h.img_inv = m->img_inv; h.txt_inv = m->txt_inv; h.B = B; h.C = c.n_cls; h.e = c.embed_dim;
    if (head_fused_fits(h, false)) {
        if (reuse_text) h.txt = nullptr;  // the normalised text features of the previous call are still in m->txt_n
        TRY(launch_head_fused_fwd(h, s));
    } else {
        TRY(launch_head_fwd(h, s));
    }
    return MUDPT_OK;
}

static int forward_impl(mudpt_model* m, const float* images, int B, hipStream_t s, bool reuse_text = false, bool skip_head = false) {
    if (m->cocoop) return cocoop_forward(m, images, B, s);
    TRY(towers_forward(m, images, B, s, reuse_text));
    if (skip_head) return MUDPT_OK;  // the training step runs the fused forward + cross-entropy + backward head itself
    return head_forward(m, B, reuse_text, s);
}

static int not_sharded(mudpt_model* m, const char* what) {
    if (m->sharded) { set_error("%s: this handle encodes classes %d..%d of %d only (mudpt_set_class_shard): use the mudpt_cp_* phases", what, m->c0, m->c0 + m->ct - 1, m->cfg.n_cls); return MUDPT_ERR_STATE; }
    return MUDPT_OK;
}

extern "C" int mudpt_forward(mudpt_model* m, const float* images, int32_t B, float* logits, void* stream) {
    return mudpt_forward_ex(m, images, B, logits, 0, stream);
}

extern "C" int mudpt_forward_ex(mudpt_model* m, const float* images, int32_t B, float* logits, int32_t flags, void* stream) {
    TRY(ready(m, B, false));
    TRY(not_sharded(m, "forward"));
    ARG_CHECK(images && logits, "forward: null argument");
    hipStream_t s = (hipStream_t)stream;
    const bool reuse = (flags & MUDPT_FWD_REUSE_TEXT) != 0 && !m->cocoop;  // CoCoOp's text features depend on the image
    if (reuse && !m->text_valid) { set_error("forward: MUDPT_FWD_REUSE_TEXT before any text-tower pass"); return MUDPT_ERR_STATE; }
    m->train_fwd = false;
    TRY(forward_impl(m, images, B, s, reuse));
    HIP_TRY(hipMemcpyAsync(logits, m->logits, (size_t)B * m->cfg.n_cls * 4, hipMemcpyDeviceToDevice, s));
    return MUDPT_OK;
}

// head of the training step: cross-entropy (mean) + cosine logits backward, trainers/mudpt.py:178-182,250 -> loss, dimg, dtxt (all classes)
static int head_train(mudpt_model* m, const int64_t* labels, int B, float grad_scale, float* loss, float* logits, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int e = c.embed_dim, C = c.n_cls;
    HeadArgs h; h.img = m->img_f; h.txt = m->txt_f; h.labels = labels; h.scale = m->scale; h.logits = m->logits; h.loss = m->loss; h.dlogits = m->dlogits;
    h.row_loss = m->row_loss; h.dimg = m->dimg; h.dtxt = m->dtxt; h.img_n = m->img_n; h.txt_n = m->txt_n; h.img_inv = m->img_inv; h.txt_inv = m->txt_inv;
    // Static loss scaling: the backward pass runs on per-sample gradients times loss_scale (dlogits = (softmax -
    // onehot) * loss_scale, independent of B and of the number of ranks), so the T copies of the token gradients
    // stay inside fp16's normal range (unscaled they are ~1e-7 at B = 256: flushed).  The four reductions that leave
    // the towers multiply by `unscale`; everything after them is fp32 and linear.
    m->cp_unscale = grad_scale / ((float)B * m->loss_scale);
    h.grad_scale = m->loss_scale * (float)B; h.B = B; h.C = C; h.e = e;
    if (head_fused_fits(h, true)) {
        TRY(launch_head_fused_train(h, s));
    } else {
        TRY(head_forward(m, B, false, s));
        TRY(launch_head_bwd(h, s));
    }
    if (logits) HIP_TRY(hipMemcpyAsync(logits, m->logits, (size_t)B * C * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(loss, m->loss, 4, hipMemcpyDeviceToDevice, s));
    return MUDPT_OK;
}

// text tower backward over this handle's classes: reads rows c0.. of dtxt and its own activations, writes only its own buffers,
// d_txt_deep and the ctx slice of the gradient bucket
static int text_backward(mudpt_model* m, float unscale, hipStream_t s2) {
    const mudpt_config& c = m->cfg;
    const int dt = c.t_width, e = c.embed_dim, n = c.n_ctx, D1 = c.depth - 1, Ct = m->ct;
    float* G = m->grads;
    Tower& X = m->txt;
    const std::vector<Tower::Seg> segs = tower_segs(X, Ct);
    const float* dfeat = m->dtxt + (size_t)m->c0 * e;
    if (segs.size() > 1) {  // length buckets: the tower's sequences are in length-sorted order
        TRY(launch_gather_rows(dfeat, (size_t)e * 4, m->class_perm, m->dtxt_sorted, (size_t)e * 4, Ct, e * 4, s2));
        dfeat = m->dtxt_sorted;
    }
    TRY(launch_sgemm(false, true, Ct, dt, e, 1.f, dfeat, e, m->tproj, e, 0.f, m->dt_ln, dt, nullptr, s2));
    LnBwdArgs bf; bf.dy = m->dt_ln; bf.lddy = dt; bf.dy_f32 = true; bf.x = X.xout_sel; bf.ldx = dt; bf.mean = m->fin_mean; bf.rstd = m->fin_rstd;
    bf.gamma = m->ln_fin_g; bf.dx = m->lp_grad ? nullptr : X.dsel; bf.lddx = dt; bf.dx_lp = X.dsel_lp; bf.lddx_lp = dt; bf.rows = Ct; bf.d = dt;
    TRY(launch_ln_bwd(m->dtype, bf, s2));
    for (int i = X.layers - 1; i >= 0; --i) {
        if (i == X.layers - 1) TRY(block_bwd_tail(m, X, Ct, s2)); else TRY(block_bwd(m, X, i, Ct, s2));
        if (i >= 1 && i - 1 < D1)
            for (const Tower::Seg& g : segs)  // bucket after bucket in a fixed order: deterministic
                TRY(launch_reduce_rows(m->dtype, m->lp_grad ? nullptr : X.dx + (size_t)g.row0 * dt, (char*)X.dx_lp + (size_t)g.row0 * dt * 2, g.nseq, g.L, dt, 1, n,
                                       m->d_txt_deep + (size_t)(i - 1) * n * dt, true, g.seq0 > 0, unscale, s2));
    }
    // d ctx (text side): rows 1..n of the first block's input, summed over the class prompts
    for (const Tower::Seg& g : segs)
        TRY(launch_reduce_rows(m->dtype, m->lp_grad ? nullptr : X.dx + (size_t)g.row0 * dt, m->lp_grad ? (char*)X.dx_lp + (size_t)g.row0 * dt * 2 : nullptr, g.nseq, g.L, dt, 1, n,
                               G + m->off[P_CTX], false, true, unscale, s2));
    return MUDPT_OK;
}

static int vision_backward(mudpt_model* m, int B, float unscale, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dv = c.v_width, e = c.embed_dim, n = c.n_ctx, D1 = c.depth - 1;
    const int Lv = m->vis.L;
    Tower& V = m->vis;
    TRY(launch_sgemm(false, true, B, dv, e, 1.f, m->dimg, e, m->vproj, e, 0.f, m->df_ln, dv, nullptr, s));
    LnBwdArgs bq; bq.dy = m->df_ln; bq.lddy = dv; bq.dy_f32 = true; bq.x = V.xout_sel; bq.ldx = dv; bq.mean = m->post_mean; bq.rstd = m->post_rstd;
    bq.gamma = m->ln_post_g; bq.dx = m->lp_grad ? nullptr : V.dsel; bq.lddx = dv; bq.dx_lp = V.dsel_lp; bq.lddx_lp = dv; bq.rows = B; bq.d = dv;
    TRY(launch_ln_bwd(m->dtype, bq, s));
    // Backward of the splice: the prompt rows of d(x_in[i]) feed d(vis_deep[i-1]) and the rows the splice overwrote get no gradient.
    // ln_1's backward writes those rows, in fp32, to vsplice[b][(i - 1) n + k][:] and zeros to the stream (LnBwdArgs::side); ONE
    // fixed-order reduction over the images after the last block replaces a reduce-and-zero launch per block on the critical path.
    const int used = (V.layers - 1 < D1 ? V.layers - 1 : D1) * n;  // layers >= depth never consume a prompt
    const size_t side_ldb = (size_t)used * dv;
    for (int i = V.layers - 1; i >= 0; --i) {
        const bool spliced = i >= 1 && i - 1 < D1;
        if (i == V.layers - 1) {
            TRY(block_bwd_tail(m, V, B, s));
            if (spliced) {  // the tail's ln_1 backward is not fused: take the rows from the stream
                TRY(launch_reduce_rows(m->dtype, m->lp_grad ? nullptr : V.dx, V.dx_lp, B, Lv, dv, Lv - n, n, m->d_vis_deep + (size_t)(i - 1) * n * dv, true, false, unscale, s));
            }
        } else {
            TRY(block_bwd(m, V, i, B, s, spliced ? m->vsplice + (size_t)(i - 1) * n * dv : nullptr, side_ldb));
        }
    }
    {
        // blocks 1 .. layers-2 (the fused ones): rows 0 .. n (layers - 2) of every image's side block
        const int fused = (V.layers - 2 < D1 ? V.layers - 2 : D1) * n;
        if (fused > 0) TRY(launch_reduce_rows(m->dtype, m->vsplice, nullptr, B, used, dv, 0, fused, m->d_vis_deep, false, false, unscale, s));
    }
    // ln_pre backward on the prompt rows only (patch / CLS rows have no trainable ancestor), in place
    LnBwdArgs bp; bp.dy = m->lp_grad ? (const void*)V.dx_lp : (const void*)V.dx; bp.lddy = dv; bp.dy_f32 = !m->lp_grad; bp.x = m->xpre; bp.ldx = dv; bp.row_index = m->vprompt_rows; bp.mean = m->pre_mean; bp.rstd = m->pre_rstd;
    bp.gamma = m->ln_pre_g; bp.dx = V.dx; bp.lddx = dv; bp.rows = B * n; bp.d = dv; bp.by_token = true;
    TRY(launch_ln_bwd(m->dtype, bp, s));
    TRY(launch_reduce_rows(m->dtype, V.dx, nullptr, B, Lv, dv, Lv - n, n, m->d_vprompt0, false, false, unscale, s));
    return MUDPT_OK;
}

// prompt learner backward (fp32, tiny), in two halves: the one that needs only the TEXT tower's prompt gradients is enqueued on the text
// stream behind that tower's backward, where it overlaps the vision tower's (round 4: 5 of the 15 tiny launches leave the step's serial
// tail); the other half runs after the join.  Every gradient tensor the halves share is a sum of two terms, one from each: fp32 addition
// of two terms does not depend on which comes first, so the result is bit-identical to the one-stream order.
static int prompt_learner_backward_text(mudpt_model* m, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dv = c.v_width, dt = c.t_width, e = c.embed_dim, n = c.n_ctx, D1 = c.depth - 1;
    float *Pm = m->params, *G = m->grads;
    if (D1 <= 0) return MUDPT_OK;
    const int R = D1 * n;
    // layers >= depth never consume a prompt: rows of d_txt_deep beyond the tower depth stay zero
    const int used_t = (m->txt.layers - 1 < D1 ? m->txt.layers - 1 : D1) * n;
    if (used_t < R) HIP_TRY(hipMemsetAsync(m->d_txt_deep + (size_t)used_t * dt, 0, (size_t)(R - used_t) * dt * 4, s));
    // txt_deep = deep_prompts + visual_ctx_deep_projections(visual_ctx_deep_prompts)   (mudpt.py:175, clip/model.py:539)
    TRY(launch_add(G + m->off[P_DEEP], m->d_txt_deep, G + m->off[P_DEEP], (size_t)R * dt, s));
    TRY(launch_sgemm(true, false, e, dv, R, 1.f, m->d_txt_deep, e, Pm + m->off[P_VDEEP], dv, 1.f, G + m->off[P_VW], dv, nullptr, s));
    TRY(launch_colsum(m->d_txt_deep, R, e, e, G + m->off[P_VB], true, s));
    TRY(launch_sgemm(false, false, R, dv, e, 1.f, m->d_txt_deep, e, Pm + m->off[P_VW], dv, 1.f, G + m->off[P_VDEEP], dv, nullptr, s));
    return MUDPT_OK;
}
static int prompt_learner_backward_vision(mudpt_model* m, hipStream_t s) {
    const mudpt_config& c = m->cfg;
    const int dv = c.v_width, dt = c.t_width, n = c.n_ctx, D1 = c.depth - 1;
    float *Pm = m->params, *G = m->grads;
    // visual_ctx and shared = embed_projection(ctx) both receive d_vprompt0 (clip/model.py:534)
    TRY(launch_add(G + m->off[P_VCTX], m->d_vprompt0, G + m->off[P_VCTX], (size_t)n * dv, s));
    TRY(launch_sgemm(true, false, dv, dt, n, 1.f, m->d_vprompt0, dv, Pm + m->off[P_CTX], dt, 1.f, G + m->off[P_EW], dt, nullptr, s));
    TRY(launch_colsum(m->d_vprompt0, n, dv, dv, G + m->off[P_EB], true, s));
    TRY(launch_sgemm(false, false, n, dt, dv, 1.f, m->d_vprompt0, dv, Pm + m->off[P_EW], dt, 1.f, G + m->off[P_CTX], dt, nullptr, s));
    if (D1 > 0) {
        const int R = D1 * n;
        const int used_v = (m->vis.layers - 1 < D1 ? m->vis.layers - 1 : D1) * n;
        if (used_v < R) HIP_TRY(hipMemsetAsync(m->d_vis_deep + (size_t)used_v * dv, 0, (size_t)(R - used_v) * dv * 4, s));
        // vis_deep = deep_projections(deep_prompts) + visual_ctx_deep_prompts   (clip/model.py:537, mudpt.py:127)
        TRY(launch_add(G + m->off[P_VDEEP], m->d_vis_deep, G + m->off[P_VDEEP], (size_t)R * dv, s));
        TRY(launch_sgemm(true, false, dv, dt, R, 1.f, m->d_vis_deep, dv, Pm + m->off[P_DEEP], dt, 1.f, G + m->off[P_DW], dt, nullptr, s));
        TRY(launch_colsum(m->d_vis_deep, R, dv, dv, G + m->off[P_DB], true, s));
        TRY(launch_sgemm(false, false, R, dt, dv, 1.f, m->d_vis_deep, dv, Pm + m->off[P_DW], dt, 1.f, G + m->off[P_DEEP], dt, nullptr, s));
    }
    return MUDPT_OK;
}
static int prompt_learner_backward(mudpt_model* m, hipStream_t s) {  // both halves on one stream (the class-parallel phases)
    TRY(prompt_learner_backward_text(m, s));
    return prompt_learner_backward_vision(m, s);
}

extern "C" int mudpt_forward_backward(mudpt_model* m, const float* images, const int64_t* labels, int32_t B, float grad_scale,
                                      float* loss, float* logits, void* stream) {
    TRY(ready(m, B, true));
    ARG_CHECK(images && labels && loss, "forward_backward: null argument");
    hipStream_t s = (hipStream_t)stream;
    m->train_fwd = true;
    if (m->cocoop) return cocoop_forward_backward(m, images, labels, B, grad_scale, loss, logits, s);
    TRY(not_sharded(m, "forward_backward"));
    TRY(towers_forward(m, images, B, s, false));
    HIP_TRY(hipMemsetAsync(m->grads, 0, m->total * 4, s));
    TRY(head_train(m, labels, B, grad_scale, loss, logits, s));
    // text tower backward on the side stream (enqueued first; joins before the prompt-learner backward)
    HIP_TRY(hipEventRecord(m->ev_fork_b, s));
    HIP_TRY(hipStreamWaitEvent(m->s2, m->ev_fork_b, 0));
    TRY(text_backward(m, m->cp_unscale, m->s2));
    TRY(prompt_learner_backward_text(m, m->s2));
    HIP_TRY(hipEventRecord(m->ev_join_b, m->s2));
    TRY(vision_backward(m, B, m->cp_unscale, s));
    HIP_TRY(hipStreamWaitEvent(s, m->ev_join_b, 0));
    return prompt_learner_backward_vision(m, s);
}

// ---- class-parallel phases (SURVEY 8e second axis; the reference runs all C prompts on every replica, trainers/mudpt.py:142-156,230-233) ----
// A handle with mudpt_set_class_shard(c0, c1) encodes classes [c0, c1) only.  One step on every rank:
//   mudpt_cp_forward          both towers; this rank's rows of the [n_cls, e] text-feature table, the other rows zero
//   <exchange 1>              all-reduce(sum) (or all-gather) of the table  (mudpt_cp_buffers: feat)
//   mudpt_cp_head             logits, loss and the head's backward over the LOCAL images and ALL classes -> dimg, dfeat [n_cls, e]
//   <exchange 2>              all-reduce(sum) of dfeat: every rank's images contribute to every class
//   mudpt_cp_backward         VISION part may be enqueued before exchange 2 completes; TEXT part (local classes + prompt learner) after it
//   <the usual all-reduce of the gradient bucket>   text-side gradients are partial sums over classes, vision-side ones over images
extern "C" int mudpt_set_class_shard(mudpt_model* m, int32_t c0, int32_t c1) {
    ARG_CHECK(m, "set_class_shard: null model");
    if (m->cocoop) { set_error("set_class_shard: CoCoOp's text features depend on the image; shard the batch instead"); return MUDPT_ERR_ARG; }
    ARG_CHECK(c0 >= 0 && c1 > c0 && c1 <= m->cfg.n_cls, "set_class_shard: [%d, %d) is not a non-empty range of the %d classes", c0, c1, m->cfg.n_cls);
    m->c0 = c0; m->ct = c1 - c0;
    m->sharded = !(c0 == 0 && c1 == m->cfg.n_cls);
    m->prompts_set = false;  // the text tower is sized and its tables are built by the next mudpt_set_class_prompts
    m->text_valid = false;
    m->cp_stage = 0;
    return MUDPT_OK;
}
extern "C" int mudpt_cp_buffers(mudpt_model* m, float** feat, float** dfeat, size_t* numel) {
    ARG_CHECK(m && !m->cocoop, "cp_buffers: not a MuDPT model");
    if (feat) *feat = m->txt_f;
    if (dfeat) *dfeat = m->dtxt;
    if (numel) *numel = (size_t)m->cfg.n_cls * m->cfg.embed_dim;
    return MUDPT_OK;
}
extern "C" int mudpt_cp_forward(mudpt_model* m, const float* images, int32_t B, int32_t flags, void* stream) {
    TRY(ready(m, B, false));
    ARG_CHECK(images && !m->cocoop, "cp_forward: null images / not a MuDPT model");
    const bool reuse = (flags & MUDPT_FWD_REUSE_TEXT) != 0;
    if (reuse && !m->text_valid) { set_error("cp_forward: MUDPT_FWD_REUSE_TEXT before any text-tower pass"); return MUDPT_ERR_STATE; }
    m->train_fwd = (flags & MUDPT_FWD_TRAINING) != 0;
    TRY(towers_forward(m, images, B, (hipStream_t)stream, reuse));
    m->cp_B = B; m->cp_stage = 1;
    return MUDPT_OK;
}
extern "C" int mudpt_cp_head(mudpt_model* m, const int64_t* labels, int32_t B, float grad_scale, float* loss, float* logits, int32_t flags, void* stream) {
    TRY(ready(m, B, labels != nullptr));
    ARG_CHECK(!m->cocoop && (labels ? loss != nullptr : logits != nullptr), "cp_head: training needs labels and loss, inference needs logits");
    if (m->cp_stage < 1 || m->cp_B != B) { set_error("cp_head: call mudpt_cp_forward with the same batch first"); return MUDPT_ERR_STATE; }
    hipStream_t s = (hipStream_t)stream;
    if (!labels) {  // inference: logits only (flags & MUDPT_FWD_REUSE_TEXT: the normalised table of the previous call)
        TRY(head_forward(m, B, (flags & MUDPT_FWD_REUSE_TEXT) != 0, s));
        HIP_TRY(hipMemcpyAsync(logits, m->logits, (size_t)B * m->cfg.n_cls * 4, hipMemcpyDeviceToDevice, s));
        m->cp_stage = 0;
        return MUDPT_OK;
    }
    HIP_TRY(hipMemsetAsync(m->grads, 0, m->total * 4, s));
    TRY(head_train(m, labels, B, grad_scale, loss, logits, s));
    m->cp_stage = 2;
    return MUDPT_OK;
}
extern "C" int mudpt_cp_backward(mudpt_model* m, int32_t part, void* stream) {
    ARG_CHECK(m && !m->cocoop && (part == MUDPT_CP_VISION || part == MUDPT_CP_TEXT), "cp_backward: part must be MUDPT_CP_VISION or MUDPT_CP_TEXT");
    hipStream_t s = (hipStream_t)stream;
    if (part == MUDPT_CP_VISION) {
        if (m->cp_stage != 2) { set_error("cp_backward: call mudpt_cp_head (training) first"); return MUDPT_ERR_STATE; }
        TRY(vision_backward(m, m->cp_B, m->cp_unscale, s));
        m->cp_stage = 3;
        return MUDPT_OK;
    }
    if (m->cp_stage != 3) { set_error("cp_backward: the vision part comes first"); return MUDPT_ERR_STATE; }
    TRY(text_backward(m, m->cp_unscale, s));  // on the caller's stream: ordered behind its exchange of dfeat
    m->cp_stage = 0;
    return prompt_learner_backward(m, s);
}

extern "C" int mudpt_sgd_step(mudpt_model* m, float lr, float momentum, float wd, float dampening, int32_t nesterov, void* stream) {
    ARG_CHECK(m && m->params && m->grads, "sgd_step: parameters / gradients not bound");
    TRY(launch_sgd(m->params, m->grads, m->momentum, m->total, lr, momentum, wd, dampening, nesterov != 0, m->sgd_first, (hipStream_t)stream));
    m->sgd_first = false;
    return MUDPT_OK;
}
extern "C" int mudpt_sgd_reset(mudpt_model* m) {
    ARG_CHECK(m, "sgd_reset: null model");
    m->sgd_first = true;
    return MUDPT_OK;
}

// ---- data-parallel exchange for hosts without torch.distributed ---------------------------------------------------------------
// The ONE collective of a step: sum of the flat gradient bucket over the ranks of an RCCL communicator the caller
// created (ncclCommInitRank; one process per GPU).  RCCL is resolved at first use from the process (librccl.so.1, the SONAME torch
// ships and /opt/rocm installs), so the library has no link-time dependency on it and single-GPU hosts never load it.
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static nccl_allreduce_fn resolve_allreduce() {
    static nccl_allreduce_fn fn = nullptr;
    if (fn) return fn;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (!sym) {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {  // returns the already-loaded copy when the process has one
                sym = dlsym(h, "ncclAllReduce");
                if (sym) break;
            }
        }
    }
    fn = (nccl_allreduce_fn)sym;
    return fn;
}
extern "C" int mudpt_allreduce_grads(mudpt_model* m, void* nccl_comm, void* stream) {
    ARG_CHECK(m && nccl_comm, "allreduce_grads: null model / communicator");
    if (!m->grads) { set_error("allreduce_grads: no gradient bucket bound"); return MUDPT_ERR_STATE; }
    nccl_allreduce_fn fn = resolve_allreduce();
    if (!fn) { set_error("allreduce_grads: RCCL (librccl.so.1) is not loadable in this process: %s", dlerror()); return MUDPT_ERR_STATE; }
    const int rc = fn(m->grads, m->grads, m->total, /*ncclFloat32*/ 7, /*ncclSum*/ 0, nccl_comm, (hipStream_t)stream);
    if (rc != 0) { set_error("allreduce_grads: ncclAllReduce failed with ncclResult_t %d", rc); return MUDPT_ERR_HIP; }
    return MUDPT_OK;
}

extern "C" int mudpt_set_loss_scale(mudpt_model* m, float loss_scale) {
    ARG_CHECK(m && loss_scale > 0.f && std::isfinite(loss_scale), "set_loss_scale: scale must be positive and finite");
    m->loss_scale = loss_scale;
    return MUDPT_OK;
}

// Per-handle tuning knobs for A/B runs in one process (tools/gemm_bench.py, tests); state of THIS model only.
extern "C" int mudpt_model_set(mudpt_model* m, const char* name, int32_t value) {
    ARG_CHECK(m && name, "model_set: null argument");
    if (!strcmp(name, "gemm_variant")) { m->gemm_variant = value; return MUDPT_OK; }
    // both stream copies are always allocated.  lp_grad = 0 (bf16 mode) also returns the forward's update stream to fp32, as before the two were separate knobs
    if (!strcmp(name, "lp_grad")) { m->lp_grad = value != 0; if (m->dtype == MUDPT_BF16) m->lp_upd = value != 0; return MUDPT_OK; }
    if (!strcmp(name, "lp_upd")) { m->lp_upd = value != 0; return MUDPT_OK; }
    if (!strcmp(name, "gelu_q8")) { m->gelu_q8 = value != 0; return MUDPT_OK; }  // between steps only: the backward decodes what the forward stored
    if (!strcmp(name, "txt_trim")) { m->txt_trim = value != 0; m->prompts_set = false; return MUDPT_OK; }  // read by the next mudpt_set_class_prompts
    if (!strcmp(name, "attn_two_kernels")) { m->attn_two_kernels = value != 0; return MUDPT_OK; }
    if (!strcmp(name, "fwd_split_k")) { m->fwd_split_k = value != 0; return MUDPT_OK; }
    if (!strcmp(name, "attn_fused_w1")) { m->attn_fused_w1 = value != 0; return MUDPT_OK; }
    if (!strcmp(name, "split_k")) { m->split_k = value != 0; return MUDPT_OK; }
    if (!strcmp(name, "attn_window")) { m->attn_window = value != 0; return MUDPT_OK; }
    if (!strcmp(name, "last_single")) { m->last_single = value != 0; return MUDPT_OK; }  // (a tower with the fp32 attention forward runs the general kernels anyway)
    if (!strcmp(name, "txt_bucket_cost")) { m->txt_bucket_cost = value > 0 ? value : 0; m->prompts_set = false; return MUDPT_OK; }
    if (!strcmp(name, "txt_buckets")) { m->txt_buckets = value > 1 ? value : 1; m->prompts_set = false; return MUDPT_OK; }  // read by the next mudpt_set_class_prompts
    if (!strcmp(name, "prof_stride")) { m->prof_stride = value > 1 ? value : 1; return MUDPT_OK; }
    if (!strcmp(name, "cocoop_chunk")) { m->cocoop_chunk = value; m->prompts_set = false; return MUDPT_OK; }  // likewise
    // Split operands (Tower::split).  Effective from the next forward: the low-half buffers and the e4m3 weights exist whenever
    // the tower may split at all (vision tower: the parity mode MUDPT_F32; text tower: MUDPT_F16 and MUDPT_F32).
    //   vis_lo / txt_lo          0 = no low halves, 1 = fp16 pairs (22 bits), 2 = e4m3 remainders on the fp8 matrix pipe (vision tower only)
    //   txt_split                the round-2 name of txt_lo = 0 / 1
    //   vis_sites / txt_sites    bit mask of the sites that take part (model.cpp Site: 1 in_proj, 2 out_proj, 4 c_fc, 8 c_proj, 16 patch embed)
    //   vis_exact_attn / txt_exact_attn   attention forward in fp32 (attention_exact.hip); parity mode only
    for (Tower* t : {&m->vis, &m->txt}) {
        const char* pre = t == &m->vis ? "vis_" : "txt_";
        if (strncmp(name, pre, 4)) continue;
        const char* k = name + 4;
        if (!strcmp(k, "lo") || (t == &m->txt && !strcmp(k, "split"))) {
            if (value != LO_NONE && !t->may_split) { set_error("model_set: %s needs a mode whose %s tower keeps low halves (dtype fp32%s)", name, t == &m->vis ? "vision" : "text", t == &m->txt ? " or fp16" : ""); return MUDPT_ERR_STATE; }
            ARG_CHECK(value == LO_NONE || value == LO_F16 || (value == LO_F8 && !t->w.empty() && t->w[0].w_in8), "model_set: %s = %d is not available for this tower", name, value);
            t->split = value;
            m->text_valid = false;
            return MUDPT_OK;
        }
        if (!strcmp(k, "sites")) { t->sites = value & 0x1f; m->text_valid = false; return MUDPT_OK; }
        if (!strcmp(k, "exact_attn")) {
            if (value && !(t->may_split && m->exact)) { set_error("model_set: %s needs the parity mode (dtype fp32)", name); return MUDPT_ERR_STATE; }
            t->exact_attn = value != 0;
            m->text_valid = false;
            return MUDPT_OK;
        }
    }
    set_error("model_set: unknown knob '%s'", name);
    return MUDPT_ERR_ARG;
}

extern "C" int mudpt_profile_enable(mudpt_model* m, int32_t enable) {
    ARG_CHECK(m, "profile_enable: null model");
    m->prof = enable != 0;
    m->prof_mask = enable == 1 ? 1 : enable;  // 1 = the dominant kernel only (class 0); otherwise a bit mask of MUDPT_PROF_CLASSES classes
    m->ev_used = 0;
    m->ev_rec.clear();
    m->exec_flop = 0;
    m->pp_seen = 0;
    return MUDPT_OK;
}
// Synchronises the device; sums per kernel class over the launches recorded since the last enable / read, then clears the records.
// out arrays have MUDPT_PROF_CLASSES entries: ms, work (class 0: algorithmic FLOPs 2 M N K; classes 1-4: algorithmic HBM bytes), launches.
extern "C" int mudpt_profile_read_classes(mudpt_model* m, double* ms, double* work, int64_t* launches, double* executed_flop) {
    ARG_CHECK(m && ms && work && launches, "profile_read: null argument");
    HIP_TRY(hipDeviceSynchronize());
    for (int c = 0; c < PC_COUNT; ++c) { ms[c] = 0; work[c] = 0; launches[c] = 0; }
    for (size_t i = 0; i < m->ev_used; i += 2) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, m->ev[i], m->ev[i + 1]));
        const mudpt_model::ProfRec& r = m->ev_rec[i / 2];
        ms[r.cls] += t; work[r.cls] += r.work; launches[r.cls] += 1;
    }
    if (executed_flop) *executed_flop = m->exec_flop;
    m->ev_used = 0;
    m->ev_rec.clear();
    m->exec_flop = 0;
    return MUDPT_OK;
}
extern "C" int mudpt_profile_read(mudpt_model* m, double* gemm_ms, double* gemm_flop, int64_t* launches) {
    ARG_CHECK(m && gemm_ms && gemm_flop && launches, "profile_read: null argument");
    double ms[PC_COUNT], work[PC_COUNT];
    int64_t n[PC_COUNT];
    if (int rc = mudpt_profile_read_classes(m, ms, work, n, nullptr)) return rc;
    *gemm_ms = ms[PC_GEMM]; *gemm_flop = work[PC_GEMM]; *launches = n[PC_GEMM];
    return MUDPT_OK;
}

// Copy an internal fp32 activation of the LAST call to the host (synchronises the device): per-block parity tests.
// names: "vis.x_in.<i>", "vis.x_out", "txt.x_in.<i>", "txt.x_out", "image_features", "text_features"
extern "C" int mudpt_debug_read(mudpt_model* m, const char* name, int32_t batch, float* host_out, size_t capacity, size_t* numel) {
    ARG_CHECK(m && name && numel, "debug_read: null argument");
    const std::string k(name);
    const mudpt_config& c = m->cfg;
    const float* src = nullptr;
    size_t n = 0;
    auto tower = [&](Tower& t, const std::string& rest, int nseq) {
        if (!t.segs.empty() && rest != "x_out") return;  // packed in length buckets: no [seq, L, d] view (set txt_buckets = 1 to tap)
        n = (size_t)nseq * t.L * t.d;
        if (rest == "x_out") { src = t.xout_sel; n = (size_t)nseq * t.d; }  // the used row (CLS / EOT) of every sequence only
        else if (rest.rfind("x_in.", 0) == 0) {
            const int i = atoi(rest.c_str() + 5);
            if (i >= 0 && i < t.layers) src = t.a[i].x_in;
        }
    };
    ARG_CHECK(batch > 0 && batch <= c.max_batch, "debug_read: bad batch %d", batch);
    if (k.rfind("vis.", 0) == 0) tower(m->vis, k.substr(4), batch);
    else if (k.rfind("txt.", 0) == 0) tower(m->txt, k.substr(4), m->ct);  // this handle's classes
    else if (k == "image_features") { src = m->img_f; n = (size_t)batch * c.embed_dim; }
    else if (k == "text_features") { src = m->txt_f; n = (size_t)c.n_cls * c.embed_dim; }
    ARG_CHECK(src, "debug_read: unknown tensor '%s'", name);
    *numel = n;
    if (!host_out) return MUDPT_OK;
    ARG_CHECK(capacity >= n, "debug_read: capacity %zu < %zu", capacity, n);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(host_out, src, n * 4, hipMemcpyDeviceToHost));
    return MUDPT_OK;
}

// ---- single kernels ---------------------------------------------------------------------------------------------
extern "C" int mudpt_gemm(int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda, const void* B, int32_t ldb,
                          const float* bias, void* out0, int32_t ldo0, void* out1, int32_t ldo1, const void* aux, int32_t ldaux, int32_t patches,
                          int32_t seq_len, const float* pos, int32_t variant, void* stream) {
    GemmOpts o;
    o.variant = variant & ~0x30000;
    if (variant & 0x10000) {  // unit-test hook: allow split K, with a per-device scratch made on first use (never freed)
        static float* scratch[64] = {};
        const int dev = current_device();
        if (!scratch[dev]) HIP_TRY(hipMalloc((void**)&scratch[dev], mudpt_model::kScratchElems * 4));
        o.scratch = scratch[dev];
        o.scratch_elems = mudpt_model::kScratchElems;
    }
    GemmArgs a; a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.bias = bias; a.out0 = out0; a.ldo0 = ldo0; a.out1 = out1; a.ldo1 = ldo1;
    a.aux = aux; a.ldaux = ldaux; a.patches = patches; a.seq_len = seq_len; a.pos = pos;
    a.gelu_q8 = (variant & 0x20000) != 0;  // unit-test hook: QuickGELU' in 8 bits (epilogues 1 / 3; ldo0 / ldaux are then BYTE strides)
    return launch_gemm(dtype, epi, a, (hipStream_t)stream, o);
}
extern "C" int mudpt_gemm_split(int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, const void* A, const void* A_lo, int32_t lo_mode, int32_t lda,
                                const void* B, const void* B8, int32_t b8_scale, int32_t ldb, const float* bias, void* out0, int32_t ldo0, void* out1,
                                void* out1_lo, int32_t out1_lo_mode, int32_t ldo1, const void* aux, int32_t ldaux, int32_t variant, void* stream) {
    GemmOpts o;
    o.variant = variant;
    GemmArgs a; a.A = A; a.B = B; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.bias = bias; a.out0 = out0; a.ldo0 = ldo0; a.out1 = out1; a.ldo1 = ldo1;
    a.aux = aux; a.ldaux = ldaux; a.A_lo = A_lo; a.lo_mode = lo_mode; a.B8 = B8; a.b8_scale = b8_scale; a.out1_lo = out1_lo; a.out1_lo_mode = out1_lo ? out1_lo_mode : (int)LO_F16;
    return launch_gemm(dtype, epi, a, (hipStream_t)stream, o);
}
extern "C" int mudpt_e4m3_from_f32(const float* in_host, uint8_t* out_host, size_t n, int32_t shift) {
    ARG_CHECK(in_host && out_host, "e4m3_from_f32: null argument");
    for (size_t i = 0; i < n; ++i) out_host[i] = f32_to_e4m3(std::ldexp(in_host[i], shift));
    return MUDPT_OK;
}
extern "C" int mudpt_layernorm_fwd_split(int32_t dtype, const float* x, int32_t ldx, const float* gamma, const float* beta, void* out, void* out_lo, int32_t lo_mode,
                                         int32_t ldo, int32_t rows, int32_t d, void* stream) {
    LnFwdArgs a; a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta; a.out = out; a.out_lo = out_lo; a.lo_mode = lo_mode; a.ldo = ldo; a.rows = rows; a.d = d;
    ARG_CHECK(out_lo && (lo_mode == LO_F16 || lo_mode == LO_F8), "layernorm_fwd_split: needs out_lo and lo_mode 1 / 2");
    return launch_ln_fwd(dtype, a, (hipStream_t)stream);
}
extern "C" int mudpt_layernorm_fwd(int32_t dtype, const float* x, int32_t ldx, const int32_t* row_index, const float* gamma, const float* beta, void* out,
                                   int32_t ldo, int32_t out_f32, float* mean, float* rstd, int32_t rows, int32_t d, void* stream) {
    LnFwdArgs a; a.x = x; a.ldx = ldx; a.row_index = row_index; a.gamma = gamma; a.beta = beta; a.out = out; a.ldo = ldo; a.out_f32 = out_f32 != 0;
    a.mean = mean; a.rstd = rstd; a.rows = rows; a.d = d;
    return launch_ln_fwd(dtype, a, (hipStream_t)stream);
}
extern "C" int mudpt_layernorm_bwd(int32_t dtype, const void* dy, int32_t lddy, int32_t dy_f32, const float* x, int32_t ldx, const int32_t* row_index,
                                   const float* mean, const float* rstd, const float* gamma, const float* dres, int32_t lddres, float* dx, int32_t lddx,
                                   void* dx_lp, int32_t lddx_lp, int32_t rows, int32_t d, void* stream) {
    LnBwdArgs a; a.dy = dy; a.lddy = lddy; a.dy_f32 = dy_f32 != 0; a.x = x; a.ldx = ldx; a.row_index = row_index; a.mean = mean; a.rstd = rstd; a.gamma = gamma;
    a.dres = dres; a.lddres = lddres; a.dx = dx; a.lddx = lddx; a.dx_lp = dx_lp; a.lddx_lp = lddx_lp; a.rows = rows; a.d = d;
    return launch_ln_bwd(dtype, a, (hipStream_t)stream);
}
extern "C" int mudpt_layernorm_fwd_fused(int32_t dtype, const float* x, int32_t ldx, const float* add, const void* add_lp, int32_t ldadd, const float* ov_rows,
                                         int32_t ov_row0, int32_t ov_n, int32_t ov_L, float* xout, int32_t ldxout, const float* gamma, const float* beta, void* out,
                                         int32_t ldo, int32_t out_f32, float* mean, float* rstd, int32_t rows, int32_t d, void* stream) {
    LnFwdArgs a; a.x = x; a.ldx = ldx; a.add = add; a.add_lp = add_lp; a.ldadd = ldadd; a.xout = xout; a.ldxout = ldxout; a.gamma = gamma; a.beta = beta;
    a.out = out; a.ldo = ldo; a.out_f32 = out_f32 != 0; a.mean = mean; a.rstd = rstd; a.rows = rows; a.d = d;
    if (ov_rows) { a.ov_rows = ov_rows; a.ov_row0 = ov_row0; a.ov_n = ov_n; a.ov_L = ov_L; }
    return launch_ln_fwd(dtype, a, (hipStream_t)stream);
}
extern "C" int mudpt_head(const float* img, const float* txt, const int64_t* labels, float scale, float grad_scale, int32_t B, int32_t C, int32_t e,
                          float* logits, float* loss, float* dimg, float* dtxt, void* stream) {
    ARG_CHECK(img && txt && logits && B > 0 && C > 0 && e > 0, "head: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;
    const size_t need = (size_t)B * e + (size_t)C * e + B + C + (size_t)B * C + B;
    HIP_TRY(hipMalloc((void**)&scratch, need * 4));
    HeadArgs h; h.img = img; h.txt = txt; h.labels = labels; h.scale = scale; h.logits = logits; h.loss = loss; h.dimg = dimg; h.dtxt = dtxt;
    h.img_n = scratch; h.txt_n = h.img_n + (size_t)B * e; h.img_inv = h.txt_n + (size_t)C * e; h.txt_inv = h.img_inv + B;
    h.dlogits = h.txt_inv + C; h.row_loss = h.dlogits + (size_t)B * C; h.grad_scale = grad_scale; h.B = B; h.C = C; h.e = e;
    int rc;
    if (head_fused_fits(h, labels != nullptr)) rc = labels ? launch_head_fused_train(h, s) : launch_head_fused_fwd(h, s);
    else {
        rc = launch_head_fwd(h, s);
        if (!rc && labels) rc = launch_head_bwd(h, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(scratch);
    return rc;
}
extern "C" int mudpt_reduce_rows(int32_t dtype, float* src, void* src_lp, int32_t B, int32_t L, int32_t d, int32_t row0, int32_t n, float* out,
                                 int32_t zero_src, int32_t accumulate, float scale, void* stream) {
    return launch_reduce_rows(dtype, src, src_lp, B, L, d, row0, n, out, zero_src != 0, accumulate != 0, scale, (hipStream_t)stream);
}
extern "C" int mudpt_cocoop_dbias(int32_t dtype, const float* dx_f32, const void* dx_lp, float* dbias, int32_t B, int32_t C, int32_t L, int32_t d, int32_t n,
                                  float scale, void* stream) {
    return launch_cocoop_dbias(dtype, dx_f32, dx_lp, dbias, B, C, L, d, n, scale, (hipStream_t)stream);
}
extern "C" int mudpt_sgemm(int32_t tA, int32_t tB, int32_t M, int32_t N, int32_t K, float alpha, const float* A, int32_t lda, const float* B, int32_t ldb,
                           float beta, float* C, int32_t ldc, const float* bias, void* stream) {
    return launch_sgemm(tA != 0, tB != 0, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, (hipStream_t)stream);
}
extern "C" int mudpt_attention_fwd_single(int32_t dtype, const void* qkv, const void* q_sel, const int32_t* sel_rows, void* out_sel, float* lse_sel, int32_t B, int32_t L,
                                          int32_t H, int32_t causal, void* stream) {
    AttnArgs a; a.qkv = qkv; a.sel_rows = sel_rows; a.B = B; a.L = L; a.H = H; a.causal = causal != 0;
    return launch_attn_fwd_single(dtype, a, q_sel, out_sel, nullptr, H * 64, lse_sel, (hipStream_t)stream);
}
extern "C" int mudpt_attention_bwd_single(int32_t dtype, const void* qkv, const void* q_sel, const int32_t* sel_rows, const void* out_sel, const void* dout_sel,
                                          const float* lse_sel, void* dqkv, void* dq_sel, int32_t B, int32_t L, int32_t H, int32_t causal, void* stream) {
    AttnArgs a; a.qkv = qkv; a.sel_rows = sel_rows; a.dqkv = dqkv; a.B = B; a.L = L; a.H = H; a.causal = causal != 0;
    return launch_attn_bwd_single(dtype, a, q_sel, out_sel, H * 64, dout_sel, lse_sel, dq_sel, (hipStream_t)stream);
}
extern "C" int mudpt_attention_padded_len(int32_t L) { return attn_padded_len(L); }
extern "C" int mudpt_attention_fwd(int32_t dtype, const void* qkv, void* out, float* lse, int32_t B, int32_t L, int32_t H, int32_t causal, void* stream) {
    AttnArgs a; a.qkv = qkv; a.out = out; a.lse = lse; a.B = B; a.L = L; a.H = H; a.causal = (causal & 1) != 0;
    a.tiled_fwd_16 = (causal & 2) != 0;  // 224 < L <= 640: the staged 16-query-block kernel instead of the resident form (A/B, tests)
    return launch_attn_fwd(dtype, a, (hipStream_t)stream);
}
extern "C" int mudpt_attention_fwd_exact(const float* qkv32, void* qkv_lp, void* out_hi, void* out_lo, int32_t lo_mode, int32_t ld_out, float* lse, int32_t B, int32_t L,
                                         int32_t H, int32_t causal, void* stream) {
    ARG_CHECK(!out_lo || lo_mode == LO_F16 || lo_mode == LO_F8, "attention_fwd_exact: lo_mode must be 1 (fp16) or 2 (e4m3)");
    AttnArgs a; a.qkv32 = qkv32; a.qkv_lp = qkv_lp; a.out = out_hi; a.out_lo = out_lo; a.lo_mode = out_lo ? lo_mode : (int)LO_F16; a.ld_out = ld_out; a.lse = lse; a.B = B; a.L = L; a.H = H; a.causal = causal != 0;
    return launch_attn_fwd_exact(a, (hipStream_t)stream);
}
extern "C" int mudpt_attention_bwd(int32_t dtype, const void* qkv, const void* out, const void* dout, const float* lse, float* delta, void* dqkv, int32_t B,
                                   int32_t L, int32_t H, int32_t causal, void* stream) {
    AttnArgs a; a.qkv = qkv; a.out = (void*)out; a.dout = dout; a.lse = (float*)lse; a.delta = delta; a.dqkv = dqkv; a.B = B; a.L = L; a.H = H;
    a.causal = (causal & 1) != 0; a.two_kernels = (causal & 2) != 0; a.fused_w1 = (causal & 4) != 0; a.force_fused = (causal & 12) != 0; a.sweep = (causal & 16) != 0;
    a.win_n = (causal >> 20) & 0xff; a.win_row0 = (causal >> 8) & 0xfff;  // bits 8-19: first wanted row, bits 20-27: number of wanted rows (0 = all)
    return launch_attn_bwd(dtype, a, (hipStream_t)stream);
}
