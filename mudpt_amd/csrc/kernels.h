// Host-side launchers of the gfx950 kernels.  Every launcher validates operand shapes on the host
// before the launch (a faulting kernel can reset the whole node) and returns MUDPT_OK / error code.
#pragma once
#include "common.h"

namespace mudpt {

// ------------------------------------------------------------------------------------------------
// MFMA GEMM  C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous, dtype bf16 or fp16,
// fp32 accumulate) with a fused epilogue.  nn.Linear stores weight [out,in] = B; the dX GEMMs of
// the frozen layers use a pre-transposed copy of the weight, so one layout serves fwd and bwd.
// ------------------------------------------------------------------------------------------------
enum Epilogue : int {
    EPI_STORE = 0,     // out0 (T)   = acc + bias
    EPI_GELU = 1,      // out0 (T)   = u = acc + bias ; out1 (T) = QuickGELU(u)
    EPI_RESIDUAL = 2,  // out0 (f32) = aux (f32) + acc + bias
    EPI_GELU_BWD = 3,  // out0 (T)   = acc * QuickGELU'(aux (T))
    EPI_PATCH = 4,     // out0 (f32) [(m / P) * L + 1 + m % P] = acc + pos[1 + m % P]   (patch embed)
    EPI_STORE_F32 = 5  // out0 (f32) = acc + bias
};

struct GemmArgs {
    const void* A = nullptr;  // [M, K], row stride lda (elements)
    const void* B = nullptr;  // [N, K], row stride ldb
    int M = 0, N = 0, K = 0, lda = 0, ldb = 0;
    const float* bias = nullptr;  // [N] or null
    void* out0 = nullptr; int ldo0 = 0;
    void* out1 = nullptr; int ldo1 = 0;
    void* out1_lo = nullptr;       // EPI_GELU only: the low half of out1 as a split operand (common.h LoMode; same row stride in bytes as out1)
    int out1_lo_mode = LO_F16;     // ... its form: LO_F16 (T) or LO_F8 (e4m3 bytes)
    // Split A operand (common.h): a second pass contracts the low half A_lo (same row stride in bytes as A) against B (lo_mode LO_F16)
    // or against the e4m3 weights B8 (LO_F8: [N, K] bytes at the row stride of B in bytes, block scale 2^(b8_scale - 127)); K % 128 == 0 then.
    const void* A_lo = nullptr; int lo_mode = LO_NONE;
    const void* B8 = nullptr; int b8_scale = 127;
    const void* aux = nullptr; int ldaux = 0;
    // QuickGELU' in 8 bits (common.h): EPI_GELU stores out0 = byte codes of QuickGELU'(u) (row stride ldo0 BYTES) instead of u in T;
    // EPI_GELU_BWD reads such codes from aux (row stride ldaux BYTES) instead of u
    bool gelu_q8 = false;
    int flags = 0;                 // bit 0: no XCD remap of the block id (tuning)
    int patches = 0, seq_len = 0;  // EPI_PATCH: P, L
    const float* pos = nullptr;    // EPI_PATCH: [1 + P, N]
    // split K (gemm_nt_kernel, EPI_STORE_F32 partials): slice z = blockIdx.y contracts k in [z * ksplit, (z + 1) * ksplit) and writes its
    // fp32 partial tile set to out0 + z * split_stride elements; launch_gemm sums the slices in order afterwards (splitk_reduce_kernel)
    int ksplit = 0; size_t split_stride = 0;
};
// Host-side options of one launch (never passed to the device): they belong to the calling model handle, not to the process.
struct GemmOpts {
    int variant = 0;                                 // tuning knob, see gemm.hip (mudpt_model_set "gemm_variant")
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;  // if set, a gemm_pp_kernel launch records them (start / end of the kernel)
    // fp32 scratch of the calling stream (one per tower: the towers run concurrently) for split-K partials; null = never split
    float* scratch = nullptr; size_t scratch_elems = 0;
};
int launch_gemm(int dtype, int epi, const GemmArgs& a, hipStream_t s, const GemmOpts& o = GemmOpts());
bool gemm_uses_pp(int epi, const GemmArgs& a, int variant = 0);  // true if launch_gemm dispatches to gemm_pp_kernel
int launch_gemm_pp(int dtype, int epi, const GemmArgs& a, hipStream_t s, const GemmOpts& o);  // persistent 256x256 ping-pong kernel (gemm_pp.hip)

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (fp32 statistics, eps 1e-5; clip/model.py:164-170).
// rows = number of normalised rows; row r reads x[row_index ? row_index[r] : r].
// ------------------------------------------------------------------------------------------------
struct LnFwdArgs {
    const float* x = nullptr; int ldx = 0;     // fp32 input rows
    const int* row_index = nullptr;            // optional gather of input rows
    const float* gamma = nullptr; const float* beta = nullptr;
    void* out = nullptr; int ldo = 0;          // T or fp32 (out_f32)
    void* out_lo = nullptr;                    // optional (T only): y - out, the low half of a split operand (row stride of out in bytes)
    int lo_mode = LO_F16;                      // ... its form (common.h LoMode)
    float* mean = nullptr; float* rstd = nullptr;  // [rows] saved statistics (may be null)
    int rows = 0, d = 0; bool out_f32 = false;
    // Fused residual add + prompt splice (identity row map only): v = x[r] + add[r]; rows (r % ov_L) in
    // [ov_row0, ov_row0 + ov_n) are REPLACED by ov_rows[(r % ov_L) - ov_row0] (the deep-prompt splice,
    // clip/model.py:281-297); v is written to xout[r] (the block input the backward needs) and normalised.
    const float* add = nullptr; int ldadd = 0;
    const void* add_lp = nullptr;              // ... or the addend in T (stride ldadd) when the update stream is kept in T
    float* xout = nullptr; int ldxout = 0;
    const float* ov_rows = nullptr; int ov_row0 = 0, ov_n = 0, ov_L = 1;
};
int launch_ln_fwd(int dtype, const LnFwdArgs& a, hipStream_t s, const LaunchProf* prof = nullptr);

struct LnBwdArgs {
    const void* dy = nullptr; int lddy = 0; bool dy_f32 = false;  // grad of LN output (T or fp32), compact rows
    const float* x = nullptr; int ldx = 0;    // LN input rows (gathered through row_index like fwd)
    const int* row_index = nullptr;
    const float* mean = nullptr; const float* rstd = nullptr; const float* gamma = nullptr;
    const float* dres = nullptr; int lddres = 0;  // optional residual gradient added to the result (same row map as out)
    const void* dres_lp = nullptr;                // ... or the same in T (stride lddres) when the gradient stream is kept in T
    float* dx = nullptr; int lddx = 0;        // fp32 result rows, written at row_index[r] (scatter) or r; may be null if dx_lp is set
    void* dx_lp = nullptr; int lddx_lp = 0;   // optional T copy of the result
    int rows = 0, d = 0;
    bool by_token = false;  // dy / mean / rstd rows are indexed by the token row (row_index[r]) instead of r
    bool stats_by_token = false;  // only mean / rstd are indexed by the token row; dy stays compact (row r)
    // Backward of the prompt splice, fused (identity row map only): result rows whose position inside their sequence (r % side_L) lies in
    // [side_row0, side_row0 + side_n) go, in fp32, to side[(r / side_L) * side_ldb + (pos - side_row0) * d] instead -- the gradient of
    // the spliced-in prompt row -- and ZERO is written to dx / dx_lp: the rows the splice overwrote receive no gradient (clip/model.py:281-297).
    float* side = nullptr; int side_row0 = 0, side_n = 0, side_L = 1; size_t side_ldb = 0;
};
int launch_ln_bwd(int dtype, const LnBwdArgs& a, hipStream_t s, const LaunchProf* prof = nullptr);

// ------------------------------------------------------------------------------------------------
// Attention on packed qkv [B, L, 3*H*64] (q | k | v, heads contiguous inside each third), head dim 64.
// ------------------------------------------------------------------------------------------------
struct AttnArgs {
    const void* qkv = nullptr;  // T [B, L, 3*H*64]
    void* out = nullptr;        // fwd: T [B, L, H*64] (row stride ld_out elements, 0 = H*64)
    void* out_lo = nullptr;     // fwd, optional: out_lo = O - out, the low half of a split operand (same row stride in bytes)
    int lo_mode = LO_F16;       // ... its form (common.h LoMode)
    int ld_out = 0;
    float* lse = nullptr;       // [B, H, Lp] natural-log-sum-exp of the scaled scores (Lp = padded L)
    const void* dout = nullptr; // bwd: T [B, L, H*64]
    void* dqkv = nullptr;       // bwd: T [B, L, 3*H*64]
    float* delta = nullptr;     // bwd scratch [B, H, Lp]
    // bwd, optional: dout is zero except on ONE row per sequence, token row sel_rows[b] (the last block: only the CLS / EOT row of its
    // output is used): dQ is computed for that row's 16-query block only (zero elsewhere) and dK / dV sum over that block's chunk
    const int* sel_rows = nullptr;
    bool sweep = false;         // bwd, non-causal: the single-sweep kernel (S / dP computed once, dS through LDS)
    bool force_fused = false;   // bwd: the fused single pass also where the dispatcher prefers the two kernels (NC > 3; A/B, tests)
    bool fused_w1 = false;      // bwd, fused form: NC waves with two 16-row blocks each instead of 2 NC waves with one (A/B)
    bool two_kernels = false;   // bwd: the dQ kernel + dK/dV kernel pair instead of the fused single pass (A/B runs, tests)
    // bwd, optional: only the dqkv rows win_row0 .. win_row0 + win_n - 1 of every sequence are wanted (block 0 of a tower: its input
    // gradient is needed on the prompt rows only).  The 16-row blocks that hold such a row are computed in full -- dQ from all keys,
    // dK / dV from all queries, same sums in the same order as without the window -- and every other row of dqkv is left UNWRITTEN.
    // Honoured by the two-kernel form (whole sequence on chip, non-causal) and the tiled kernels; elsewhere everything is computed.
    int win_row0 = 0, win_n = 0;
    bool tiled_fwd_16 = false;  // fwd, 224 < L <= 640: the staged 16-query-block kernel instead of the resident form (A/B runs, tests)
    int B = 0, L = 0, H = 0; bool causal = false;
    // exact-fp32 forward (attention_exact.hip): q | k | v in fp32 [B, L, 3*H*64] and, optionally, where to leave their T copy for the backward
    const float* qkv32 = nullptr;
    void* qkv_lp = nullptr;
};
// Single-query forms for the last block (attention_single.hip): ONE query row per sequence (token row a.sel_rows[b]) against all keys
// (causal: the first pos + 1).  q_sel / dout_sel / dq_sel are compact [B, H*64]; out_sel has row stride ld_out (optional low half out_lo);
// the backward writes the k and v thirds of a.dqkv for EVERY row (zeros behind a causal limit) and leaves its q third untouched.
int launch_attn_fwd_single(int dtype, const AttnArgs& a, const void* q_sel, void* out_sel, void* out_lo, int ld_out, float* lse_sel, hipStream_t s);  // out_lo in a.lo_mode
int launch_attn_bwd_single(int dtype, const AttnArgs& a, const void* q_sel, const void* out_sel, int ld_out, const void* dout_sel, const float* lse_sel,
                           void* dq_sel, hipStream_t s);
int attn_padded_len(int L);
int launch_attn_fwd(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof = nullptr);
int launch_attn_bwd(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof = nullptr);  // prof spans both kernels
// Exact-fp32 forward (the "fp32" mode): reads a.qkv32, writes a.out (fp16 hi) + a.out_lo (fp16 lo, optional) + a.lse and, if a.qkv_lp
// is set, the fp16 copy of q, k, v that the backward kernels read.
int launch_attn_fwd_exact(const AttnArgs& a, hipStream_t s, const LaunchProf* prof = nullptr);
// attention_resident.hip: both streamed operands of a (sequence, head) pair resident in LDS (224 < L, Lr * 256 (+ 8 Lr backward) <= 160 KB)
bool attn_resident_fits(int L, bool bwd);
int launch_attn_fwd_resident(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof = nullptr);
int launch_attn_bwd_resident(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof = nullptr);  // dQ (+ delta) kernel, then dK/dV kernel

// ------------------------------------------------------------------------------------------------
// Small / HBM-bound helpers
// ------------------------------------------------------------------------------------------------
// images fp32 [B,3,H,W] -> patches T [B*P, ldk], columns 3*p*p.. zero  (im2col of the stride-p conv, clip/model.py:527-529)
int launch_patchify(int dtype, const float* images, void* patches, int B, int image_size, int patch, int ldk, hipStream_t s);
// the same as a split operand: patches = hi, patches_lo = the remainder in lo_mode (common.h LoMode), both with rows of ldk * 2 bytes
int launch_patchify_split(int dtype, const float* images, void* patches, void* patches_lo, int lo_mode, int B, int image_size, int patch, int ldk, hipStream_t s);
// x[b, row0 + i, :] = rows[i, :] (+ add[i, :])  for i < n : CLS row, prompt rows, deep-prompt splice.
int launch_set_rows(float* x, int B, int L, int d, int row0, int n, const float* rows, const float* add, hipStream_t s);
// dst[r] = src[rows[r]] (gather) / dst[rows[r]] = src[r] (scatter): whole rows of row_bytes (multiple of 16), strides in bytes.
int launch_gather_rows(const void* src, size_t src_stride, const int* rows, void* dst, size_t dst_stride, int nrows, int row_bytes, hipStream_t s);
int launch_scatter_rows(const void* src, size_t src_stride, const int* rows, void* dst, size_t dst_stride, int nrows, int row_bytes, hipStream_t s);
// dst[rows[r], :] += src[r, :] for nrows rows of d elements of T (fp32 add, one rounding); rows must be distinct
int launch_add_rows(int dtype, const void* src, const int* rows, void* dst, int nrows, int d, hipStream_t s);
// out[i, :] = sum_b src[b, row0 + i, :] in fixed order (deterministic); optionally zero the source rows
// (fp32 and its T copy) afterwards: backward of the splice.  accumulate: out += instead of =.
// scale multiplies the sum (undoes the static loss scale of the backward pass).
int launch_reduce_rows(int dtype, float* src, void* src_lp, int B, int L, int d, int row0, int n, float* out,
                       bool zero_src, bool accumulate, float scale, hipStream_t s);
// fp32 C[M,N] = alpha * op(A) . op(B) (+ bias[N]) (+ beta * C); small shapes only (prompt projections, head).
int launch_sgemm(bool transA, bool transB, int M, int N, int K, float alpha, const float* A, int lda,
                 const float* B, int ldb, float beta, float* C, int ldc, const float* bias, hipStream_t s);
// out[n] (+)= sum_m A[m, n]
int launch_colsum(const float* A, int M, int N, int lda, float* out, bool accumulate, hipStream_t s);
// y = a + b (elementwise fp32)
int launch_add(const float* a, const float* b, float* y, size_t n, hipStream_t s);
// cast fp32 -> T
int launch_cast(int dtype, const float* x, void* y, size_t n, hipStream_t s);
// Cosine-logit head + mean cross-entropy (trainers/mudpt.py:178-182,250), fwd and bwd in one launch each.
struct HeadArgs {
    const float* img = nullptr;   // [B, e] raw image features
    const float* txt = nullptr;   // [C, e] raw text features
    const int64_t* labels = nullptr;  // [B] (bwd / loss only)
    float scale = 1.f;            // exp(logit_scale)
    float* logits = nullptr;      // [B, C]
    float* loss = nullptr;        // [1] mean CE over B (multiplied by loss_weight for the reported value? no: plain mean)
    float* dlogits = nullptr;     // [B, C] scratch
    float* row_loss = nullptr;    // [B] scratch
    float* dimg = nullptr;        // [B, e] grad of raw image features
    float* dtxt = nullptr;        // [C, e] grad of raw text features
    float* img_n = nullptr;       // [B, e] scratch: normalised features
    float* txt_n = nullptr;       // [C, e]
    float* img_inv = nullptr;     // [B] 1/||img||
    float* txt_inv = nullptr;     // [C]
    float grad_scale = 1.f;       // dlogits = (softmax - onehot) * grad_scale / B  (the caller folds loss scaling in here)
    int B = 0, C = 0, e = 0;
    int B_total = 0;              // pair head, chunked over the images: the batch the mean is taken over (0 = B); then the caller computes the loss (launch_mean)
};
int launch_head_fwd(const HeadArgs& a, hipStream_t s);
int launch_head_bwd(const HeadArgs& a, hipStream_t s);
// Fused form (head.hip): the logit contraction on the matrix cores in exact fp32 (v_mfma_f32_16x16x4_f32), cross-entropy and dlogits
// in the workgroup that made the logit rows.  a.txt == null: keep the normalised text features of the previous call.
bool head_fused_fits(const HeadArgs& a, bool train);
int launch_head_fused_fwd(const HeadArgs& a, hipStream_t s);
int launch_head_fused_train(const HeadArgs& a, hipStream_t s);  // logits + loss + dimg + dtxt (gradients of the raw features)
// CoCoOp (trainers/cocoop.py): per-image text features.  txt / txt_n / txt_inv / dtxt have B * C rows (row i * C + c).
int launch_pair_head_fwd(const HeadArgs& a, hipStream_t s);
int launch_pair_head_bwd(const HeadArgs& a, hipStream_t s);  // loss, dlogits, dtxt (gradient of the raw text features)
int launch_mean(const float* v, int n, float* out, hipStream_t s);  // out[0] = mean(v) in a fixed order
// text-tower input of every (image, class) pair: class prompt + positional embedding, context rows = ctx + meta_net(image) + pos
int launch_cocoop_prompts(float* x0, const float* emb_pos, const float* ctx, const float* bias, const float* pos, int B, int C, int L, int d, int n, hipStream_t s);
// d bias[i] = scale * sum over the classes and the n context rows of the text-input gradient (fp32 dx or its T copy)
int launch_cocoop_dbias(int dtype, const float* dx, const void* dx_lp, float* dbias, int B, int C, int L, int d, int n, float scale, hipStream_t s);
// y = x / ||x|| per row, inv = 1 / ||x||
int launch_l2norm(const float* x, float* y, float* inv, int rows, int e, hipStream_t s);
int launch_relu(float* y, size_t n, hipStream_t s);
int launch_relu_bwd(float* dy, const float* y, size_t n, hipStream_t s);
// Fused SGD (torch.optim.SGD semantics) over the flat bucket.
int launch_sgd(float* p, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay,
               float dampening, bool nesterov, bool first_step, hipStream_t s);

}  // namespace mudpt
