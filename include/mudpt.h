/*
 * mudpt.h -- C ABI of libmudpt_hip.so, the MI355X (gfx950) implementation of the MuDPT prompt-tuning
 * hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *   mudpt_create / mudpt_set_weight      clip/model.py:881-921 build_model + CLIP.__init__ :667-779
 *                                        (frozen weights ingested under OpenAI CLIP state-dict keys)
 *   mudpt_set_class_prompts              trainers/mudpt.py:83-95   token_prefix / token_suffix buffers,
 *                                        tokenized_prompts.argmax (EOT position, :154)
 *   mudpt_param_* / mudpt_bind_params    trainers/mudpt.py:205-218 the 10 trainable tensors (freeze rule)
 *   mudpt_forward                        trainers/mudpt.py:170-184 CustomCLIP.forward == model_inference()
 *   mudpt_forward_backward               trainers/mudpt.py:249-251 forward, F.cross_entropy, backward
 *   mudpt_sgd_step                       trainers/mudpt.py:251     model_backward_and_update's optimizer step
 *   mudpt_allreduce_grads                trainers/mudpt.py:230-233 nn.DataParallel's gradient reduce (here: one RCCL all-reduce)
 *   mudpt_set_class_shard / mudpt_cp_*   trainers/mudpt.py:142-156,178-182 the same step with the class prompts divided over the ranks
 *   mudpt_gemm / _layernorm_* / _attention_*   the ATen ops under clip/model.py:164-175,257-301 (unit parity)
 * With mudpt_config.variant = MUDPT_VARIANT_COCOOP the same entry points run the CoCoOp path (trainers/cocoop.py):
 *   mudpt_create / mudpt_set_weight      trainers/cocoop.py:22-40  load_clip_to_cpu: vanilla CLIP (clip/model.py:443-496 ViT)
 *   mudpt_set_class_prompts              trainers/cocoop.py:113-122 token_prefix / token_suffix, tokenized_prompts
 *   mudpt_param_*                        trainers/cocoop.py:96-107,222-226  ctx + meta_net.linear1/2 (5 tensors)
 *   mudpt_forward                        trainers/cocoop.py:178-198 CustomCLIP.forward in eval mode (logits [B, C])
 *   mudpt_forward_backward               trainers/cocoop.py:196-197,258-261 cross-entropy inside forward + backward
 *
 * Conventions: every function returns 0 on success or a MUDPT_ERR_* code; mudpt_last_error() gives
 * the message of the calling thread's last failure.  No exceptions cross the ABI.  A model handle is
 * not re-entrant.  Device pointers are HIP device memory owned by the caller unless stated; `stream`
 * is a hipStream_t passed as void* (NULL = default stream).  All launches are asynchronous.
 */
#ifndef MUDPT_H
#define MUDPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUDPT_ABI_VERSION 6

#define MUDPT_OK 0
#define MUDPT_ERR_ARG 1   /* bad argument / shape (the reference raises AssertionError, mudpt.py:52,55,190) */
#define MUDPT_ERR_HIP 2   /* a HIP runtime call failed */
#define MUDPT_ERR_STATE 3 /* call order: weights / prompts / parameters not set yet */

#define MUDPT_BF16 0
#define MUDPT_F16 1
/* The PARITY mode, what PREC = "fp32" selects (the reference's CPU path is fp32 whatever PREC says: clip/clip.py:142-143): the mode that
 * holds north_star's bound -- logits within 1e-3 of the reference -- at the logit scale pretrained checkpoints carry (exp(logit_scale) =
 * 100, trainers/mudpt.py:181).  fp16 MFMA operands with SPLIT forward GEMM operands (hi = fp16(v) plus the remainder, contracted in a
 * second pass against the fp16-exact weights, as CLIP checkpoints store them):
 *   text tower    remainder as fp16 (22 bits) + attention forward in fp32 on the matrix cores (v_mfma_f32_16x16x4_f32): each of its
 *                 rounding sites alone moves the logits by 2.5e-3;
 *   vision tower  remainder as e4m3, contracted against an e4m3 copy of the weights on the MX-scaled fp8 matrix instruction
 *                 (v_mfma_scale_f32_16x16x128_f8f6f4: twice the fp16 rate, so the second pass costs half of the first), pixels included;
 *                 attention in fp16.
 * Measured against the reference's own logits at scale 100: <= 3.5e-4 (ViT-B/16; 1.2e-4 ViT-L/14@336), 1.26x the bf16 step.  Knobs
 * (mudpt_model_set): vis_exact_attn = 1 -> 8e-5; vis_lo = 1 + vis_exact_attn = 1 -> round 3's "exact" mode, 2e-5 at 1.6x.  The
 * backward is the MUDPT_F16 one with fp16 activation gradients.  Per-site ablation: DESIGN.md 2. */
#define MUDPT_F32 2

#define MUDPT_VARIANT_MUDPT 0  /* trainers/mudpt.py: deep multi-modal prompts, 10 trainables */
#define MUDPT_VARIANT_COCOOP 1 /* trainers/cocoop.py: instance-conditioned text prompts, 5 trainables; depth is ignored */

/* Model shape.  ViT-B/16 MuDPT: {224,16,768,12,12, 512,12,8,77, 512, 4,12, n_cls, max_batch, dtype, 0}. */
typedef struct mudpt_config {
    int32_t image_size, patch, v_width, v_layers, v_heads;
    int32_t t_width, t_layers, t_heads, ctx_len;
    int32_t embed_dim;
    int32_t n_ctx, depth; /* TRAINER.MUDPT.N_CTX / DEEP_PROMPT_DEPTH (train.py:115-119) */
    int32_t n_cls;        /* number of class prompts */
    int32_t max_batch;    /* activations are sized for this many images */
    int32_t dtype;        /* MUDPT_BF16 / MUDPT_F16: MFMA operand type (fp32 accumulate, fp32 residual stream); MUDPT_F32: the parity mode */
    int32_t variant;      /* MUDPT_VARIANT_*; CoCoOp runs max_batch * n_cls text sequences per step */
} mudpt_config;

typedef struct mudpt_model mudpt_model;

int mudpt_abi_version(void);
const char* mudpt_last_error(void);

int mudpt_create(const mudpt_config* cfg, mudpt_model** out);
int mudpt_destroy(mudpt_model* m);

/* Frozen weight by OpenAI CLIP state-dict key ("visual.transformer.resblocks.0.attn.in_proj_weight", ...),
 * fp32 HOST data in the checkpoint's own layout; the library converts / transposes to its device layout.
 * Keys the path does not use ("token_embedding.weight", ...) are accepted and ignored. */
int mudpt_set_weight(mudpt_model* m, const char* key, const float* host_data, size_t numel);

/* token_embedding(tokenized "<ctx words> <classname>.") [n_cls, ctx_len, t_width] fp32 HOST and the EOT
 * position of every class prompt; rows 1..n_ctx are replaced by the trainable ctx at run time.  The text tower then runs on
 * positions 0..max(eot_index) only: under the causal mask (clip/model.py:407-413) later positions reach neither the EOT
 * feature (trainers/mudpt.py:154) nor any gradient. */
int mudpt_set_class_prompts(mudpt_model* m, const float* embedding, const int32_t* eot_index);
/* What the text tower runs per pass after mudpt_set_class_prompts: token rows, length buckets, longest kept length.  With many classes
 * (>= 2048 rows) the prompts are sorted by length and run in up to "txt_buckets" (mudpt_model_set, default 3) groups, each to its own
 * longest EOT, instead of all to the overall longest: the kept rows are bit-identical, the padding rows are not computed. */
int mudpt_text_layout(const mudpt_model* m, int32_t* rows, int32_t* buckets, int32_t* max_len);

/* The 10 trainable tensors live in ONE flat fp32 bucket (= the data-parallel all-reduce payload). */
int mudpt_param_count(const mudpt_model* m);   /* 10 (MuDPT) or 5 (CoCoOp) */
size_t mudpt_param_numel(const mudpt_model* m); /* elements of the flat bucket */
/* name = the reference's CustomCLIP state-dict key; shape has ndim entries (ndim <= 3). */
int mudpt_param_info(const mudpt_model* m, int index, const char** name, size_t* offset, size_t* numel,
                     int32_t* ndim, int64_t shape[3]);
/* Device pointers to the flat parameter bucket and the flat gradient bucket (caller-owned, fp32). */
int mudpt_bind_params(mudpt_model* m, float* params_dev, float* grads_dev);

/* logits[B, n_cls] fp32 for images[B,3,S,S] fp32 (CLIP-normalised pixels), both device memory. */
int mudpt_forward(mudpt_model* m, const float* images_dev, int32_t batch, float* logits_dev, void* stream);

/* Same with flags.  MUDPT_FWD_REUSE_TEXT: keep the text features of the previous call (valid while the bound parameters
 * are unchanged): the reference recomputes the text tower for every test batch (trainers/mudpt.py:170-184). */
#define MUDPT_FWD_REUSE_TEXT 1
/* mudpt_cp_forward only: this forward is the first phase of a TRAINING step (mudpt_forward_backward implies it).  A training forward may
 * split the contraction of the vision tower's out_proj / c_proj at tiny batches (<= 8 ViT-B images; a differently associated fp32 sum);
 * an inference forward never does, so the logits of an image do not depend on the size of the test batch it arrives in. */
#define MUDPT_FWD_TRAINING 2
int mudpt_forward_ex(mudpt_model* m, const float* images_dev, int32_t batch, float* logits_dev, int32_t flags, void* stream);

/* One training step's forward + backward: loss_dev[0] = mean cross-entropy over the batch, gradients of
 * (grad_scale * loss) written to the bound gradient bucket.  logits_dev may be NULL.  grad_scale = 1/world
 * makes the sum over data-parallel ranks the gradient of the global-batch mean. */
int mudpt_forward_backward(mudpt_model* m, const float* images_dev, const int64_t* labels_dev, int32_t batch,
                           float grad_scale, float* loss_dev, float* logits_dev, void* stream);

/* Data parallelism (replaces nn.DataParallel, trainers/mudpt.py:230-233): one process per GPU, each computing the gradient of
 * (1/world) * local-mean loss (grad_scale above); the ONE exchange of a step is the sum of the flat gradient bucket over the ranks.
 * A host with torch.distributed does that itself (mudpt_amd/parallel.py: dist.all_reduce on the bound bucket); any other host
 * passes its RCCL communicator (ncclComm_t as void*, created with ncclCommInitRank) here.  In place on the bound bucket, asynchronous
 * on `stream`; RCCL is resolved from the process at first use (no link-time dependency). */
int mudpt_allreduce_grads(mudpt_model* m, void* nccl_comm, void* stream);

/* Class-parallel text tower: the second axis for many classes (ImageNet, C = 1000).  The reference replicates the text tower over all C
 * class prompts on every GPU (trainers/mudpt.py:142-156 inside nn.DataParallel, :230-233); here rank r may encode classes [c0, c1) only.
 * mudpt_set_class_shard comes before mudpt_set_class_prompts (which still receives ALL n_cls prompts on every rank and keeps its own).
 * A sharded handle refuses mudpt_forward / mudpt_forward_backward; one step is
 *   mudpt_cp_forward(images)        both towers; rows c0..c1-1 of feat[n_cls, embed] = this rank's text features, other rows zero
 *   exchange 1                      all-reduce(sum) -- or all-gather -- of feat over the ranks
 *   mudpt_cp_head(labels, ...)      logits / loss of the LOCAL images against ALL classes; backward -> dfeat[n_cls, embed]
 *                                   (labels NULL: inference, logits only, no exchange 2 / backward)
 *   exchange 2                      all-reduce(sum) of dfeat (every rank's images contribute to every class); it may overlap
 *   mudpt_cp_backward(MUDPT_CP_VISION)   ... the vision tower's backward, which does not read dfeat
 *   mudpt_cp_backward(MUDPT_CP_TEXT)     text tower backward over the local classes + prompt-learner backward, after exchange 2
 *   the gradient-bucket all-reduce  as in pure data parallelism (text-side gradients are partial sums over classes)
 * mudpt_cp_buffers returns the two fp32 device tables (library-owned, numel = n_cls * embed_dim).  Every phase is asynchronous on `stream`;
 * the caller orders its exchanges on that stream.  An unsharded handle may run the phases too (one rank: no exchange needed). */
#define MUDPT_CP_VISION 1
#define MUDPT_CP_TEXT 2
int mudpt_set_class_shard(mudpt_model* m, int32_t class_begin, int32_t class_end);
int mudpt_cp_buffers(mudpt_model* m, float** feat_dev, float** dfeat_dev, size_t* numel);
int mudpt_cp_forward(mudpt_model* m, const float* images_dev, int32_t batch, int32_t flags, void* stream);
int mudpt_cp_head(mudpt_model* m, const int64_t* labels_dev, int32_t batch, float grad_scale, float* loss_dev, float* logits_dev,
                  int32_t flags, void* stream);
int mudpt_cp_backward(mudpt_model* m, int32_t part, void* stream);

/* Static loss scale of the backward pass (default 128): per-sample logit gradients are multiplied by it so the
 * fp16 copies of the token gradients stay normal; the gradients written to the bucket are unscaled again.
 * The reference's analogue is GradScaler under PREC == "amp" (trainers/mudpt.py:228,243-246). */
int mudpt_set_loss_scale(mudpt_model* m, float loss_scale);

/* torch.optim.SGD update of the bound parameters from the bound gradients (momentum buffer library-owned). */
int mudpt_sgd_step(mudpt_model* m, float lr, float momentum, float weight_decay, float dampening,
                   int32_t nesterov, void* stream);
int mudpt_sgd_reset(mudpt_model* m);

/* ======================================================================================================================================
 * Everything below this line is TEST / MEASUREMENT / DEBUG surface, not part of the drop-in boundary: a trainer plugin needs none of it.
 * ====================================================================================================================================== */

/* Test hook: copy an internal fp32 activation of the last call to HOST memory (synchronises the device).
 * name: "vis.x_in.<i>" / "txt.x_in.<i>" (input of block i, after the prompt splice; [seq, L, d], text L = max(eot) + 1), "vis.x_out" / "txt.x_out"
 * (output of the last block on the ONE row per sequence the model uses -- CLS / EOT token -- [seq, d]: the tail of the last
 * block runs on those rows only), "image_features", "text_features".  host_out may be NULL to query *numel. */
int mudpt_debug_read(mudpt_model* m, const char* name, int32_t batch, float* host_out, size_t capacity, size_t* numel);

/* Debug knobs of ONE handle, for A/B measurements in one process and for tests (tools/gemm_bench.py, bench.py flags, tests/).  Nothing is
 * process-global: two models in one process do not interfere.  Defaults are what the product runs; no knob is needed for correct results.
 *
 *   knob               default         meaning
 *   gemm_variant       0               kernel choice of the MFMA GEMMs (gemm.hip launch_epi / gemm_pp.hip; 0 = the default dispatch; 12 = the default
 *                                      without the 128-deep K-tiles of the small grids; other values force one tile form: A/B runs)
 *   split_k            1               0 = never split the contraction of the vision tower's small-grid, long-K store GEMMs
 *   fwd_split_k        1               0 = ... of the forward ones (out_proj / c_proj up to 320 tiles of 64 x 64, i.e. <= 8 images of ViT-B): logits
 *                                      of a batch then equal the logits of its chunks bit for bit at EVERY chunk size (default: above that size)
 *   attn_window        1               0 = block 0's attention backward on all rows instead of the prompt rows' blocks
 *   last_single        1 (0 exact)     0 = the last block's attention on all rows instead of the single-query form
 *   attn_two_kernels   0               1 = attention backward as the dQ + dK/dV kernel pair
 *   attn_fused_w1      0               1 = fused attention backward with two 16-row blocks per wave
 *   lp_grad            bf16: 1, else 0 gradient stream of the residual in T (bf16 mode: 0 also returns the forward's update stream to fp32;
 *                                      fp16 mode: 1 trades 30 % more gradient error for 0.9 ms)
 *   lp_upd             bf16: 1, else 0 the forward's update stream in T (fp16 mode: would cost 2e-4 of logit error)
 *   txt_split          fp16 / fp32: 1  0 = no split operands in the text tower (fp16 mode; only before the first mudpt_set_weight)
 *   txt_trim           1               0 = run the text tower on all ctx_len positions        } read by the next
 *   txt_buckets        3               maximum number of length buckets of the class prompts  } mudpt_set_class_prompts,
 *   txt_bucket_cost    1024            token rows one more bucket must save                   } which must follow
 *   cocoop_chunk       0 (= budget)    cap on the images per CoCoOp text-tower pass           }
 *   prof_stride        1               measurement mode brackets every prof_stride-th persistent-GEMM launch, counted across steps */
int mudpt_model_set(mudpt_model* m, const char* name, int32_t value);

/* Measurement hook (bench.py): bracket every MFMA GEMM launch of the path with HIP events on its launch stream.
 * mudpt_profile_read synchronises and returns the summed duration, the summed algorithmic FLOPs (2 M N K) and
 * the number of launches since the last enable / read. */
/* enable: 0 = off; 1 = the persistent MFMA GEMM launches only (class 0 below); otherwise a bit mask of the classes of
 * mudpt_profile_read_classes (31 = all).  An event pair costs ~5 us of queue time per bracketed launch (measured: 157 pairs = 0.8 ms
 * per step), so the timed region of bench.py brackets the dominant kernel only and the other classes are measured in a separate pass. */
int mudpt_profile_enable(mudpt_model* m, int32_t enable);
int mudpt_profile_read(mudpt_model* m, double* gemm_ms, double* gemm_flop, int64_t* launches);
/* The same per kernel class (arrays of MUDPT_PROF_CLASSES entries): 0 the persistent MFMA GEMM (work = algorithmic FLOPs), 1 / 2
 * LayerNorm forward / backward, 3 / 4 attention forward / backward (dQ + dK/dV kernels together) of the vision tower (work =
 * algorithmic HBM bytes: every operand read once, every result written once).  executed_flop (may be NULL): MFMA FLOPs actually
 * executed by all GEMM and attention launches of both towers. */
#define MUDPT_PROF_CLASSES 5
int mudpt_profile_read_classes(mudpt_model* m, double* ms, double* work, int64_t* launches, double* executed_flop);

/* ---- single kernels, exported for parity tests (all pointers device memory) ---------------------------- */
/* epilogues: 0 store T | 1 bias+QuickGELU (out0 = u, out1 = gelu(u)) | 2 f32 out0 = aux + acc + bias |
 *            3 out0 = acc * QuickGELU'(aux) | 4 patch-embed scatter + pos | 5 store f32 */
int mudpt_gemm(int32_t dtype, int32_t epilogue, int32_t M, int32_t N, int32_t K, const void* A, int32_t lda,
               const void* B, int32_t ldb, const float* bias, void* out0, int32_t ldo0, void* out1, int32_t ldo1,
               const void* aux, int32_t ldaux, int32_t patches, int32_t seq_len, const float* pos, int32_t variant, void* stream);
/* variant: kernel-choice knob for tests / tuning (0 = the default dispatch).  Bit 16: allow split K.  Bit 17: QuickGELU' in 8 bits (what the
 * bf16 mode keeps for the backward instead of u): epilogue 1 writes out0 = byte codes rint((QuickGELU'(u) + 0.1) * 212) with a row stride of
 * ldo0 BYTES, epilogue 3 reads such codes from aux (row stride ldaux bytes). */
/* A GEMM with a SPLIT A operand (the forward GEMMs of the parity mode): A = T(v), A_lo = the remainder v - A in a second buffer
 * with the row stride of A in bytes.  lo_mode 1: A_lo holds T values and a second pass contracts it against the same B (22 bits); lo_mode 2:
 * A_lo holds OCP e4m3 bytes of (v - A) * 2^12 (the first K bytes of each row) and the second pass runs on the MX-scaled fp8 matrix
 * instruction against B8, the e4m3 copy of B ([N, K] bytes at the row stride of B in bytes, values B * 2^shift, b8_scale = 127 - shift =
 * the E8M0 block scale); K % 128 == 0 and dtype fp16 then.  lo_mode 0: no second pass.  epilogue 0 | 1 | 2 | 5 as mudpt_gemm; with
 * epilogue 1, out1_lo (may be NULL) receives the low half of out1 = QuickGELU(u) in form out1_lo_mode (1 / 2), at the row stride of out1 in bytes. */
int mudpt_gemm_split(int32_t dtype, int32_t epilogue, int32_t M, int32_t N, int32_t K, const void* A, const void* A_lo, int32_t lo_mode, int32_t lda,
                     const void* B, const void* B8, int32_t b8_scale, int32_t ldb, const float* bias, void* out0, int32_t ldo0, void* out1,
                     void* out1_lo, int32_t out1_lo_mode, int32_t ldo1, const void* aux, int32_t ldaux, int32_t variant, void* stream);
/* HOST helper (no device work): out[i] = OCP e4m3 (round to nearest even, saturating at +-448) of in[i] * 2^shift -- the conversion the
 * library applies to the frozen weights for the e4m3 second pass. */
int mudpt_e4m3_from_f32(const float* in_host, uint8_t* out_host, size_t n, int32_t shift);
/* LayerNorm forward writing a split operand: out = T(y), out_lo = the remainder in form lo_mode (1: T, 2: e4m3 bytes of (y - out) * 2^12),
 * rows of ldo elements of T / 2 ldo bytes. */
int mudpt_layernorm_fwd_split(int32_t dtype, const float* x, int32_t ldx, const float* gamma, const float* beta, void* out, void* out_lo,
                              int32_t lo_mode, int32_t ldo, int32_t rows, int32_t d, void* stream);
int mudpt_layernorm_fwd(int32_t dtype, const float* x, int32_t ldx, const int32_t* row_index, const float* gamma,
                        const float* beta, void* out, int32_t ldo, int32_t out_f32, float* mean, float* rstd,
                        int32_t rows, int32_t d, void* stream);
int mudpt_layernorm_bwd(int32_t dtype, const void* dy, int32_t lddy, int32_t dy_f32, const float* x, int32_t ldx,
                        const int32_t* row_index, const float* mean, const float* rstd, const float* gamma,
                        const float* dres, int32_t lddres, float* dx, int32_t lddx, void* dx_lp, int32_t lddx_lp,
                        int32_t rows, int32_t d, void* stream);
int mudpt_attention_padded_len(int32_t L);
/* causal: bit 0 = causal mask; bit 1 (tests / A-B, 224 < L <= 640 only) = the staged 16-query-block kernel instead of the resident one. */
int mudpt_attention_fwd(int32_t dtype, const void* qkv, void* out, float* lse, int32_t B, int32_t L, int32_t H,
                        int32_t causal, void* stream);
/* The fp32 attention forward (parity mode MUDPT_F32: the text tower, knob for the vision tower): q | k | v in fp32 [B, L, 3*H*64] -> the
 * output as a split operand out_hi (fp16) + out_lo (may be NULL; form lo_mode 1 / 2 as in mudpt_gemm_split; row stride ld_out elements of
 * fp16 = 2 ld_out bytes, 0 = H*64), lse, and -- if qkv_lp is not NULL -- the fp16 copy of q | k | v the backward kernels read. */
int mudpt_attention_fwd_exact(const float* qkv32, void* qkv_lp, void* out_hi, void* out_lo, int32_t lo_mode, int32_t ld_out, float* lse, int32_t B,
                              int32_t L, int32_t H, int32_t causal, void* stream);
/* causal: bit 0 = causal mask.  Kernel choice (tests / A-B).  Default: padded length <= 96 (the text tower): the fused two-sweep pass over
 * resident Q, K, V, dO; longer non-causal sequences up to 224 (the vision tower): the single-sweep kernel (S, dP, exp computed once, dS
 * crosses LDS for dQ); otherwise a dQ kernel + dK/dV kernel pair (delta through `delta`): for L > 224 the resident pair while both operands of a
 * (sequence, head) fit LDS (L <= 608), else (and for the window form, and with bit 1) the staged pair.  bit 1 = force the (staged) two kernels, bit 3 = force the
 * fused two-sweep pass, bit 2 = the same with two 16-row blocks per wave, bit 4 = force the single sweep (non-causal, L <= 224).
 * Window form (block 0 of a tower needs its input gradient on the prompt rows only): bits 20-27 = n > 0 wanted rows per sequence starting
 * at row bits 8-19.  The 16-row blocks (L > 224: 128-row groups) holding a wanted row are computed exactly as without the window; all other
 * rows of dqkv are left unwritten. */
int mudpt_attention_bwd(int32_t dtype, const void* qkv, const void* out, const void* dout, const float* lse,
                        float* delta, void* dqkv, int32_t B, int32_t L, int32_t H, int32_t causal, void* stream);
/* Single-query attention of a tower's LAST block (only the CLS / EOT row of its output is used, clip/model.py:549, trainers/mudpt.py:154):
 * one query per sequence -- q_sel [B, H*64], the query of token row sel_rows[b] (= b * L + position) -- against the K / V thirds of the
 * packed qkv buffer (causal: keys 0 .. position).  Forward: out_sel [B, H*64], lse_sel [B, H].  Backward: dq_sel [B, H*64] and the k, v
 * thirds of dqkv for EVERY row (zeros behind a causal limit); the q third of dqkv is not written. */
int mudpt_attention_fwd_single(int32_t dtype, const void* qkv, const void* q_sel, const int32_t* sel_rows, void* out_sel, float* lse_sel,
                               int32_t B, int32_t L, int32_t H, int32_t causal, void* stream);
int mudpt_attention_bwd_single(int32_t dtype, const void* qkv, const void* q_sel, const int32_t* sel_rows, const void* out_sel,
                               const void* dout_sel, const float* lse_sel, void* dqkv, void* dq_sel, int32_t B, int32_t L, int32_t H,
                               int32_t causal, void* stream);
/* LayerNorm forward with everything the transformer block fuses into it (clip/model.py:281-301): v = x[r] + add[r] (fp32 add or T
 * add_lp, either may be NULL; row stride ldadd); rows whose position (r % ov_L) lies in [ov_row0, ov_row0 + ov_n) are REPLACED by
 * ov_rows[(r % ov_L) - ov_row0] (the deep-prompt splice); v is written to xout (fp32, may be NULL) and normalised into out (T or fp32). */
int mudpt_layernorm_fwd_fused(int32_t dtype, const float* x, int32_t ldx, const float* add, const void* add_lp, int32_t ldadd,
                              const float* ov_rows, int32_t ov_row0, int32_t ov_n, int32_t ov_L, float* xout, int32_t ldxout,
                              const float* gamma, const float* beta, void* out, int32_t ldo, int32_t out_f32, float* mean,
                              float* rstd, int32_t rows, int32_t d, void* stream);
/* Cosine-logit head + mean cross-entropy, forward and backward (trainers/mudpt.py:178-182,250): logits[B, C] = scale *
 * normalise(img) . normalise(txt)^T, loss[0] = mean CE, dimg / dtxt = gradients of (grad_scale * loss) w.r.t. the RAW features.
 * labels / loss / dimg / dtxt may be NULL (forward only).  Scratch is allocated and freed inside (synchronises: a test hook). */
int mudpt_head(const float* img, const float* txt, const int64_t* labels, float scale, float grad_scale, int32_t B, int32_t C,
               int32_t e, float* logits, float* loss, float* dimg, float* dtxt, void* stream);
/* out[i, :] (+)= scale * sum_b src[b, row0 + i, :], b ascending in a fixed tree (bitwise reproducible): the backward of the prompt
 * splice.  src_f32 [B, L, d] or its T copy src_lp (at least one non-NULL); zero_src clears the summed rows afterwards. */
int mudpt_reduce_rows(int32_t dtype, float* src_f32, void* src_lp, int32_t B, int32_t L, int32_t d, int32_t row0, int32_t n,
                      float* out, int32_t zero_src, int32_t accumulate, float scale, void* stream);
/* CoCoOp (trainers/cocoop.py:141-146,187-194): dbias[i, :] = scale * sum over image i's C prompts and their n context rows (rows
 * 1..n of every L-row sequence) of the text-input gradient dx [B * C, L, d] (fp32, or its T copy dx_lp); fixed order: reproducible. */
int mudpt_cocoop_dbias(int32_t dtype, const float* dx_f32, const void* dx_lp, float* dbias, int32_t B, int32_t C, int32_t L, int32_t d,
                       int32_t n, float scale, void* stream);
/* fp32 C[M,N] = alpha * op(A) . op(B) (+ bias[N]) (+ beta * C): the prompt projections (trainers/mudpt.py:127-128, clip/model.py:539). */
int mudpt_sgemm(int32_t transA, int32_t transB, int32_t M, int32_t N, int32_t K, float alpha, const float* A, int32_t lda,
                const float* B, int32_t ldb, float beta, float* C, int32_t ldc, const float* bias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUDPT_H */
