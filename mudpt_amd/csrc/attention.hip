// Attention forward / backward for gfx950, head dim 64, whole sequence of one (sequence, head) on chip.
//
// Replaces nn.MultiheadAttention's per-head softmax(Q K^T / 8 [+ causal mask]) V and its autograd
// (clip/model.py:271-273; mask clip/model.py:810-816).  Sequence lengths on this path are short and fixed
// (vision 197 + n_ctx = 201, text 77; SURVEY.md §5 "long-context: absent"), so one workgroup holds the two
// operand matrices of a (sequence, head) pair in LDS and every wave owns 16-row blocks; the deep-prompt rows are
// ordinary rows of the same tile.  L is padded to 32 * NC inside the kernels only (masked keys, unstored rows).
//
// v_mfma_f32_16x16x32 everywhere (A: row = lane & 15, k = 8 (lane >> 4) + j; B: k likewise, col = lane & 15;
// C/D: col = lane & 15, row = 4 (lane >> 4) + r).  With 16-row blocks a block's whole score column set is 14 x 4
// registers, so a wave needs ~100 VGPRs and two workgroups (14 waves) share a CU and hide each other's latency.
//
// LDS holds ONLY row-major images [rows][64] with a 160-byte row stride: conflict-free both for ds_read_b128 row
// fragments and for ds_read_b64_tr_b16, the hardware-transposed read that produces the operand of the products
// contracting over the image's ROW index (P.V over keys, dS^T.K over keys, dO^T.P and Q^T.dS over queries):
// no transposed copy is ever staged.
//   fwd / dq pass -- "query on the lane":  S^T = K Q^T (A = K rows, B = Q fragment from HBM); a query's scores are
//     in 4 lanes (lane, ^16, ^32, ^48), softmax needs two xor-shuffles; the converted accumulators of two
//     consecutive 16-key tiles are directly the B operand of O^T = V^T P^T / dQ^T = K^T dS^T (the contraction index
//     is permuted identically in the transposed read: element j of lane group g is key 32 kc + (j < 4 ? 4 g + j :
//     16 + 4 g + j - 4)).
//   dk/dv pass -- "key on the lane": S = Q K^T, dP = dO V^T (B = K / V fragments in registers for the whole pass),
//     dV^T += dO^T P, dK^T += Q^T dS.
// dQ comes from its own sweep instead of atomics or a cross-wave reduction: bitwise reproducible, at the price of
// recomputing S and dP once (attention is < 10 % of the step's FLOPs).
#include "kernels.h"

namespace mudpt {

constexpr int RS = 64;  // LDS row stride in elements: dense 128-byte rows; 16-byte chunk c of row r sits in slot c ^ (r & 7)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float SC = 0.125f * LOG2E;  // 1 / sqrt(64) folded into the base-2 exponent

typedef __attribute__((ext_vector_type(4))) short s16x4;
using lds_s16x4 = __attribute__((address_space(3))) s16x4;

int attn_padded_len(int L) { return (L + 31) / 32 * 32; }
__device__ inline int attn_padded_len_dev(int L) { return (L + 31) / 32 * 32; }

template <typename T>
struct Attn {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;

    // row-major image img[Lp][RS] of src[L][64] (row stride ld elements); rows >= L are zero
    __device__ static inline void stage(elem* img, const elem* src, size_t ld, int L, int Lp, int tid, int nthreads) {
        for (int idx = tid; idx < Lp * 8; idx += nthreads) {
            const int row = idx >> 3, ch = idx & 7;
            vec8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (elem)0.f;
            if (row < L) v = *(const vec8*)(src + (size_t)row * ld + ch * 8);
            *(vec8*)(img + row * RS + ((ch ^ (row & 7)) << 3)) = v;
        }
    }
    // Two images at once, ALL global loads issued before the first LDS store (one HBM round trip per workgroup instead of
    // one per chunk).  CH = chunks per thread per image = Lp * 8 / nthreads (exact for nthreads = 2 * Lp).
    template <int CH>
    __device__ static inline void stage2(elem* img0, const elem* src0, size_t ld0, elem* img1, const elem* src1, size_t ld1, int L,
                                         int tid, int nthreads) {
        vec8 v0[CH], v1[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int idx = tid + k * nthreads, row = idx >> 3, ch = idx & 7;
#pragma unroll
            for (int i = 0; i < 8; ++i) { v0[k][i] = (elem)0.f; v1[k][i] = (elem)0.f; }
            if (row < L) {
                v0[k] = *(const vec8*)(src0 + (size_t)row * ld0 + ch * 8);
                v1[k] = *(const vec8*)(src1 + (size_t)row * ld1 + ch * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int idx = tid + k * nthreads, row = idx >> 3, ch = idx & 7;
            *(vec8*)(img0 + row * RS + ((ch ^ (row & 7)) << 3)) = v0[k];
            *(vec8*)(img1 + row * RS + ((ch ^ (row & 7)) << 3)) = v1[k];
        }
    }
    // 16 x 32 row fragment (A or B operand whose k runs along the image's columns): rows row0 + (lane & 15)
    __device__ static inline vec8 rows(const elem* img, int row0, int ks, int lane) {
        const int r = row0 + (lane & 15);
        return *(const vec8*)(img + r * RS + (((ks * 4 + (lane >> 4)) ^ (r & 7)) << 3));
    }
    // operand whose k runs along the image's ROWS (32 rows row0 .. row0 + 31) and whose row/col index is the image
    // column col0 + (lane & 15): two hardware-transposed 4 x 16 block reads; element j <-> image row
    // row0 + (j < 4 ? 4 g + j : 16 + 4 g + j - 4), g = lane >> 4 (the order pack2() produces).
    __device__ static inline vec8 cols(const elem* img, int row0, int col0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const int rr = row0 + 4 * g + (i >> 2);  // row0 is a multiple of 16, so row rr + 16 of the second read has the same swizzle
        const elem* p = img + rr * RS + (((((col0 >> 3) + ((i & 3) >> 1)) ^ (rr & 7)) << 3) + 4 * (i & 1));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * RS));
        // plain concatenation of the two 64-bit results into the 128-bit MFMA operand: no element moves
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        return __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
    // accumulators of two consecutive 16-row tiles -> the 32-deep operand contracting over those rows
    __device__ static inline vec8 pack2(const f32x4& x0, const f32x4& x1) {
        vec8 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) { r[k] = (elem)x0[k]; r[4 + k] = (elem)x1[k]; }
        return r;
    }
    // this lane's 16-byte piece of row `row` of a [.., 64]-wide global matrix for k-step ks (zero beyond L)
    __device__ static inline vec8 grow(const elem* src, size_t ld, int row, int L, int ks, int lane) {
        vec8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (elem)0.f;
        if (row < L) v = *(const vec8*)(src + (size_t)row * ld + ks * 32 + (lane >> 4) * 8);
        return v;
    }
    // the same, plus the rounding remainder to a second row (the low half of a split operand, common.h LoMode): lo_row points at the
    // row's 64 columns of this head in the low buffer -- elements of T (LO_F16) or e4m3 bytes (LO_F8)
    __device__ static inline void store_t_split(elem* dst_row, void* lo_row, int lo_mode, const f32x4 (&x)[4], float scale, int lane) {
        const int g = lane >> 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            vec4 v, l;
            float rem[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                elem hv;
                rem[r] = split_rem(x[dt][r] * scale, hv);
                v[r] = hv;
                l[r] = (elem)rem[r];
            }
            *(vec4*)(dst_row + 16 * dt + 4 * g) = v;
            if (lo_mode == LO_F8) *(uint32_t*)((char*)lo_row + 16 * dt + 4 * g) = pack_lo8(rem[0], rem[1], rem[2], rem[3]);
            else *(vec4*)((elem*)lo_row + 16 * dt + 4 * g) = l;
        }
    }
    // address of row `row`, head `hd` in the low buffer of a split output (same row stride in bytes as the T buffer: ldo elements of T)
    __device__ static inline void* lo_ptr(void* base, int lo_mode, size_t row, size_t ldo, int hd) {
        return lo_mode == LO_F8 ? (void*)((char*)base + row * ldo * 2 + hd * 64) : (void*)((elem*)base + row * ldo + hd * 64);
    }
    // D^T tile set (4 tiles of 16 d x 16 rows) -> dst[row][16 dt + 4 g + r]
    __device__ static inline void store_t(elem* dst_row, const f32x4 (&x)[4], float scale, int lane) {
        const int g = lane >> 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            vec4 v = {(elem)(x[dt][0] * scale), (elem)(x[dt][1] * scale), (elem)(x[dt][2] * scale), (elem)(x[dt][3] * scale)};
            *(vec4*)(dst_row + 16 * dt + 4 * g) = v;
        }
    }
};

__device__ inline float group_max(float v) {  // over the 4 lanes that share lane & 15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ inline float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// One 16-query block of one (sequence, head) pair in two halves: scores + softmax numerators (fwd_scores), P.V + store (fwd_pv_store).
// q0 / q1 = this lane's Q fragments.  S holds the scores, then the UNNORMALISED probabilities exp(s - m); m = row maximum (raw scores), l = row sum.
// Key tiles that lie wholly beyond L are skipped (L = 201: tile 13 of 14; P = 0 there), and only a tile that straddles L reads the key mask
// (round 4: the 14 mask reads were a quarter of this half's LDS instructions).
template <typename T, int NC, bool CAUSAL>
__device__ inline void fwd_scores(int L, const typename T::elem* Ks, const float* kmask, int qb, typename T::vec8 q0, typename T::vec8 q1, int lane,
                                  f32x4 (&S)[2 * NC], float& m, float& l) {
    using A = Attn<T>;
    const int g = lane >> 4, c = lane & 15;
    const int q = qb * 16 + c;
    const int nkc = CAUSAL ? (qb >> 1) + 1 : NC;  // 32-key chunks that hold a visible key
    m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2 * NC; ++kt) {
        S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (kt < 2 * nkc && (kt < 2 * NC - 2 || kt * 16 < L)) {  // 32 (NC - 1) < L <= 32 NC (the dispatcher's choice of NC): only the last two tiles can hold padding
            if (kt >= 2 * NC - 2 && kt * 16 + 16 > L) S[kt] = *(const f32x4*)(kmask + kt * 16 + 4 * g);  // padding keys start (and stay) at -inf: no per-element mask
            S[kt] = T::mfma16(A::rows(Ks, kt * 16, 0, lane), q0, S[kt]);
            S[kt] = T::mfma16(A::rows(Ks, kt * 16, 1, lane), q1, S[kt]);
            if constexpr (CAUSAL) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + 4 * g + r > q) S[kt][r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, S[kt][r]);
        }
    }
    m = group_max(m);
    const float nm = -m * SC;
    l = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2 * NC; ++kt)
        if (kt < 2 * nkc && (kt < 2 * NC - 2 || kt * 16 < L)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][r], SC, nm));
                S[kt][r] = e;
                l += e;
            }
        }
    l = group_sum(l);
}

template <typename T, int NC, bool CAUSAL>
__device__ inline void fwd_pv_store(const AttnArgs& p, const typename T::elem* Vs, int pair, int qb, const f32x4 (&S)[2 * NC], float m, float l, int lane) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int Lp = NC * 32;
    const int g = lane >> 4, c = lane & 15, L = p.L, HD = p.H * 64;
    const int q = qb * 16 + c;
    const int nkc = CAUSAL ? (qb >> 1) + 1 : NC;
    // (Round 4, measured and not kept: the row sums as a fifth "d tile" of ones on the matrix pipe instead of 56 VALU adds: 90 vs 77 us -- seven
    // dependent MFMAs and 8 more registers -- and a sum of ROUNDED probabilities misses the 1e-3 bound on lse in bf16.)
    f32x4 O[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) O[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < NC; ++kc)
        if (kc < nkc) {
            // keep each chunk's transposed V reads next to their MFMAs: they do not depend on the softmax, and hoisted
            // above it all 28 fragments (112 VGPRs) would be live at once
            __builtin_amdgcn_sched_barrier(0);
            const vec8 pb = A::pack2(S[2 * kc], S[2 * kc + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) O[dt] = T::mfma16(A::cols(Vs, kc * 32, dt * 16, lane), pb, O[dt]);
        }
    const int b = pair / p.H, hd = pair - b * p.H;
    const size_t ldo = p.ld_out ? (size_t)p.ld_out : (size_t)HD;
    if (q < L) {
        const size_t off = ((size_t)b * L + q) * ldo + hd * 64;
        if (p.out_lo) A::store_t_split((elem*)p.out + off, A::lo_ptr(p.out_lo, p.lo_mode, (size_t)b * L + q, ldo, hd), p.lo_mode, O, 1.f / l, lane);
        else A::store_t((elem*)p.out + off, O, 1.f / l, lane);
    }
    if (g == 0 && p.lse) p.lse[(size_t)pair * Lp + q] = q < L ? m * 0.125f + __logf(l) : 0.f;
}

template <typename T, int NC, bool CAUSAL>
__device__ inline void fwd_qblock(const AttnArgs& p, const typename T::elem* Ks, const typename T::elem* Vs, const float* kmask, int pair,
                                  int qb, typename T::vec8 q0, typename T::vec8 q1, int lane) {
    f32x4 S[2 * NC];
    float m, l;
    fwd_scores<T, NC, CAUSAL>(p.L, Ks, kmask, qb, q0, q1, lane, S, m, l);
    fwd_pv_store<T, NC, CAUSAL>(p, Vs, pair, qb, S, m, l, lane);
}

// Causal forward (text tower): one workgroup per (sequence, head) pair, several per CU.  The persistent form below pays a
// barrier per pair, and with the causal mask the query blocks cost 1, 1, 2, 2, 3 key chunks: one block per wave would idle 40 %
// of the waves (measured 10 % slower at 1000 x 8 pairs), so causal keeps two blocks per wave and no inter-pair barrier.  (A persistent variant that prefetched the next pair's
// K / V into registers was measured 35 % SLOWER: the extra 32 VGPRs push the 16-row block over 128 registers.)
template <typename T, int NC, bool CAUSAL>
__global__ __launch_bounds__(NC * 64) void attn_fwd_pair_kernel(AttnArgs p) {
    using A = Attn<T>;
    using elem = typename T::elem;
    constexpr int Lp = NC * 32, NT = NC * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem* Ks = (elem*)smem;      // [Lp][RS]
    elem* Vs = Ks + Lp * RS;     // [Lp][RS]
    float* kmask = (float*)(Vs + Lp * RS);  // [Lp] 0 for real keys, -inf for padding: the score accumulators' initial value

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;

    // This wave's (at most two: L <= 32 NC) query blocks.  Their Q fragments are requested BEFORE the K / V staging so the
    // HBM round trip overlaps it (fetched inside the block loop, every block exposed one more memory latency per wave).
    const int nqb = (L + 15) >> 4, qbA = wave, qbB = wave + NC;
    const typename T::vec8 qa0 = A::grow(base, ld, qbA * 16 + c, L, 0, lane), qa1 = A::grow(base, ld, qbA * 16 + c, L, 1, lane);
    const typename T::vec8 qb0 = A::grow(base, ld, qbB * 16 + c, L, 0, lane), qb1 = A::grow(base, ld, qbB * 16 + c, L, 1, lane);
    A::template stage2<4>(Ks, base + HD, ld, Vs, base + 2 * HD, ld, L, tid, NT);
    for (int i = tid; i < Lp; i += NT) kmask[i] = i < L ? 0.f : -INFINITY;
    __syncthreads();

    if (qbA < nqb) fwd_qblock<T, NC, CAUSAL>(p, Ks, Vs, kmask, pair, qbA, qa0, qa1, lane);
    if (qbB < nqb) fwd_qblock<T, NC, CAUSAL>(p, Ks, Vs, kmask, pair, qbB, qb0, qb1, lane);
}

// Persistent forward: one workgroup of 2 NC waves per CU walks (sequence, head) pairs; wave w owns query block w (L <= 32 NC,
// so there are at most 2 NC blocks).  K and V of the NEXT pair stream into the second pair of LDS images by LDS-DMA
// (buffer_load ... lds: no VGPR round trip, 4 instructions of 1 KiB per wave) while the current pair is computed, and the next
// pair's Q fragments are requested at the same time, so the only exposed memory latency is the first pair's.  (The previous
// form -- one pair per workgroup, two workgroups per CU, load -> barrier -> compute -- ran at 3.4 TB/s, neither HBM- nor
// MFMA-bound: the load and compute phases of a CU's two workgroups rarely overlapped.)
// Rows >= L of an image hold the next sequence's rows (finite values; zeros past the end of the tensor through the buffer
// descriptor): padded keys are masked by the -inf initial value of their score accumulators, so P = 0 multiplies them.
// Round 4, measured and not kept: "late" waves -- every other wave of a SIMD one phase group behind (P.V of the PREVIOUS pair, then scores +
// softmax of the current one, a third V image) so that one half's exponentials fall on the other half's LDS phases: 85.6 vs 75.2 us.  Nor is
// it the fourth block of the one SIMD that gets 4 of the 13 query blocks: a timing run without block 12 took 70.0 vs 76.6 us, its share of the
// work.  The time follows the instruction count (VALU ~ 1 900 cycles per block, half of them the 56 quarter-rate exponentials).
template <typename T, int NC, bool CAUSAL>
__global__ __launch_bounds__(NC * 128) void attn_fwd_kernel(AttnArgs p, int npairs) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int Lp = NC * 32, NT = NC * 128, IMG = Lp * 128;  // bytes of one image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* kmask = (float*)(smem + 4 * IMG);  // [Lp] 0 for real keys, -inf for padding: the score accumulators' initial value

    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.qkv), 0, (int)((size_t)p.B * L * ld * 2), 0x00020000);
    // lane part of a DMA source: row (lane >> 3) of an 8-row group, 16-byte chunk (lane & 7) ^ (row & 7) (the LDS slot is lane-linear)
    const int lane_src = (int)(((size_t)(lane >> 3) * ld + (size_t)(((lane & 7) ^ (lane >> 3)) << 3)) * 2);
    auto issue = [&](int pair, int buf) {
        const int b = pair / p.H, hd = pair - b * p.H;
        const unsigned base = (unsigned)(((size_t)b * L * ld + (size_t)hd * 64) * 2);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = wave * 4 + k, img = j / (Lp / 8), rg = j - img * (Lp / 8);  // 2 Lp / 8 row groups over 2 NC waves: 4 each
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (buf * 2 + img) * IMG + rg * 1024), 16,
                                                     // the whole address goes through the bounds-checked vector offset (soffset is not range-checked)
                                                     (int)(base + (unsigned)((1 + img) * HD * 2) + (unsigned)lane_src + (unsigned)(rg * 8) * (unsigned)(ld * 2)), 0, 0, 0);
        }
    };
    auto qfrag = [&](int pair, int ks) {
        const int b = pair / p.H, hd = pair - b * p.H;
        return A::grow((const elem*)p.qkv + (size_t)b * L * ld + hd * 64, ld, wave * 16 + c, L, ks, lane);
    };

    const int nqb = (L + 15) >> 4;
    int pair = blockIdx.x;
    for (int i = tid; i < Lp; i += NT) kmask[i] = i < L ? 0.f : -INFINITY;
    vec8 q0, q1;
    if (pair < npairs) {
        issue(pair, 0);
        q0 = qfrag(pair, 0);
        q1 = qfrag(pair, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (; pair < npairs; pair += gridDim.x) {
        const int nxt = pair + gridDim.x;
        vec8 n0 = q0, n1 = q1;
        if (nxt < npairs) {
            issue(nxt, cur ^ 1);
            n0 = qfrag(nxt, 0);
            n1 = qfrag(nxt, 1);
        }
        const elem* Ks = (const elem*)(smem + cur * 2 * IMG);
        if (wave < nqb) fwd_qblock<T, NC, CAUSAL>(p, Ks, Ks + Lp * RS, kmask, pair, wave, q0, q1, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next pair's images have landed (and this wave's stores are out)
        __syncthreads();                                  // ... for every wave, and everyone is done reading the current images
        q0 = n0;
        q1 = n1;
        cur ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// backward, pass 1: dQ (and delta = rowsum(dO * O)); query on the lane
// ------------------------------------------------------------------------------------------------
template <typename T, int NC, bool CAUSAL>
__global__ __launch_bounds__(NC * 64) void attn_bwd_dq_kernel(AttnArgs p, const void* fwd_out) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int Lp = NC * 32, NT = NC * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem* Ks = (elem*)smem;
    elem* Vs = Ks + Lp * RS;
    float* kmask = (float*)(Vs + Lp * RS);  // [Lp] 0 / -inf, see the forward kernel

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;
    const size_t ldof = p.ld_out ? (size_t)p.ld_out : (size_t)HD;  // (a caller may keep the forward output in rows wider than H * 64)
    const elem* Of = (const elem*)fwd_out + (size_t)b * L * ldof + hd * 64;

    struct Frags { vec8 q0, q1, g0, g1, o0, o1; };
    // window form (AttnArgs::win_n): a block without a wanted row only contributes delta -- its Q fragments are not fetched
    auto in_win = [&](int blk) { return p.win_n <= 0 || (blk * 16 < p.win_row0 + p.win_n && blk * 16 + 16 > p.win_row0); };
    auto fetch = [&](int qb) {
        const int q = qb * 16 + c, lq = in_win(qb) ? L : 0;  // lq = 0: grow() returns zeros without touching memory (wave-uniform)
        return Frags{A::grow(base, ld, q, lq, 0, lane), A::grow(base, ld, q, lq, 1, lane), A::grow(dO, HD, q, L, 0, lane),
                     A::grow(dO, HD, q, L, 1, lane), A::grow(Of, ldof, q, L, 0, lane), A::grow(Of, ldof, q, L, 1, lane)};
    };
    // the first block's Q / dO / O fragments travel during the K / V staging, the second block's during the first block's compute
    const int nqb = (L + 15) >> 4, qbA = wave, qbB = wave + NC;
    const Frags fa = fetch(qbA);
    // both blocks' log-sum-exp values travel with the first fragments (otherwise a dependent load at the head of each block)
    const size_t stat0 = ((size_t)b * p.H + hd) * Lp;
    const float lseA = p.lse[stat0 + qbA * 16 + c], lseB = qbB * 16 + c < Lp ? p.lse[stat0 + qbB * 16 + c] : 0.f;
    A::template stage2<4>(Ks, base + HD, ld, Vs, base + 2 * HD, ld, L, tid, NT);
    for (int i = tid; i < Lp; i += NT) kmask[i] = i < L ? 0.f : -INFINITY;
    __syncthreads();
    const Frags fb = fetch(qbB);

    const int qb_sel = p.sel_rows ? (p.sel_rows[b] - b * L) >> 4 : -1;
    auto block = [&](int qb, const Frags& f, float lse_q) {
        const int q = qb * 16 + c;
        if (qb_sel >= 0 && qb != qb_sel) {  // dO = 0 on every row of this block: delta = 0, dQ = 0
            if (g == 0) p.delta[((size_t)b * p.H + hd) * Lp + q] = 0.f;
            const f32x4 z[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (q < L) A::store_t((elem*)p.dqkv + ((size_t)b * L + q) * ld + hd * 64, z, 0.f, lane);
            return;
        }
        const vec8 q0 = f.q0, q1 = f.q1, g0 = f.g0, g1 = f.g1, o0 = f.o0, o1 = f.o1;
        float delta = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) delta += (float)g0[i] * (float)o0[i] + (float)g1[i] * (float)o1[i];
        delta = group_sum(delta);
        const size_t stat = ((size_t)b * p.H + hd) * Lp + q;
        if (g == 0) p.delta[stat] = delta;
        // window form: delta of EVERY query (the dK / dV pass sums over all of them), dQ only for the blocks that hold a wanted row
        if (!in_win(qb)) return;
        const float nlse = -lse_q * LOG2E;
        const int nkc = CAUSAL ? (qb >> 1) + 1 : NC;

        f32x4 dQ[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dQ[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kc = 0; kc < nkc; ++kc) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * kc + t;
                f32x4 S = *(const f32x4*)(kmask + kt * 16 + 4 * g), dP = {-delta, -delta, -delta, -delta};
                S = T::mfma16(A::rows(Ks, kt * 16, 0, lane), q0, S);
                S = T::mfma16(A::rows(Ks, kt * 16, 1, lane), q1, S);
                dP = T::mfma16(A::rows(Vs, kt * 16, 0, lane), g0, dP);
                dP = T::mfma16(A::rows(Vs, kt * 16, 1, lane), g1, dP);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padding keys: S = -inf -> p = 0; padding queries (q >= L) are never stored
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nlse));
                    if (CAUSAL && kt * 16 + 4 * g + r > q) pr = 0.f;
                    ds[t][r] = pr * dP[r];  // dS^T up to the 1 / sqrt(d) applied at the store; delta entered through dP's initial value
                }
            }
            const vec8 db = A::pack2(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dQ[dt] = T::mfma16(A::cols(Ks, kc * 32, dt * 16, lane), db, dQ[dt]);
        }
        if (q < L) A::store_t((elem*)p.dqkv + ((size_t)b * L + q) * ld + hd * 64, dQ, 0.125f, lane);
    };
    if (qbA < nqb) block(qbA, fa, lseA);
    if (qbB < nqb) block(qbB, fb, lseB);
}

// ------------------------------------------------------------------------------------------------
// backward, pass 2: dK, dV; key on the lane
// ------------------------------------------------------------------------------------------------
template <typename T, int NC, bool CAUSAL>
__global__ __launch_bounds__(NC * 64) void attn_bwd_dkv_kernel(AttnArgs p) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int Lp = NC * 32, NT = NC * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    elem* Qs = (elem*)smem;            // [Lp][RS]
    elem* Gs = Qs + Lp * RS;           // [Lp][RS]  dO rows
    float* lse_s = (float*)(Gs + Lp * RS);  // [Lp]  -lse * log2(e)
    float* del_s = lse_s + Lp;              // [Lp]

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / p.H, hd = blockIdx.x % p.H;
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;

    // this wave's (at most two) key blocks: K / V fragments requested before the Q / dO staging
    const int nkb = (L + 15) >> 4, kbA = wave, kbB = wave + NC;
    struct Frags { vec8 k0, k1, v0, v1; };
    auto in_win = [&](int blk) { return p.win_n <= 0 || (blk * 16 < p.win_row0 + p.win_n && blk * 16 + 16 > p.win_row0); };
    auto fetch = [&](int kb) {
        const int key = kb * 16 + c, lk = in_win(kb) ? L : 0;  // window form: blocks without a wanted key row fetch nothing
        return Frags{A::grow(base + HD, ld, key, lk, 0, lane), A::grow(base + HD, ld, key, lk, 1, lane),
                     A::grow(base + 2 * HD, ld, key, lk, 0, lane), A::grow(base + 2 * HD, ld, key, lk, 1, lane)};
    };
    const Frags fa = fetch(kbA), fb = fetch(kbB);
    // the row statistics (Lp = NT / 2 values: one per thread) are requested with the fragments, BEFORE the staging waits for
    // its loads: one HBM round trip per workgroup instead of two (a workgroup is a latency chain: 2 per CU)
    float lse_v = -INFINITY, del_v = 0.f;  // padding queries: p = exp2(. - inf) = 0, no per-element mask
    if (tid < L) {
        const size_t stat = ((size_t)b * p.H + hd) * Lp + tid;
        lse_v = -p.lse[stat] * LOG2E;
        del_v = p.delta[stat];
    }
    A::template stage2<4>(Qs, base, ld, Gs, dO, (size_t)HD, L, tid, NT);
    if (tid < Lp) { lse_s[tid] = lse_v; del_s[tid] = del_v; }
    __syncthreads();

    const int qc_sel = p.sel_rows ? (p.sel_rows[b] - b * L) >> 5 : -1;  // dO is zero outside this 32-query chunk
    auto block = [&](int kb, const Frags& f) {
        const int key = kb * 16 + c;
        if (!in_win(kb)) return;  // window form: no wanted key row in this block
        const vec8 k0 = f.k0, k1 = f.k1, v0 = f.v0, v1 = f.v1;
        f32x4 dK[4], dV[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int qc_lo = CAUSAL ? (kb >> 1) : 0;  // query chunks that hold a query >= this block's first key
        const int qc0 = qc_sel >= 0 ? (qc_sel > qc_lo ? qc_sel : qc_lo) : qc_lo, qc1 = qc_sel >= 0 ? qc_sel + 1 : NC;
#pragma unroll 1
        for (int qc = qc0; qc < qc1; ++qc) {
            f32x4 P[2], dS[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = 2 * qc + t;
                const f32x4 nl = *(const f32x4*)(lse_s + qt * 16 + 4 * g);
                const f32x4 d4 = *(const f32x4*)(del_s + qt * 16 + 4 * g);
                f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = -d4;
                S = T::mfma16(A::rows(Qs, qt * 16, 0, lane), k0, S);
                S = T::mfma16(A::rows(Qs, qt * 16, 1, lane), k1, S);
                dP = T::mfma16(A::rows(Gs, qt * 16, 0, lane), v0, dP);
                dP = T::mfma16(A::rows(Gs, qt * 16, 1, lane), v1, dP);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padding queries have nl = -inf -> p = 0; padding keys (key >= L) are whole lanes that are never stored
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nl[r]));
                    if (CAUSAL && key > qt * 16 + 4 * g + r) pr = 0.f;
                    P[t][r] = pr;
                    dS[t][r] = pr * dP[r];  // 1 / sqrt(d) applied at the store
                }
            }
            const vec8 pb = A::pack2(P[0], P[1]), db = A::pack2(dS[0], dS[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dV[dt] = T::mfma16(A::cols(Gs, qc * 32, dt * 16, lane), pb, dV[dt]);
                dK[dt] = T::mfma16(A::cols(Qs, qc * 32, dt * 16, lane), db, dK[dt]);
            }
        }
        if (key < L) {
            elem* ok = (elem*)p.dqkv + ((size_t)b * L + key) * ld + HD + hd * 64;
            A::store_t(ok, dK, 0.125f, lane);
            A::store_t(ok + HD, dV, 1.f, lane);
        }
    };
    if (kbA < nkb) block(kbA, fa);
    if (kbB < nkb) block(kbB, fb);
}

// ------------------------------------------------------------------------------------------------
// backward, fused: dQ, dK, dV of a (sequence, head) pair from ONE resident pass over Q, K, V, dO
// ------------------------------------------------------------------------------------------------
// The two-kernel form above reads Q, K, V and dO from HBM twice (948 MB per layer at B 256 / L 201 / H 12 where one pass needs
// 632 MB) and round-trips delta through HBM.  Here one persistent workgroup per CU keeps all four [Lp][64] images of a pair in LDS
// (4 x 28 KB at Lp = 224) and runs both sweeps on them -- still two independent sweeps (no dS image, no atomics, bitwise
// reproducible), recomputing S and dP once as before:
//   phase A (query on the lane, = attn_bwd_dq_kernel's block): dQ from the K / V images and this wave's Q / dO / O fragments; delta and
//     -lse log2(e) go to LDS for phase B.  Meanwhile the pair's Q / dO images arrive by LDS-DMA.
//   phase B (key on the lane, = attn_bwd_dkv_kernel's block): dK, dV from the Q / dO images; the wave's K / V fragments were read from
//     the K / V images at the end of phase A, so those images are free: the NEXT pair's K / V stream into them, and the next pair's
//     Q / dO / O fragments travel to registers.
// One barrier between the phases and one per pair; the only exposed memory latency is the first pair's.  The fragments of the
// "own" operand (Q / dO in phase A) are fetched straight from global memory half a pair before the same lines are DMA-ed as images:
// that second touch is served by the L2 / Infinity Cache, not by HBM.
// Image rows >= L hold the next sequence's rows (finite values; zeros past the end of the tensor through the buffer descriptor):
// padded keys are masked by the -inf initial value of their score accumulators, padded queries by lse = -inf, so P = 0 meets them.
// W2 = waves per 32 rows: 1 -> NC waves, two 16-row blocks per wave and phase; 2 -> 2 NC waves, one block each.
// Measured (round 2, tools/attn_bench.py, bf16): text tower, 1000 x 8 pairs of L = 77: 212 us vs 260 us for the two kernels; 11 x 8 pairs
// 7.5 vs 12.9 us -- several workgroups fit a CU (NC <= 3: <= 51 KB of LDS).  Vision tower (B 256, L 201, H 12: NC = 7, 117 KB, ONE
// workgroup per CU): 297 us (W2 = 2, 24 spilled VGPRs at the 128-register cap of 14 waves) / 275 us (W2 = 1) vs 261 us for the two
// kernels at two workgroups per CU: the kernels are bound by instruction issue (VALU ~ MFMA ~ LDS issue, ~20 us per pair per CU against
// 4.4 us of matrix-pipe time), not by the 948 -> 632 MB of HBM traffic the single pass saves, and one workgroup per CU hides less
// latency.  So the dispatcher uses this kernel for NC <= 3 and the two-kernel form above it.
template <typename T, int NC, bool CAUSAL, int W2>
__global__ __launch_bounds__(NC * 64 * W2) void attn_bwd_fused_kernel(AttnArgs p, const void* fwd_out, int npairs) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int Lp = NC * 32, NW = NC * W2, NT = NW * 64, IMG = Lp * 128, NBLK = 2 / W2, RGW = 8 / W2;  // RGW: 8-row groups per wave per image pair
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const elem* Ks = (const elem*)smem;
    const elem* Vs = (const elem*)(smem + IMG);
    const elem* Qs = (const elem*)(smem + 2 * IMG);
    const elem* Gs = (const elem*)(smem + 3 * IMG);
    float* kmask = (float*)(smem + 4 * IMG);  // [Lp] 0 for real keys, -inf for padding
    float* lse_s = kmask + Lp;                // [Lp] -lse log2(e) of the current pair's queries (-inf for padding)
    float* del_s = lse_s + Lp;                // [Lp] delta = rowsum(dO * O)

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const size_t ldof = p.ld_out ? (size_t)p.ld_out : (size_t)HD;  // (a caller may keep the forward output in rows wider than H * 64)
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.qkv), 0, (int)((size_t)p.B * L * ld * 2), 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dout), 0, (int)((size_t)p.B * L * HD * 2), 0x00020000);
    // lane part of a DMA source: row (lane >> 3) of an 8-row group, 16-byte chunk (lane & 7) ^ (row & 7) (the LDS slot is lane-linear)
    const unsigned lane_q = (unsigned)(((size_t)(lane >> 3) * ld + (size_t)(((lane & 7) ^ (lane >> 3)) << 3)) * 2);
    const unsigned lane_g = (unsigned)(((size_t)(lane >> 3) * HD + (size_t)(((lane & 7) ^ (lane >> 3)) << 3)) * 2);
    using lds_ptr = __attribute__((address_space(3))) void*;
    // K and V images of a pair -> smem[0, 2 IMG)
    auto issue_kv = [&](int pair) {
        const int b = pair / p.H, hd = pair - b * p.H;
        const unsigned base = (unsigned)(((size_t)b * L * ld + (size_t)hd * 64) * 2);
#pragma unroll
        for (int k = 0; k < RGW; ++k) {
            const int j = wave * RGW + k, img = j / (Lp / 8), rg = j - img * (Lp / 8);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr)(smem + img * IMG + rg * 1024), 16,
                                                     (int)(base + (unsigned)((1 + img) * HD * 2) + lane_q + (unsigned)(rg * 8) * (unsigned)(ld * 2)), 0, 0, 0);
        }
    };
    // Q and dO images of a pair -> smem[2 IMG, 4 IMG)
    auto issue_qg = [&](int pair) {
        const int b = pair / p.H, hd = pair - b * p.H;
        const unsigned baseq = (unsigned)(((size_t)b * L * ld + (size_t)hd * 64) * 2), baseg = (unsigned)(((size_t)b * L * HD + (size_t)hd * 64) * 2);
#pragma unroll
        for (int k = 0; k < RGW; ++k) {
            const int j = wave * RGW + k, img = j / (Lp / 8), rg = j - img * (Lp / 8);
            if (img == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr)(smem + 2 * IMG + rg * 1024), 16, (int)(baseq + lane_q + (unsigned)(rg * 8) * (unsigned)(ld * 2)), 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr)(smem + 3 * IMG + rg * 1024), 16, (int)(baseg + lane_g + (unsigned)(rg * 8) * (unsigned)(HD * 2)), 0, 0, 0);
        }
    };
    struct QFrags { vec8 q0, q1, g0, g1, o0, o1; float lse; };
    auto fetch_q = [&](int pair, int qb) {
        const int b = pair / p.H, hd = pair - b * p.H, q = qb * 16 + c;
        const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
        const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;
        const elem* Of = (const elem*)fwd_out + (size_t)b * L * ldof + hd * 64;
        QFrags f{A::grow(base, ld, q, L, 0, lane), A::grow(base, ld, q, L, 1, lane), A::grow(dO, HD, q, L, 0, lane), A::grow(dO, HD, q, L, 1, lane),
                 A::grow(Of, ldof, q, L, 0, lane), A::grow(Of, ldof, q, L, 1, lane), 0.f};
        f.lse = q < Lp ? p.lse[(size_t)pair * Lp + q] : 0.f;
        return f;
    };
    struct KFrags { vec8 k0, k1, v0, v1; };

    const int nb16 = (L + 15) >> 4;  // 16-row blocks that hold a real row (queries and keys alike)
    // ---- phase A block: dQ of query block qb; leaves delta / lse of its queries in LDS ------------------------------------
    auto phase_a = [&](int pair, int qb, const QFrags& f) {
        const int b = pair / p.H, hd = pair - b * p.H, q = qb * 16 + c;
        float delta = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) delta += (float)f.g0[i] * (float)f.o0[i] + (float)f.g1[i] * (float)f.o1[i];
        delta = group_sum(delta);
        const float nlse = -f.lse * LOG2E;
        if (g == 0) {
            del_s[q] = delta;
            lse_s[q] = q < L ? nlse : -INFINITY;
        }
        const int nkc = CAUSAL ? (qb >> 1) + 1 : NC;
        f32x4 dQ[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dQ[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int kc = 0; kc < nkc; ++kc) {
            f32x4 ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kt = 2 * kc + t;
                f32x4 S = *(const f32x4*)(kmask + kt * 16 + 4 * g), dP = {-delta, -delta, -delta, -delta};
                S = T::mfma16(A::rows(Ks, kt * 16, 0, lane), f.q0, S);
                S = T::mfma16(A::rows(Ks, kt * 16, 1, lane), f.q1, S);
                dP = T::mfma16(A::rows(Vs, kt * 16, 0, lane), f.g0, dP);
                dP = T::mfma16(A::rows(Vs, kt * 16, 1, lane), f.g1, dP);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nlse));
                    if (CAUSAL && kt * 16 + 4 * g + r > q) pr = 0.f;
                    ds[t][r] = pr * dP[r];
                }
            }
            const vec8 db = A::pack2(ds[0], ds[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dQ[dt] = T::mfma16(A::cols(Ks, kc * 32, dt * 16, lane), db, dQ[dt]);
        }
        if (q < L) A::store_t((elem*)p.dqkv + ((size_t)b * L + q) * ld + hd * 64, dQ, 0.125f, lane);
    };
    // ---- phase B block: dK, dV of key block kb --------------------------------------------------------------------------------
    auto phase_b = [&](int pair, int kb, const KFrags& f) {
        const int b = pair / p.H, hd = pair - b * p.H, key = kb * 16 + c;
        f32x4 dK[4], dV[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const int qc0 = CAUSAL ? (kb >> 1) : 0;
#pragma unroll 1
        for (int qc = qc0; qc < NC; ++qc) {
            f32x4 P[2], dS[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = 2 * qc + t;
                const f32x4 nl = *(const f32x4*)(lse_s + qt * 16 + 4 * g);
                const f32x4 d4 = *(const f32x4*)(del_s + qt * 16 + 4 * g);
                f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = -d4;
                S = T::mfma16(A::rows(Qs, qt * 16, 0, lane), f.k0, S);
                S = T::mfma16(A::rows(Qs, qt * 16, 1, lane), f.k1, S);
                dP = T::mfma16(A::rows(Gs, qt * 16, 0, lane), f.v0, dP);
                dP = T::mfma16(A::rows(Gs, qt * 16, 1, lane), f.v1, dP);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nl[r]));
                    if (CAUSAL && key > qt * 16 + 4 * g + r) pr = 0.f;
                    P[t][r] = pr;
                    dS[t][r] = pr * dP[r];
                }
            }
            const vec8 pb = A::pack2(P[0], P[1]), db = A::pack2(dS[0], dS[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dV[dt] = T::mfma16(A::cols(Gs, qc * 32, dt * 16, lane), pb, dV[dt]);
                dK[dt] = T::mfma16(A::cols(Qs, qc * 32, dt * 16, lane), db, dK[dt]);
            }
        }
        if (key < L) {
            elem* ok = (elem*)p.dqkv + ((size_t)b * L + key) * ld + HD + hd * 64;
            A::store_t(ok, dK, 0.125f, lane);
            A::store_t(ok + HD, dV, 1.f, lane);
        }
    };

    for (int i = tid; i < Lp; i += NT) { kmask[i] = i < L ? 0.f : -INFINITY; lse_s[i] = -INFINITY; del_s[i] = 0.f; }
    int pair = blockIdx.x;
    QFrags qf[NBLK];
    if (pair < npairs) {
        issue_kv(pair);
#pragma unroll
        for (int j = 0; j < NBLK; ++j) qf[j] = fetch_q(pair, wave + j * NW);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (; pair < npairs; pair += gridDim.x) {
        // ---- phase A ----
        issue_qg(pair);
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
            if (wave + j * NW < nb16) phase_a(pair, wave + j * NW, qf[j]);
        KFrags kf[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
            const int r0 = (wave + j * NW) * 16;  // < Lp: 2 NC blocks in all
            kf[j] = KFrags{A::rows(Ks, r0, 0, lane), A::rows(Ks, r0, 1, lane), A::rows(Vs, r0, 0, lane), A::rows(Vs, r0, 1, lane)};
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // Q / dO images landed, this wave's K / V fragments and dQ stores are done
        __syncthreads();                                              // ... for every wave: the K / V images are free, lse_s / del_s are complete
        // ---- phase B ----
        const int nxt = pair + gridDim.x;
        if (nxt < npairs) {
            issue_kv(nxt);
#pragma unroll
            for (int j = 0; j < NBLK; ++j) qf[j] = fetch_q(nxt, wave + j * NW);
        }
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
            if (wave + j * NW < nb16) phase_b(pair, wave + j * NW, kf[j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next pair's K / V images and fragments have landed
        __syncthreads();                                  // ... for every wave, and everyone is done with the Q / dO images and lse_s / del_s
    }
}

// ------------------------------------------------------------------------------------------------
// backward, single sweep: S, P, dP, dS computed ONCE per (query, key); dS crosses LDS for dQ
// ------------------------------------------------------------------------------------------------
// The kernels above are bound by instruction issue (VALU ~ MFMA ~ LDS issue), not by memory, and both the dQ pass and the dK/dV pass
// recompute S = Q K^T, dP = dO V^T and the exponentials: 8 of their 20 MFMAs per 16 x 32 tile and half of their VALU work are done twice.
// This kernel runs the dK/dV sweep ("key on the lane": wave w owns key block w, K / V fragments in registers, Q / dO images in LDS)
// and stores every dS tile, already rounded to T for the dK product, into an LDS image DS[key][query]; afterwards wave w computes
// dQ of query block w = sum over ALL keys of dS[q, key] K[key, :] from that image (hardware-transposed reads give the operand with the
// query on the lane) and a K image that is DMA-ed into the Q image's place once the sweep is over.  No atomics, no cross-wave
// reduction: every sum has one owner and a fixed order (bitwise reproducible).  delta = rowsum(dO * O) is computed by the owner of
// the query block before the sweep and stays in LDS.
// LDS at Lp = 224: Q and dO images 2 x 28 KB + DS 224 x 224 x 2 B = 98 KB + statistics = 155.8 KB: one workgroup of 2 NC waves per CU.
// Measured (round 2, B 256 / L 201 / H 12, bf16): 216 us against 261 us for the two kernels.  Timing ablations of this kernel (phases
// skipped by a run-time flag): loads alone 81 us (474 MB at 5.8 TB/s: that phase is HBM-bound), + sweep 186 us, all 235 us -- a CU
// runs its phases one after the other.  Two attempts to overlap them were SLOWER and are not kept: a persistent form with the K
// image resident, the query range swept in two halves (half the dS image) and the next pair's images / fragments prefetched: 255 us
// (252 us with the workgroups' starts staggered; its dQ phases keep only 8 and 5 of the 14 waves busy and the persistent
// workgroups load and compute in lockstep), and fully unrolled loops with immediate LDS offsets (214 us: not the instruction count).
// DS image: panels of 32 queries, [panel][key][32 queries] with 64-byte rows; the two 32-byte halves of a row are swapped for keys
// with bit 2 set, which makes the transposed 4-key x 16-query block reads conflict-free; inside a half the four 8-byte units (4 queries
// each) are rotated by two for keys with bit 3 set (round 4): the sweep's 8-byte writes of keys c and c + 8 of a wave otherwise fall on
// the same banks (2-way conflicts on every dS store: SQ_LDS_BANK_CONFLICT was 18 % of the LDS pipe's active cycles); the block reads
// stay conflict-free (a 16-lane group reads four keys that share bits 2 and 3).
template <int LP>
__device__ inline int ds_off(int key, int q) {  // byte offset of element (key, q); q & 3 == 0 for the 8-byte accesses
    return (q >> 5) * (LP * 64) + key * 64 + ((((q >> 4) & 1) ^ ((key >> 2) & 1)) << 5) + ((((q & 15) >> 2) ^ (((key >> 3) & 1) << 1)) << 3) + (q & 3) * 2;
}

template <typename T, int NC>
__global__ __launch_bounds__(NC * 128) void attn_bwd_sweep_kernel(AttnArgs p, const void* fwd_out) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    constexpr int Lp = NC * 32, NW = NC * 2, NT = NW * 64, IMG = Lp * 128, DSB = Lp * Lp * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const elem* Qs = (const elem*)smem;           // Q image during the sweep, K image afterwards
    const elem* Gs = (const elem*)(smem + IMG);   // dO image
    char* DS = smem + 2 * IMG;
    float* lse_s = (float*)(smem + 2 * IMG + DSB);  // [Lp] -lse log2(e) (-inf for padded queries)
    float* del_s = lse_s + Lp;                      // [Lp] delta

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L;
    const size_t ld = (size_t)3 * HD;
    const size_t ldof = p.ld_out ? (size_t)p.ld_out : (size_t)HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;
    const elem* Of = (const elem*)fwd_out + (size_t)b * L * ldof + hd * 64;
    const auto rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.qkv), 0, (int)((size_t)p.B * L * ld * 2), 0x00020000);
    const auto rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dout), 0, (int)((size_t)p.B * L * HD * 2), 0x00020000);
    const unsigned lane_q = (unsigned)(((size_t)(lane >> 3) * ld + (size_t)(((lane & 7) ^ (lane >> 3)) << 3)) * 2);
    const unsigned lane_g = (unsigned)(((size_t)(lane >> 3) * HD + (size_t)(((lane & 7) ^ (lane >> 3)) << 3)) * 2);
    const unsigned baseq = (unsigned)(((size_t)b * L * ld + (size_t)hd * 64) * 2), baseg = (unsigned)(((size_t)b * L * HD + (size_t)hd * 64) * 2);
    using lds_ptr = __attribute__((address_space(3))) void*;

    // ---- loads: Q and dO images by LDS-DMA (4 row groups of 8 rows per wave), this wave's K / V fragments, and the O / dO fragments
    // of its query block for delta
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = wave * 4 + k, img = j / (Lp / 8), rg = j - img * (Lp / 8);
        if (img == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr)(smem + rg * 1024), 16, (int)(baseq + lane_q + (unsigned)(rg * 8) * (unsigned)(ld * 2)), 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsG, (lds_ptr)(smem + IMG + rg * 1024), 16, (int)(baseg + lane_g + (unsigned)(rg * 8) * (unsigned)(HD * 2)), 0, 0, 0);
    }
    const int nb16 = (L + 15) >> 4;
    const int row = wave * 16 + c;  // this lane's key (sweep) and query (delta, final phase)
    const vec8 k0 = A::grow(base + HD, ld, row, L, 0, lane), k1 = A::grow(base + HD, ld, row, L, 1, lane);
    const vec8 v0 = A::grow(base + 2 * HD, ld, row, L, 0, lane), v1 = A::grow(base + 2 * HD, ld, row, L, 1, lane);
    {
        const vec8 g0 = A::grow(dO, HD, row, L, 0, lane), g1 = A::grow(dO, HD, row, L, 1, lane);
        const vec8 o0 = A::grow(Of, ldof, row, L, 0, lane), o1 = A::grow(Of, ldof, row, L, 1, lane);
        const float lse_q = row < L ? p.lse[(size_t)pair * Lp + row] : 0.f;
        // DS rows of the key blocks that hold no real key are never written by a sweep: they must read as zero
        for (int i = tid; i < (Lp - nb16 * 16) * NC * 4; i += NT) {  // (Lp - 16 nb16) keys x NC panels x 64 bytes, 16 bytes per thread
            const int chunk = i & 3, rp = i >> 2, key = nb16 * 16 + rp % (Lp - nb16 * 16), panel = rp / (Lp - nb16 * 16);
            *(f32x4*)(DS + panel * (Lp * 64) + key * 64 + chunk * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float delta = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) delta += (float)g0[i] * (float)o0[i] + (float)g1[i] * (float)o1[i];
        delta = group_sum(delta);
        if (g == 0) {
            del_s[row] = delta;
            lse_s[row] = row < L ? -lse_q * LOG2E : -INFINITY;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- sweep: dK, dV of key block `wave`; dS -> DS -------------------------------------------------------------------------
    if (wave < nb16) {
        const bool kvalid = row < L;
        f32x4 dK[4], dV[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { dK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        // (round 4: skipping the last query tile when it lies wholly beyond L -- tile 13 of 14 at L = 201 -- behind a uniform branch: 215.5 vs 212.0 us)
        for (int qc = 0; qc < NC; ++qc) {  // fully unrolled: every LDS address becomes a loop-invariant lane base + an immediate offset
            f32x4 P[2], dS[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int qt = 2 * qc + t;
                const f32x4 nl = *(const f32x4*)(lse_s + qt * 16 + 4 * g);
                const f32x4 d4 = *(const f32x4*)(del_s + qt * 16 + 4 * g);
                f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = -d4;
                S = T::mfma16(A::rows(Qs, qt * 16, 0, lane), k0, S);
                S = T::mfma16(A::rows(Qs, qt * 16, 1, lane), k1, S);
                dP = T::mfma16(A::rows(Gs, qt * 16, 0, lane), v0, dP);
                dP = T::mfma16(A::rows(Gs, qt * 16, 1, lane), v1, dP);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded queries have nl = -inf -> p = 0; padded keys (whole lanes) are zeroed below
                    const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nl[r]));
                    P[t][r] = pr;
                    dS[t][r] = pr * dP[r];  // 1 / sqrt(d) applied at the stores
                }
            }
            const vec8 pb = A::pack2(P[0], P[1]), db = A::pack2(dS[0], dS[1]);
            // dS of this lane's key for queries 32 qc + 4 g .. + 3 (tile 2 qc) and 32 qc + 16 + 4 g .. (tile 2 qc + 1): 8 bytes each
            {
                vec4 lo = {db[0], db[1], db[2], db[3]}, hi = {db[4], db[5], db[6], db[7]};
                if (!kvalid) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { lo[i] = (elem)0.f; hi[i] = (elem)0.f; }
                }
                *(vec4*)(DS + ds_off<Lp>(row, 32 * qc + 4 * g)) = lo;
                *(vec4*)(DS + ds_off<Lp>(row, 32 * qc + 16 + 4 * g)) = hi;
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dV[dt] = T::mfma16(A::cols(Gs, qc * 32, dt * 16, lane), pb, dV[dt]);
                dK[dt] = T::mfma16(A::cols(Qs, qc * 32, dt * 16, lane), db, dK[dt]);
            }
        }
        if (kvalid) {
            elem* ok = (elem*)p.dqkv + ((size_t)b * L + row) * ld + HD + hd * 64;
            A::store_t(ok, dK, 0.125f, lane);
            A::store_t(ok + HD, dV, 1.f, lane);
        }
    }
    __syncthreads();  // DS is complete and nobody reads the Q image any more

    // ---- the K image takes the Q image's place (2 row groups per wave) -------------------------------------------------------
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int rg = wave * 2 + k;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr)(smem + rg * 1024), 16, (int)(baseq + (unsigned)(HD * 2) + lane_q + (unsigned)(rg * 8) * (unsigned)(ld * 2)), 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- dQ of query block `wave`: dQ^T[d, q] = sum over keys K[key, d] dS[q, key] --------------------------------------------
    if (wave < nb16) {
        f32x4 dQ[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dQ[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // transposed block read of DS: lane i of 16-lane group g addresses key row 32 kc + 4 g + (i >> 2), queries 16 wave + 4 (i & 3) .. + 3
        // and receives query 16 wave + i of the block's four keys; the second read takes the keys 16 further (same swizzle)
        const char* dsp = DS + ds_off<Lp>(4 * g + (c >> 2), 16 * wave + 4 * (c & 3));
#pragma unroll
        for (int kc = 0; kc < NC; ++kc) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dsp + kc * 32 * 64));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(dsp + kc * 32 * 64 + 16 * 64));
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            const vec8 db = __builtin_bit_cast(vec8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dQ[dt] = T::mfma16(A::cols(Qs, kc * 32, dt * 16, lane), db, dQ[dt]);
        }
        if (row < L) A::store_t((elem*)p.dqkv + ((size_t)b * L + row) * ld + hd * 64, dQ, 0.125f, lane);
    }
}

// ------------------------------------------------------------------------------------------------
// Long sequences (L > 224: ViT-L/14@336 has 577 + n_ctx vision tokens, BASELINE configs[4]), STAGED form: the other operand
// streams through 64-row stages and the forward keeps a running (online) softmax.  Since the second half of round 3 the default up to
// L = 608 / 640 (backward / forward) is the RESIDENT form of attention_resident.hip, where a pair's streamed operands do fit LDS whole;
// these kernels serve longer sequences, the window and single-row forms and A/B runs (flag bit 1).  Same
// fragment conventions and inner products as the whole-sequence kernels above.  One workgroup = 8 waves = 128 rows of the "lane"
// operand (round 3; 4 waves / 64 rows before: every stage of the streamed operand now feeds twice the MFMAs).  The stages are double
// buffered: the global loads of stage st + 1 are issued before stage st is computed and written to the other LDS buffer after it
// (registers carry them across the compute), so a stage costs ONE barrier and no exposed memory round trip (two barriers and a
// synchronous global -> LDS copy per stage before).
// ------------------------------------------------------------------------------------------------
constexpr int TW = 8, TROWS = TW * 16;  // waves per workgroup, rows of the lane operand per workgroup

template <typename T>
struct Stage64 {  // one 64-row stage of two [.., 64]-wide operands: 64 rows x 8 chunks x 2 images over 512 threads = one chunk of each per thread
    using vec8 = typename T::vec8;
    using elem = typename T::elem;
    vec8 a, b;
    __device__ inline void load(const elem* src0, size_t ld0, const elem* src1, size_t ld1, int row0, int L, int tid) {
        const int row = tid >> 3, ch = tid & 7;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = (elem)0.f; b[i] = (elem)0.f; }
        if (row0 + row < L) {
            a = *(const vec8*)(src0 + (size_t)(row0 + row) * ld0 + ch * 8);
            b = *(const vec8*)(src1 + (size_t)(row0 + row) * ld1 + ch * 8);
        }
    }
    __device__ inline void write(elem* img0, elem* img1, int tid) const {
        const int row = tid >> 3, ch = tid & 7;
        *(vec8*)(img0 + row * RS + ((ch ^ (row & 7)) << 3)) = a;
        *(vec8*)(img1 + row * RS + ((ch ^ (row & 7)) << 3)) = b;
    }
};

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(TW * 64) void attn_fwd_tiled_kernel(AttnArgs p, int nsb) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    __shared__ __attribute__((aligned(16))) elem Ks[2][64 * RS];
    __shared__ __attribute__((aligned(16))) elem Vs[2][64 * RS];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x / nsb, sb = blockIdx.x - pair * nsb, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L, Lp = attn_padded_len_dev(L);
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const int q = sb * TROWS + wave * 16 + c;
    const bool live = sb * TROWS + wave * 16 < L;  // this wave holds a real query (uniform per wave)
    const vec8 q0 = A::grow(base, ld, q, L, 0, lane), q1 = A::grow(base, ld, q, L, 1, lane);
    float m = -INFINITY, l = 0.f;
    f32x4 O[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) O[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nall = (L + 63) / 64, last_q = sb * TROWS + TROWS - 1;
    const int nst = CAUSAL ? (last_q / 64 + 1 < nall ? last_q / 64 + 1 : nall) : nall;
    Stage64<T> sg;
    sg.load(base + HD, ld, base + 2 * HD, ld, 0, L, tid);
    sg.write(Ks[0], Vs[0], tid);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) sg.load(base + HD, ld, base + 2 * HD, ld, (st + 1) * 64, L, tid);
        if (live && (!CAUSAL || st * 64 <= sb * TROWS + wave * 16 + 15)) {
            f32x4 S[4];
            float mloc = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                S[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                S[kt] = T::mfma16(A::rows(Ks[cur], kt * 16, 0, lane), q0, S[kt]);
                S[kt] = T::mfma16(A::rows(Ks[cur], kt * 16, 1, lane), q1, S[kt]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = st * 64 + kt * 16 + 4 * g + r;
                    if (key >= L || (CAUSAL && key > q)) S[kt][r] = -INFINITY;
                    mloc = fmaxf(mloc, S[kt][r]);
                }
            }
            const float mnew = fmaxf(m, group_max(mloc));  // finite from the first stage on: key 0 is visible to every query
            const float alpha = mnew == -INFINITY ? 1.f : __builtin_amdgcn_exp2f((m - mnew) * SC), nm = mnew == -INFINITY ? 0.f : -mnew * SC;
            m = mnew;
            l *= alpha;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(S[kt][r], SC, nm));
                    S[kt][r] = e;
                    l += e;
                }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) O[dt] *= alpha;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const vec8 pb = A::pack2(S[2 * kc], S[2 * kc + 1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) O[dt] = T::mfma16(A::cols(Vs[cur], kc * 32, dt * 16, lane), pb, O[dt]);
            }
        }
        if (st + 1 < nst) sg.write(Ks[cur ^ 1], Vs[cur ^ 1], tid);
        __syncthreads();  // stage st + 1 is in place; everyone is done reading stage st
    }
    l = group_sum(l);
    const size_t ldo = p.ld_out ? (size_t)p.ld_out : (size_t)HD;
    if (q < L) {
        const size_t off = ((size_t)b * L + q) * ldo + hd * 64;
        if (p.out_lo) A::store_t_split((elem*)p.out + off, A::lo_ptr(p.out_lo, p.lo_mode, (size_t)b * L + q, ldo, hd), p.lo_mode, O, 1.f / l, lane);
        else A::store_t((elem*)p.out + off, O, 1.f / l, lane);
    }
    if (g == 0 && p.lse && q < Lp) p.lse[(size_t)pair * Lp + q] = q < L ? m * 0.125f + __logf(l) : 0.f;
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(TW * 64) void attn_bwd_dq_tiled_kernel(AttnArgs p, const void* fwd_out, int nsb) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    __shared__ __attribute__((aligned(16))) elem Ks[2][64 * RS];
    __shared__ __attribute__((aligned(16))) elem Vs[2][64 * RS];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x / nsb, sb = blockIdx.x - pair * nsb, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L, Lp = attn_padded_len_dev(L);
    const size_t ld = (size_t)3 * HD, ldof = p.ld_out ? (size_t)p.ld_out : (size_t)HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;
    const elem* Of = (const elem*)fwd_out + (size_t)b * L * ldof + hd * 64;
    const int q = sb * TROWS + wave * 16 + c;
    if (p.sel_rows && (p.sel_rows[b] - b * L) / TROWS != sb) {  // dO = 0 on this workgroup's queries (uniform branch)
        if (g == 0 && q < Lp) p.delta[(size_t)pair * Lp + q] = 0.f;
        const f32x4 z[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (q < L) A::store_t((elem*)p.dqkv + ((size_t)b * L + q) * ld + hd * 64, z, 0.f, lane);
        return;
    }
    const vec8 q0 = A::grow(base, ld, q, L, 0, lane), q1 = A::grow(base, ld, q, L, 1, lane);
    const vec8 g0 = A::grow(dO, HD, q, L, 0, lane), g1 = A::grow(dO, HD, q, L, 1, lane);
    const vec8 o0 = A::grow(Of, ldof, q, L, 0, lane), o1 = A::grow(Of, ldof, q, L, 1, lane);
    float delta = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) delta += (float)g0[i] * (float)o0[i] + (float)g1[i] * (float)o1[i];
    delta = group_sum(delta);
    const size_t stat = (size_t)pair * Lp + (q < Lp ? q : 0);
    if (g == 0 && q < Lp) p.delta[stat] = delta;
    // window form: delta of every query, dQ only for the workgroups that hold a wanted row; a wave that leaves skips the barriers
    // below, so the decision is made per WORKGROUP (its TROWS queries)
    if (p.win_n > 0 && (sb * TROWS >= p.win_row0 + p.win_n || sb * TROWS + TROWS <= p.win_row0)) return;
    const bool live = sb * TROWS + wave * 16 < L;
    const float nlse = q < L ? -p.lse[stat] * LOG2E : 0.f;
    f32x4 dQ[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dQ[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nall = (L + 63) / 64, last_q = sb * TROWS + TROWS - 1;
    const int nst = CAUSAL ? (last_q / 64 + 1 < nall ? last_q / 64 + 1 : nall) : nall;
    Stage64<T> sg;
    sg.load(base + HD, ld, base + 2 * HD, ld, 0, L, tid);
    sg.write(Ks[0], Vs[0], tid);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int cur = st & 1;
        if (st + 1 < nst) sg.load(base + HD, ld, base + 2 * HD, ld, (st + 1) * 64, L, tid);
        if (live && (!CAUSAL || st * 64 <= sb * TROWS + wave * 16 + 15)) {
            // only the stage that holds the end of the sequence (and causal stages) pays for the per-key mask (8 compare / select pairs per
            // 16 x 64 block: a sixth of the stage's vector issue)
            auto block = [&](auto masked_c) {
                constexpr bool MASKED = decltype(masked_c)::value;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    f32x4 ds[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int kt = 2 * kc + t;
                        f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = {-delta, -delta, -delta, -delta};
                        S = T::mfma16(A::rows(Ks[cur], kt * 16, 0, lane), q0, S);
                        S = T::mfma16(A::rows(Ks[cur], kt * 16, 1, lane), q1, S);
                        dP = T::mfma16(A::rows(Vs[cur], kt * 16, 0, lane), g0, dP);
                        dP = T::mfma16(A::rows(Vs[cur], kt * 16, 1, lane), g1, dP);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = st * 64 + kt * 16 + 4 * g + r;
                            float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nlse));
                            if (MASKED && (key >= L || (CAUSAL && key > q))) pr = 0.f;
                            ds[t][r] = pr * dP[r];
                        }
                    }
                    const vec8 db = A::pack2(ds[0], ds[1]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) dQ[dt] = T::mfma16(A::cols(Ks[cur], kc * 32, dt * 16, lane), db, dQ[dt]);
                }
            };
            if (CAUSAL || st * 64 + 64 > L) block(std::true_type{});
            else block(std::false_type{});
        }
        if (st + 1 < nst) sg.write(Ks[cur ^ 1], Vs[cur ^ 1], tid);
        __syncthreads();
    }
    if (q < L) A::store_t((elem*)p.dqkv + ((size_t)b * L + q) * ld + hd * 64, dQ, 0.125f, lane);
}

template <typename T, bool CAUSAL>
__global__ __launch_bounds__(TW * 64) void attn_bwd_dkv_tiled_kernel(AttnArgs p, int nsb) {
    using A = Attn<T>;
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    __shared__ __attribute__((aligned(16))) elem Qs[2][64 * RS];
    __shared__ __attribute__((aligned(16))) elem Gs[2][64 * RS];
    __shared__ float lse_s[2][64], del_s[2][64];
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pair = blockIdx.x / nsb, sb = blockIdx.x - pair * nsb, b = pair / p.H, hd = pair - b * p.H;
    const int HD = p.H * 64, L = p.L, Lp = attn_padded_len_dev(L);
    const size_t ld = (size_t)3 * HD;
    const elem* base = (const elem*)p.qkv + (size_t)b * L * ld + hd * 64;
    const elem* dO = (const elem*)p.dout + (size_t)b * L * HD + hd * 64;
    const int key = sb * TROWS + wave * 16 + c;
    if (p.win_n > 0 && (sb * TROWS >= p.win_row0 + p.win_n || sb * TROWS + TROWS <= p.win_row0)) return;  // window form: no wanted key in this workgroup
    const bool live = sb * TROWS + wave * 16 < L;
    const vec8 k0 = A::grow(base + HD, ld, key, L, 0, lane), k1 = A::grow(base + HD, ld, key, L, 1, lane);
    const vec8 v0 = A::grow(base + 2 * HD, ld, key, L, 0, lane), v1 = A::grow(base + 2 * HD, ld, key, L, 1, lane);
    f32x4 dK[4], dV[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int st_sel = p.sel_rows ? (p.sel_rows[b] - b * L) >> 6 : -1;  // dO is zero outside this 64-query stage
    const int st_lo = CAUSAL ? (sb * TROWS) >> 6 : 0;  // causal: query stages that hold a query >= this workgroup's first key
    const int st0 = st_sel >= 0 ? (st_sel > st_lo ? st_sel : st_lo) : st_lo, nst = st_sel >= 0 ? st_sel + 1 : (L + 63) / 64;
    Stage64<T> sg;
    float lse_v = 0.f, del_v = 0.f;
    auto load_stats = [&](int st) {
        if (tid < 64) {
            const int qi = st * 64 + tid;
            lse_v = qi < L ? -p.lse[(size_t)pair * Lp + qi] * LOG2E : -INFINITY;  // padding queries: p = 0
            del_v = qi < L ? p.delta[(size_t)pair * Lp + qi] : 0.f;
        }
    };
    if (st0 < nst) {
        sg.load(base, ld, dO, (size_t)HD, st0 * 64, L, tid);
        load_stats(st0);
        sg.write(Qs[0], Gs[0], tid);
        if (tid < 64) { lse_s[0][tid] = lse_v; del_s[0][tid] = del_v; }
    }
    __syncthreads();
    for (int st = st0; st < nst; ++st) {
        const int cur = (st - st0) & 1;
        if (st + 1 < nst) { sg.load(base, ld, dO, (size_t)HD, (st + 1) * 64, L, tid); load_stats(st + 1); }
        if (live && (!CAUSAL || st * 64 + 63 >= sb * TROWS + wave * 16)) {
#pragma unroll
            for (int qc = 0; qc < 2; ++qc) {
                f32x4 P[2], dS[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int qt = 2 * qc + t;
                    const f32x4 nl = *(const f32x4*)(lse_s[cur] + qt * 16 + 4 * g);
                    const f32x4 d4 = *(const f32x4*)(del_s[cur] + qt * 16 + 4 * g);
                    f32x4 S = {0.f, 0.f, 0.f, 0.f}, dP = -d4;
                    S = T::mfma16(A::rows(Qs[cur], qt * 16, 0, lane), k0, S);
                    S = T::mfma16(A::rows(Qs[cur], qt * 16, 1, lane), k1, S);
                    dP = T::mfma16(A::rows(Gs[cur], qt * 16, 0, lane), v0, dP);
                    dP = T::mfma16(A::rows(Gs[cur], qt * 16, 1, lane), v1, dP);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(S[r], SC, nl[r]));
                        if (CAUSAL && key > st * 64 + qt * 16 + 4 * g + r) pr = 0.f;
                        P[t][r] = pr;
                        dS[t][r] = pr * dP[r];
                    }
                }
                const vec8 pb = A::pack2(P[0], P[1]), db = A::pack2(dS[0], dS[1]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    dV[dt] = T::mfma16(A::cols(Gs[cur], qc * 32, dt * 16, lane), pb, dV[dt]);
                    dK[dt] = T::mfma16(A::cols(Qs[cur], qc * 32, dt * 16, lane), db, dK[dt]);
                }
            }
        }
        if (st + 1 < nst) {
            sg.write(Qs[cur ^ 1], Gs[cur ^ 1], tid);
            if (tid < 64) { lse_s[cur ^ 1][tid] = lse_v; del_s[cur ^ 1][tid] = del_v; }
        }
        __syncthreads();
    }
    if (key < L) {
        elem* ok = (elem*)p.dqkv + ((size_t)b * L + key) * ld + HD + hd * 64;
        A::store_t(ok, dK, 0.125f, lane);
        A::store_t(ok + HD, dV, 1.f, lane);
    }
}

template <typename T, bool BWD>
static int tiled_launch(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    const LaunchProf p1{prof ? prof->start : nullptr, nullptr}, p2{nullptr, prof ? prof->stop : nullptr};
    const int nsb = (a.L + TROWS - 1) / TROWS;
    const size_t nwg = (size_t)a.B * a.H * nsb;
    ARG_CHECK(nwg < 0x7fffffffull, "attention: too many workgroups (%zu)", nwg);
    const dim3 grid((unsigned)nwg), block(TW * 64);
    if (!BWD) {
        if (!a.tiled_fwd_16 && attn_resident_fits(a.L, false)) {  // K and V of a pair fit one CU's LDS (L <= 640): one workgroup per pair
            return launch_attn_fwd_resident(T::id, a, s, prof);
        } else {  // the staged 16-query-block form
            if (a.causal) MUDPT_LAUNCH((attn_fwd_tiled_kernel<T, true>), grid, block, 0, s, prof, a, nsb);
            else MUDPT_LAUNCH((attn_fwd_tiled_kernel<T, false>), grid, block, 0, s, prof, a, nsb);
        }
    } else if (!a.two_kernels && !a.sel_rows && a.win_n <= 0 && attn_resident_fits(a.L, true)) {
        return launch_attn_bwd_resident(T::id, a, s, prof);
    } else {
        if (a.causal) {
            MUDPT_LAUNCH((attn_bwd_dq_tiled_kernel<T, true>), grid, block, 0, s, &p1, a, (const void*)a.out, nsb);
            MUDPT_LAUNCH((attn_bwd_dkv_tiled_kernel<T, true>), grid, block, 0, s, &p2, a, nsb);
        } else {
            MUDPT_LAUNCH((attn_bwd_dq_tiled_kernel<T, false>), grid, block, 0, s, &p1, a, (const void*)a.out, nsb);
            MUDPT_LAUNCH((attn_bwd_dkv_tiled_kernel<T, false>), grid, block, 0, s, &p2, a, nsb);
        }
    }
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int check(const AttnArgs& a, bool bwd) {
    ARG_CHECK(a.qkv && a.B > 0 && a.L > 0 && a.H > 0, "attention: bad arguments B=%d L=%d H=%d", a.B, a.L, a.H);
    ARG_CHECK(a.L <= 4096, "attention: L=%d exceeds the supported 4096 rows", a.L);
    ARG_CHECK((uintptr_t)a.qkv % 16 == 0, "attention: qkv must be 16-byte aligned");
    ARG_CHECK((size_t)a.B * a.L * 3 * a.H * 64 * 2 < 0x7fffffffull, "attention: qkv larger than 2 GiB");  // 32-bit DMA offsets
    if (!bwd) ARG_CHECK(a.out && (uintptr_t)a.out % 16 == 0, "attention: null/unaligned out");
    if (bwd) ARG_CHECK(a.out && a.dout && a.dqkv && a.lse && a.delta, "attention bwd: null operand");
    return MUDPT_OK;
}

template <typename K>
static int set_lds(K kern, int bytes) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return MUDPT_OK;
}

template <typename T, int NC, bool CAUSAL>
static int fwd_pair_cfg(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    constexpr int lds = 2 * NC * 32 * RS * 2 + NC * 32 * 4;
    auto kern = attn_fwd_pair_kernel<T, NC, CAUSAL>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) { if (int e = set_lds(kern, lds)) return e; pd.done[dev] = true; }
    MUDPT_LAUNCH(kern, dim3(a.B * a.H), dim3(NC * 64), lds, s, prof, a);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

template <typename T, int NC, bool CAUSAL>
static int fwd_cfg(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    if constexpr (CAUSAL) return fwd_pair_cfg<T, NC, CAUSAL>(a, s, prof);
    constexpr int lds = 4 * NC * 32 * 128 + NC * 32 * 4;  // two (K, V) image pairs + the key mask
    auto kern = attn_fwd_kernel<T, NC, CAUSAL>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        if (int e = set_lds(kern, lds)) return e;
        HIP_TRY(hipDeviceGetAttribute(&pd.ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        pd.done[dev] = true;
    }
    const int ncu = pd.ncu[dev];
    const int npairs = a.B * a.H, per_cu = 163840 / lds > 0 ? 163840 / lds : 1;
    const int cap = ncu * (per_cu * 2 * NC <= 32 ? per_cu : 32 / (2 * NC));  // resident workgroups: LDS and the 32-wave limit
    MUDPT_LAUNCH(kern, dim3(npairs < cap ? npairs : cap), dim3(NC * 128), lds, s, prof, a, npairs);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

template <typename T, int NC, bool CAUSAL>
static int bwd_cfg(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    const LaunchProf p1{prof ? prof->start : nullptr, nullptr}, p2{nullptr, prof ? prof->stop : nullptr};
    constexpr int lds1 = 2 * NC * 32 * RS * 2 + NC * 32 * 4;
    constexpr int lds2 = 2 * NC * 32 * RS * 2 + 2 * NC * 32 * 4;
    auto k1 = attn_bwd_dq_kernel<T, NC, CAUSAL>;
    auto k2 = attn_bwd_dkv_kernel<T, NC, CAUSAL>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        if (int e = set_lds(k1, lds1)) return e;
        if (int e = set_lds(k2, lds2)) return e;
        pd.done[dev] = true;
    }
    MUDPT_LAUNCH(k1, dim3(a.B * a.H), dim3(NC * 64), lds1, s, &p1, a, (const void*)a.out);
    MUDPT_LAUNCH(k2, dim3(a.B * a.H), dim3(NC * 64), lds2, s, &p2, a);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// fused backward (one resident pass): everything except the sel_rows form of the last block
template <typename T, int NC, bool CAUSAL, int W2>
static int bwd_fused_cfg(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    constexpr int lds = 4 * NC * 32 * 128 + 3 * NC * 32 * 4;
    auto kern = attn_bwd_fused_kernel<T, NC, CAUSAL, W2>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) {
        if (int e = set_lds(kern, lds)) return e;
        HIP_TRY(hipDeviceGetAttribute(&pd.ncu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        pd.done[dev] = true;
    }
    const int npairs = a.B * a.H, by_lds = 163840 / lds, by_waves = 32 / (NC * W2);
    int per_cu = by_lds < by_waves ? by_lds : by_waves;
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int cap = pd.ncu[dev] * per_cu;
    MUDPT_LAUNCH(kern, dim3(npairs < cap ? npairs : cap), dim3(NC * 64 * W2), lds, s, prof, a, (const void*)a.out, npairs);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

// single sweep (non-causal, whole sequence on chip, one workgroup per pair)
template <typename T, int NC>
static int bwd_sweep_cfg(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    constexpr int lds = 2 * NC * 32 * 128 + NC * 32 * NC * 32 * 2 + 2 * NC * 32 * 4;
    static_assert(lds <= 163840, "attn_bwd_sweep_kernel: LDS");
    auto kern = attn_bwd_sweep_kernel<T, NC>;
    static PerDevice pd;
    const int dev = current_device();
    if (!pd.done[dev]) { if (int e = set_lds(kern, lds)) return e; pd.done[dev] = true; }
    MUDPT_LAUNCH(kern, dim3(a.B * a.H), dim3(NC * 128), lds, s, prof, a, (const void*)a.out);
    HIP_TRY(hipGetLastError());
    return MUDPT_OK;
}

template <typename T, bool BWD>
static int dispatch(const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    if (a.L > 224) return tiled_launch<T, BWD>(a, s, prof);  // the other operand streams through 64-row stages
    const int nc = attn_padded_len(a.L) / 32;
#define MUDPT_ATTN_CASE(N)                                                                     \
    case N:                                                                                    \
        if (BWD && a.win_n > 0 && !a.sel_rows) return a.causal ? bwd_cfg<T, N, true>(a, s, prof) : bwd_cfg<T, N, false>(a, s, prof); /* window: two kernels */ \
        if (BWD && !a.sel_rows && !a.causal && !a.two_kernels && !a.force_fused && (N >= 4 || a.sweep)) return bwd_sweep_cfg<T, N>(a, s, prof); \
        if (BWD && !a.sel_rows && !a.two_kernels && (N <= 3 || a.force_fused)) {                   \
            if (a.fused_w1) return a.causal ? bwd_fused_cfg<T, N, true, 1>(a, s, prof) : bwd_fused_cfg<T, N, false, 1>(a, s, prof); \
            return a.causal ? bwd_fused_cfg<T, N, true, 2>(a, s, prof) : bwd_fused_cfg<T, N, false, 2>(a, s, prof); \
        }                                                                                          \
        if (a.causal) return BWD ? bwd_cfg<T, N, true>(a, s, prof) : fwd_cfg<T, N, true>(a, s, prof);      \
        return BWD ? bwd_cfg<T, N, false>(a, s, prof) : fwd_cfg<T, N, false>(a, s, prof);
    switch (nc) {
        MUDPT_ATTN_CASE(1)
        MUDPT_ATTN_CASE(2)
        MUDPT_ATTN_CASE(3)
        MUDPT_ATTN_CASE(4)
        MUDPT_ATTN_CASE(5)
        MUDPT_ATTN_CASE(6)
        MUDPT_ATTN_CASE(7)
    }
#undef MUDPT_ATTN_CASE
    set_error("attention: unsupported padded length %d", nc * 32);
    return MUDPT_ERR_ARG;
}

int launch_attn_fwd(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    if (int e = check(a, false)) return e;
    if (dtype == DT_BF16) return dispatch<BF16, false>(a, s, prof);
    if (dtype == DT_F16) return dispatch<F16, false>(a, s, prof);
    set_error("attention: unknown dtype %d", dtype);
    return MUDPT_ERR_ARG;
}

int launch_attn_bwd(int dtype, const AttnArgs& a, hipStream_t s, const LaunchProf* prof) {
    if (int e = check(a, true)) return e;
    if (dtype == DT_BF16) return dispatch<BF16, true>(a, s, prof);
    if (dtype == DT_F16) return dispatch<F16, true>(a, s, prof);
    set_error("attention: unknown dtype %d", dtype);
    return MUDPT_ERR_ARG;
}

}  // namespace mudpt
